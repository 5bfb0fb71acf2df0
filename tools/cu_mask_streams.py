"""Development aid: the headline mode (S independent configs[1] jobs in flight) with the jobs' streams confined to parts of
the chip (hipExtStreamCreateWithCUMask) or given priorities (hipStreamCreateWithPriority) -- does a partitioned chip lose
less to the jobs' narrow kernels than twelve streams that all see 256 CUs?  (profiles/r04/inflight_overlap.txt: never more
than three kernels resident, a narrow kernel waits ~70 us for a wave slot behind another job's d4c_frame.)
    python tools/cu_mask_streams.py [seconds]"""
import os as _os, sys as _sys
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import ctypes as C
import sys
import time

import torch

from world_amd import synth
from world_amd.api import WorldHip, frame_count

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
FS, FFT = 48000, 2048
dev = torch.device("cuda", 0)
torch.cuda.init()
hip = C.CDLL("libamdhip64.so")
CUS = torch.cuda.get_device_properties(0).multi_processor_count
WORDS = (CUS + 31) // 32


def masked_stream(cus):
    mask = (C.c_uint32 * WORDS)()
    for i in cus:
        mask[i // 32] |= 1 << (i % 32)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), C.c_uint32(WORDS), mask)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)


def priority_stream(prio):
    s = C.c_void_p()
    rc = hip.hipStreamCreateWithPriority(C.byref(s), C.c_uint(1), C.c_int(prio))      # 1 = hipStreamNonBlocking
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)


def groups(n_groups, how):
    """CU index sets: 'mod' = CU i belongs to group i % n_groups (whole XCDs if the mask interleaves XCDs), 'div' = contiguous"""
    if how == "mod":
        return [[i for i in range(CUS) if i % n_groups == g] for g in range(n_groups)]
    per = CUS // n_groups
    return [list(range(g * per, (g + 1) * per)) for g in range(n_groups)]


def run(name, streams, jobs=480):
    S = len(streams)
    xs = [synth.vowel(FS, seconds, seed=12345 + 977 * k, base_f0=140.0 + 7.0 * ((5 * k) % 12), device=dev)[None].contiguous()
          for k in range(S)]
    nf = frame_count(FS, xs[0].shape[1], 5.0)
    whs = [WorldHip(device=0) for _ in range(S)]
    sp = [torch.empty((1, nf, FFT // 2 + 1), dtype=torch.float64, device=dev) for _ in range(S)]
    ap = [torch.empty_like(sp[0]) for _ in range(S)]

    def job(k):
        with torch.cuda.stream(streams[k]):
            whs[k].analyze(xs[k], FS, sp_out=sp[k], ap_out=ap[k])

    for k in range(2 * S):
        job(k % S)
    torch.cuda.synchronize()
    best = 0.0
    for _ in range(2):
        t0 = time.perf_counter()
        for j in range(jobs):
            job(j % S)
        torch.cuda.synchronize()
        best = max(best, nf * jobs / (time.perf_counter() - t0))
    print(f"{name:44s} {S:2d} streams  {best / 1e6:.3f} M frames/s", flush=True)
    for w in whs:
        w.close()
    del sp, ap, xs
    torch.cuda.empty_cache()


run("plain streams", [torch.cuda.Stream(device=dev) for _ in range(12)])
lo, hi = C.c_int(0), C.c_int(0)
hip.hipDeviceGetStreamPriorityRange(C.byref(lo), C.byref(hi))
print("priority range (least, greatest):", lo.value, hi.value)
run("priorities alternating", [priority_stream(hi.value if k % 2 else lo.value) for k in range(12)])
for how in ("mod", "div"):
    for n_groups, per_group in ((2, 6), (4, 3), (4, 4), (8, 2)):
        gs = groups(n_groups, how)
        run(f"{n_groups} groups of {CUS // n_groups} CUs ({how}), {per_group} streams each",
            [masked_stream(gs[k % n_groups]) for k in range(n_groups * per_group)])
run("plain streams again", [torch.cuda.Stream(device=dev) for _ in range(12)])
