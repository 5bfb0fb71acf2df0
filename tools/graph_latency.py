"""Development aid / measurement: one analysis job -- Harvest + CheapTrick + D4C
through the batched C API, ~45 kernel launches -- captured in a HIP graph (world_hip_graph_begin / _end, WorldHip.capture)
and replayed: host time per job and a lone job's latency, eager against replay; the replay's outputs must be
bit-identical.  Also replays S graphs on S streams (the bench's jobs-in-flight mode without the host cost).
    python tools/graph_latency.py [seconds] [streams]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
from world_amd import synth
from world_amd.api import WorldHip, cheaptrick_fft_size, frame_count

FS, FP = 48000, 5.0
sec = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
S = int(sys.argv[2]) if len(sys.argv) > 2 else 12
dev = torch.device("cuda", 0)
nb = cheaptrick_fft_size(FS) // 2 + 1
xs = [synth.vowel(FS, sec, seed=12345 + 977 * k, base_f0=140.0 + 7.0 * ((5 * k) % 12), device=dev)[None].contiguous() for k in range(S)]
nf = frame_count(FS, xs[0].shape[1], FP)
streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
whs = [WorldHip(device=0) for _ in range(S)]
blocks = [torch.zeros((nf, 2 + 2 * nb), dtype=torch.float64, device=dev) for _ in range(S)]


def eager(k):
    with torch.cuda.stream(streams[k]):
        whs[k].analyze_packed(xs[k], FS, blocks[k], frame_period=FP)


for _ in range(3):
    for k in range(S):
        eager(k)
torch.cuda.synchronize()
ref = [b.clone() for b in blocks]
graphs = []
for k in range(S):
    with torch.cuda.stream(streams[k]):
        graphs.append(whs[k].capture(lambda: whs[k].analyze_packed(xs[k], FS, blocks[k], frame_period=FP)))
for b in blocks:
    b.fill_(-1.0)
torch.cuda.synchronize()


def replay(k):
    with torch.cuda.stream(streams[k]):
        graphs[k].launch()


for k in range(S):
    replay(k)
torch.cuda.synchronize()
same = all(torch.equal(a, b) for a, b in zip(blocks, ref))


def lone(fn, n=20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn(0)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def host_cost(fn, n=10):
    """host time to ENQUEUE a job (the GPU is kept busy enough not to matter: time only the calls)"""
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n * S):
        fn(i % S)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / (n * S) * 1e3, nf * n * S / (t2 - t0)


out = {"seconds": sec, "streams": S, "frames_per_job": nf, "replay_bit_identical_to_eager": bool(same),
       "lone_job_ms": {"eager": lone(eager), "graph": lone(replay)}}
h_e, v_e = host_cost(eager)
h_g, v_g = host_cost(replay)
out["host_ms_per_job"] = {"eager": h_e, "graph": h_g}
out["frames_per_s_in_flight"] = {"eager": v_e, "graph": v_g}
print(json.dumps(out))
