"""Helper of tests/test_dropin.py (run as a subprocess so that WORLD_HIP_SMALL_SLAB / WORLD_HIP_SMALL_BUDGET apply):
many batched calls with DISTINCT length vectors through ONE context -- every call uploads fresh small arrays (x_len,
n_frames, y_len, nfb, ref_fft, out_row) -- and the results of all of them, for comparison between slab settings.
Usage: slab_stress.py <library> <out.npz>"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from world_amd import synth                                          # noqa: E402
from world_amd.api import (CheapTrickOption, D4COption, HarvestOption, cheaptrick_fft_size, frame_count,   # noqa: E402
                           load_library)

lib_path, out = sys.argv[1], sys.argv[2]
emu = "emu" in os.path.basename(lib_path)
L = load_library(lib_path)
fs = 16000
fft = cheaptrick_fft_size(fs)
nb = fft // 2 + 1
B, Lmax = 3, 2400
base = np.stack([synth.utterance(i, fs, Lmax / fs).numpy() for i in range(B)])
ctx = L.world_hip_create(0, None)
assert ctx
hopt, copt, dopt = HarvestOption(71.0, 800.0, 5.0), CheapTrickOption(-0.15, 71.0, fft), D4COption(0.85)
rng = np.random.default_rng(11)
n_calls = 12 if emu else 60
results = {}
if emu:
    dev_alloc = lambda a: a                                           # "device" memory is host memory in the emulation
    to_host = lambda a: a
    ptr = lambda a: a.ctypes.data
    sync = lambda: None
else:
    import torch
    dev_alloc = lambda a: torch.from_numpy(a).cuda()
    to_host = lambda t: t.cpu().numpy()
    ptr = lambda t: t.data_ptr()
    sync = torch.cuda.synchronize
x_dev = dev_alloc(base.copy())
for call in range(n_calls):
    xl = np.ascontiguousarray(rng.integers(1200, Lmax + 1, size=B), dtype=np.int32)     # a new length vector every call
    nf = [frame_count(fs, int(n), 5.0) for n in xl]
    rows = sum(nf)
    block = dev_alloc(np.full((rows, 2 + 2 * nb), np.nan))
    rc = L.world_hip_analyze_packed(ctx, B, fs, ptr(x_dev), Lmax, xl.ctypes.data_as(C.POINTER(C.c_int)), C.byref(hopt),
                                    C.byref(copt), C.byref(dopt), 0, ptr(block), 2 + 2 * nb)
    assert rc == 0, L.world_hip_last_error().decode()
    sync()
    got = to_host(block)
    assert np.isfinite(got).all()
    results[f"call{call}"] = got.copy()
L.world_hip_destroy(ctx)
np.savez(out, slabs=np.int64(n_calls), **results)
