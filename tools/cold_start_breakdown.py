"""Where a fresh process's first analysis goes (development aid; run on the GPU box): the HIP runtime's start-up, the
library's first context (code objects, twiddle / jump tables), a second context (what a context costs once the code is
loaded), the first batched Harvest / CheapTrick / D4C on that context (filter bank, workspace, randn table), the second."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
t0 = time.perf_counter()
hip = C.CDLL("libamdhip64.so")
hip.hipInit(0); hip.hipSetDevice(0); hip.hipFree(None)
t_rt = time.perf_counter()
import numpy as np                                                    # noqa: E402
import torch                                                          # noqa: E402
from world_amd import synth                                           # noqa: E402
from world_amd.api import CheapTrickOption, D4COption, HarvestOption, frame_count, load_library   # noqa: E402
x = synth.vowel(48000, 10.0, seed=12345)
xd = x.cuda()[None].contiguous()
torch.cuda.synchronize()
t_in = time.perf_counter()
L = load_library()
L.world_hip_noise_table_build_ms.restype = C.c_double
L.world_hip_noise_table_build_ms.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
ta = time.perf_counter()
ctx = L.world_hip_create(0, None)
tb = time.perf_counter()
ctx2 = L.world_hip_create(0, None)
tc = time.perf_counter()
n = xd.shape[1]
nf = frame_count(48000, n, 5.0)
xl = np.array([n], dtype=np.int32)
nfa = np.array([nf], dtype=np.int32)
ip = C.POINTER(C.c_int)
tp = torch.zeros((1, nf), dtype=torch.float64, device="cuda"); f0 = torch.zeros_like(tp)
sp = torch.zeros((1, nf, 1025), dtype=torch.float64, device="cuda"); ap = torch.zeros_like(sp)
h, c, d = HarvestOption(71.0, 800.0, 5.0), CheapTrickOption(-0.15, 71.0, 2048), D4COption(0.85)


def stages():
    out = []
    for fn in (lambda: L.world_hip_harvest_batch(ctx, 1, 48000, xd.data_ptr(), n, xl.ctypes.data_as(ip), C.byref(h), nf, tp.data_ptr(), f0.data_ptr()),
               lambda: L.world_hip_cheaptrick_batch(ctx, 1, 48000, xd.data_ptr(), n, xl.ctypes.data_as(ip), nfa.ctypes.data_as(ip), nf, tp.data_ptr(), f0.data_ptr(), C.byref(c), sp.data_ptr()),
               lambda: L.world_hip_d4c_batch(ctx, 1, 48000, xd.data_ptr(), n, xl.ctypes.data_as(ip), nfa.ctypes.data_as(ip), nf, tp.data_ptr(), f0.data_ptr(), 2048, C.byref(d), ap.data_ptr())):
        t = time.perf_counter()
        assert fn() == 0
        torch.cuda.synchronize()
        out.append(round((time.perf_counter() - t) * 1e3, 2))
    return out


first, second = stages(), stages()
b = C.c_int()
print("hip runtime start %.1f ms | first world_hip_create %.1f ms, second %.1f ms | first Harvest / CheapTrick / D4C %s ms, "
      "second %s ms | randn table %.1f ms in %d builds" % ((t_rt - t0) * 1e3, (tb - ta) * 1e3, (tc - tb) * 1e3, first, second,
                                                          L.world_hip_noise_table_build_ms(ctx, C.byref(b)), b.value))
