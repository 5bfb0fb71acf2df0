"""Development aid: cycle stamps of one ct_frame and one d4c_frame workgroup in the middle of a BATCH launch (every CU loaded
with other frames' workgroups -- what tools/trace.py's lone utterance cannot show: there the stamped workgroup has the CU
almost to itself).  Build the traced variant here (no GPU needed), run on the GPU box:
    python tools/ab.py build trace="-DWH_TRACE -DWH_TRACE_FRAME=500 -DWH_TRACE_UTT=40"
    python tools/trace_batch.py [variant name = trace]"""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import ctypes as C
import os
import sys

name = sys.argv[1] if len(sys.argv) > 1 else "trace"
os.environ["WORLD_HIP_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "world_amd", "variants",
                                           f"libworld_hip_{name}.so")
import torch
from world_amd import synth
from world_amd.api import WorldHip
wh = WorldHip()
x = torch.stack([synth.vowel(48000, 5.0, seed=100 + u, device=torch.device("cuda", 0)) for u in range(64)])
for _ in range(3):
    wh.analyze(x, 48000)
torch.cuda.synchronize()


def stamps(unit):
    buf = (C.c_longlong * 128)()
    fn = getattr(wh.lib, "world_hip_trace_read_" + unit)
    fn.argtypes = [C.POINTER(C.c_longlong), C.c_int]
    assert fn(buf, 128) == 0
    return list(buf)


print("variant", name, "-- one workgroup of a 64 x 5 s batch (WH_TRACE_UTT / WH_TRACE_FRAME of the build)")
t = stamps("ct")[:11]
names = ["setup", "window", "rfft", "dc", "segment", "scan", "smooth+log", "rfft+lifter", "irfft", "exp+store"]
print("ct_frame: total", t[10] - t[0], "cycles")
for k in range(1, 11):
    print(f"  {names[k - 1]:12s} +{t[k] - t[k - 1]}")
t = stamps("d4c")[32:52]
dn = ["win c0", "even c0", "odd c0", "centroid c0", "win c1", "even c1", "odd c1", "centroid c1", "centroid sum", "smoothed: window",
      "smoothed: fft", "smoothed: dc+smooth", "group delay 1", "group delay 2", "group delay 3", "band 0: window", "band 0: fft",
      "band 0: select", "all bands"]
print("d4c_frame: total", t[19] - t[0], "cycles")
prev = t[0]
for k in range(1, 20):
    if t[k]:
        print(f"  {k:2d} {dn[k - 1]:20s} +{t[k] - prev}")
        prev = t[k]
t = stamps("d4c")[0:10]
print("  select (last band) stamps:", [t[k] - t[0] for k in range(1, 10) if t[k]])
t = stamps("hv")[:8]
print("hv_refine (one wavefront, utterance / 1 ms frame of the build): cache fill", t[0], "window rebuilds", t[1], "DFT+reduce", t[2],
      "tails", t[3], "refined candidates", t[4])
t = stamps("hv")[24:31]
print("hv_band_events_fft (band 20, first block): twiddle table", t[1] - t[0], "H to registers", t[2] - t[1],
      "X * H + pre-twiddle", t[3] - t[2], "c2r stages", t[4] - t[3], "mirror-store term", t[5] - t[4], "events", t[6] - t[5])
t = stamps("hv")[32:35]
print("   its events phase (summed over the chunk's blocks): sample reads + masks", t[0], "block scan", t[1], "edge times + stores", t[2])
