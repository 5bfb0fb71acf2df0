"""Development aid: build with -DWH_TRACE, run one analysis, print the cycle stamps one
workgroup of d4c_band / d4c_groupdelay recorded (deltas in shader-clock cycles)."""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import ctypes as C
import os
import subprocess
import sys

os.environ["WORLD_HIP_EXTRA_FLAGS"] = "-DWH_TRACE"
subprocess.run([sys.executable, "-m", "world_amd.build", "--force"], check=True, capture_output=True)
import torch
from world_amd import synth
from world_amd.api import WorldHip
wh = WorldHip()
# seconds of signal (argv[1], default 10): at 5.1 s the traced frame 1000 falls into the launch's second round of
# workgroups, which has about one workgroup per CU -- the phases' latencies without co-resident workgroups
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
x = synth.vowel(48000, seconds, seed=12345, device=torch.device("cuda", 0))[None]
for _ in range(3):
    wh.analyze(x, 48000)
torch.cuda.synchronize()
def stamps(unit):
    buf = (C.c_longlong * 128)()
    fn = getattr(wh.lib, "world_hip_trace_read_" + unit)
    fn.argtypes = [C.POINTER(C.c_longlong), C.c_int]
    assert fn(buf, 128) == 0
    return list(buf)


for unit, name, base, n in (("d4c", "d4c_frame select (last band)", 0, 10), ("d4c", "d4c_frame", 32, 20), ("ct", "ct_frame", 0, 11)):
    t = stamps(unit)[base:base + n]
    print(name, "total", t[-1] - t[0], "cycles")
    prev = t[0]
    for k in range(1, n):
        if t[k]:
            print(f"  stamp {k:2d}: +{t[k] - prev}")
            prev = t[k]

t = stamps("d4c")
print("d4c_frame band selections since the library was loaded: one pass", t[100], "two passes", t[101], "general routine", t[102])
t = stamps("hv")[:8]
print("hv_refine (one wave, frame 5000): cache fill", t[0], "window rebuilds", t[1], "DFT+reduce", t[2], "tails", t[3],
      "refined candidates", t[4])

t = stamps("hv")[16:24]
print("hv_band_events (workgroup seg 5, band 20): setup", t[0], "tile fetch/commit", t[1], "FIR", t[2], "events", t[3], "taps", t[4])

t = stamps("hv")[24:31]
print("hv_band_events_fft (band 20, chunk 0, first block): twiddle table", t[1] - t[0], "H to registers", t[2] - t[1],
      "X * H + pre-twiddle", t[3] - t[2], "c2r stages", t[4] - t[3], "mirror-store term", t[5] - t[4], "events", t[6] - t[5])
t = stamps("hv")[32:35]
print("   its events phase: sample reads + masks", t[0], "block scan", t[1], "edge times + stores", t[2])
