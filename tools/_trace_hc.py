import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import ctypes as C, os, subprocess, sys
os.environ["WORLD_HIP_EXTRA_FLAGS"] = "-DWH_TRACE"
subprocess.run([sys.executable, "-m", "world_amd.build", "--force"], check=True, capture_output=True)
import torch
from world_amd import synth
from world_amd.api import WorldHip
wh = WorldHip()
x = synth.vowel(48000, 10.0, seed=12345, device=torch.device("cuda", 0))[None]
for _ in range(3):
    wh.analyze(x, 48000)
torch.cuda.synchronize()
buf = (C.c_longlong * 128)()
fn = wh.lib.world_hip_trace_read_hc
fn.argtypes = [C.POINTER(C.c_longlong), C.c_int]
assert fn(buf, 128) == 0
t = list(buf)
print("hc_extend sec 0: stamps", [t[i] - t[0] for i in range(6)], "len", t[10], "nslot", t[11], "sections", t[12], "dist", t[13:15], "moved", t[15:17], "walk ends", [t[17] - t[0], t[18] - t[0]])
print("hc_merge u 0: stamps", [t[32 + i] - t[32] for i in range(5)], "ns", t[40], "kept", t[41], "sub frames", t[42], "nf", t[43])
print("per section end-start (wall 100MHz ticks)", [t[80 + k] - t[96 + k] for k in range(6)], "len", t[64:70], "start rel", [t[96 + k] - t[96] for k in range(8)], "end rel", [t[80 + k] - t[96] for k in range(6)])
