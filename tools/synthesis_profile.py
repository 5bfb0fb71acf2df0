"""Development aid: analyse one 10 s utterance, then run Synthesis() on the device-resident parameters a few times --
the command to put under `rocprofv3 --kernel-trace --stats` for the per-kernel times of the synthesis path.
    python tools/synthesis_profile.py [seconds] [repeats]"""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import sys
import time

import torch

from world_amd import synth
from world_amd.api import WorldHip

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
fs = 48000
wh = WorldHip()
x = synth.vowel(fs, seconds, seed=12345, device=torch.device("cuda", 0))[None]
tpos, f0, sp, ap, nf = wh.analyze(x, fs)
n = x.shape[1]
y = wh.synthesis(f0, sp, ap, nf, 2048, 5.0, fs, n)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    wh.synthesis(f0, sp, ap, nf, 2048, 5.0, fs, n, check_pulses=False)
torch.cuda.synchronize()
print(f"Synthesis() of {seconds:g} s: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms")
