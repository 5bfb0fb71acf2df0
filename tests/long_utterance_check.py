"""Test infrastructure (run by hand on the GPU box): one long utterance (default 60 s of 48 kHz audio)
through the whole path against the reference build -- sizes well beyond the bench's 10 s."""
import os, sys, time
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
import torch
from oracle.loader import best_oracle
from world_amd import synth
from world_amd.api import WorldHip, cheaptrick_fft_size
from util import max_rel
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
fs = 48000
x = torch.cat([synth.utterance(100 + i, fs, 10.0) for i in range(int(np.ceil(seconds / 10.0)))])[:int(seconds * fs)]
x = torch.round(x * 32768.0) / 32768.0
wh, o = WorldHip(), best_oracle()
t0 = time.time()
tpos, f0, sp, ap, nf = wh.analyze(x[None].cuda().contiguous(), fs)
torch.cuda.synchronize()
t1 = time.time()
tpos, f0, sp, ap, nf = wh.analyze(x[None].cuda().contiguous(), fs)
torch.cuda.synchronize()
t2 = time.time()
xn = x.numpy()
tp_o, f0_o = o.harvest(xn, fs)
fft = cheaptrick_fft_size(fs)
sp_o, ap_o = o.cheaptrick(xn, fs, tp_o, f0_o, fft_size=fft), o.d4c(xn, fs, tp_o, f0_o, fft)
t3 = time.time()
n = int(nf[0])
assert n == len(f0_o) and np.array_equal(tpos[0, :n].cpu().numpy(), tp_o)
f0g = f0[0, :n].cpu().numpy()
print(f"{seconds:g} s, {n} frames ({int((f0_o > 0).sum())} voiced): first call {t1 - t0:.2f} s, second {1e3 * (t2 - t1):.1f} ms, "
      f"{o.kind} oracle {t3 - t2:.1f} s; workspace {wh.workspace_bytes() / 1e9:.2f} GB")
print("v/uv flips", int(((f0g > 0) != (f0_o > 0)).sum()), " f0", max_rel(f0g, f0_o), " sp", max_rel(sp[0, :n].cpu().numpy(), sp_o),
      " ap", max_rel(ap[0, :n].cpu().numpy(), ap_o))
