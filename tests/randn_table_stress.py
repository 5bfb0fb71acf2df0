"""Test infrastructure (run by tests/test_gpu_fullsize.py in its own process, or by hand on the GPU box):
the process-wide randn table under growth.  Two contexts on two streams interleave CheapTrick and D4C calls on
utterances of increasing length, so the table is rebuilt several times (each rebuild verified word for word
against the host generator, rng_fill.hip) while the other context's kernels are still reading the previous
generation.  After every step D4C and CheapTrick of a fixed probe utterance must reproduce their first results
bit for bit; at the end the probe is compared with the reference and the live table is verified once more.
Prints one JSON line."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
import torch  # noqa: E402
from oracle.loader import best_oracle  # noqa: E402
from world_amd import synth  # noqa: E402
from world_amd.api import WorldHip  # noqa: E402
from util import max_rel  # noqa: E402

fs, fft = 48000, 2048
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
a, b = WorldHip(), WorldHip()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
probe = synth.utterance(7, fs, 1.2).cuda().unsqueeze(0)           # ~241 frames: needs ~3.8 M draws
with torch.cuda.stream(sa):
    tp, f0, nf = a.harvest(probe, fs)
    sp0 = a.cheaptrick(probe, fs, tp, f0, nf, fft_size=fft)
    ap0 = a.d4c(probe, fs, tp, f0, nf, fft)
torch.cuda.synchronize()
sizes = [a.noise_table_bytes()]
d4c_ok = ct_ok = True
n_d4c = 0
# growing utterances: 2 s .. 14 s => 6 M .. 44 M draws for D4C: the table doubles four times on the way
for step, seconds in enumerate([2.0, 3.5, 5.0, 7.0, 10.0, 14.0] + [1.0] * max(0, reps // 2 - 6)):
    big = synth.utterance(20 + step, fs, seconds).cuda().unsqueeze(0)
    with torch.cuda.stream(sb):                                    # context b: the call that makes the table grow
        tpb, f0b, nfb = b.harvest(big, fs)
        spb = b.cheaptrick(big, fs, tpb, f0b, nfb, fft_size=fft)
    with torch.cuda.stream(sa):                                    # context a keeps reading whatever generation it was given
        ap1 = a.d4c(probe, fs, tp, f0, nf, fft)
    with torch.cuda.stream(sb):
        apb = b.d4c(big, fs, tpb, f0b, nfb, fft)
    with torch.cuda.stream(sa):
        sp1 = a.cheaptrick(probe, fs, tp, f0, nf, fft_size=fft)
        ap2 = a.d4c(probe, fs, tp, f0, nf, fft)
    torch.cuda.synchronize()
    d4c_ok &= torch.equal(ap1, ap0) and torch.equal(ap2, ap0)
    ct_ok &= torch.equal(sp1, sp0)
    n_d4c += 2
    if a.noise_table_bytes() != sizes[-1]:
        sizes.append(a.noise_table_bytes())
o = best_oracle()
x = probe[0].cpu().numpy()
n = int(nf[0])
tp_h, f0_h = tp[0, :n].cpu().numpy(), f0[0, :n].cpu().numpy()
e_ap = max_rel(ap0[0, :n].cpu().numpy(), o.d4c(x, fs, tp_h, f0_h, fft))
e_sp = max_rel(sp0[0, :n].cpu().numpy(), o.cheaptrick(x, fs, tp_h, f0_h, fft_size=fft))
print(json.dumps({"generations": len(sizes), "table_bytes_live_plus_superseded": sizes, "d4c_repetitions": n_d4c,
                  "d4c_bit_stable": bool(d4c_ok), "cheaptrick_bit_stable": bool(ct_ok), "table_intact": bool(a.verify_tables()),
                  "d4c_vs_reference": e_ap, "cheaptrick_vs_reference": e_sp, "oracle": o.kind}))
