"""Development aid: world_hip_analyze_coded against world_hip_analyze_packed on the same batch (the coded
call should cost <= 1.02 x the dense one now that the coders are fused into ct_frame / d4c_finish).
    python tools/coded_vs_dense.py [utterances = 64] [seconds = 5]"""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import sys
import time

import torch
from world_amd import synth
from world_amd.api import WorldHip, frame_count

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
sec = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
fs, nd = 48000, 60
dev = torch.device("cuda", 0)
wh = WorldHip()
x = torch.stack([synth.utterance(u, fs, sec, device=dev) for u in range(n)])
rows = n * frame_count(fs, x.shape[1], 5.0)
dense = torch.empty((rows, 2 + 2 * 1025), dtype=torch.float64, device=dev)
coded = torch.empty((rows, wh.lib.world_hip_coded_columns(fs, nd)), dtype=torch.float64, device=dev)


def timed(fn, reps=6):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for rnd in range(2):
    td = timed(lambda: wh.analyze_packed(x, fs, dense))
    tc = timed(lambda: wh.analyze_coded(x, fs, coded, number_of_dimensions=nd))
    print(f"[{rnd}] {n} x {sec:g} s: dense records {td:.2f} ms, coded records {tc:.2f} ms, coded / dense = {tc / td:.3f}")
