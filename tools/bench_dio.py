"""Development aid: BASELINE.json configs[4] -- 64 x (16 kHz, 5 s), Dio + StoneMask + CheapTrick(fft 1024) + D4C
in one batched call per stage -- timed on the GPU with the per-kernel breakdown (not the headline metric)."""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import json, sys, time
import torch
from world_amd import synth
from world_amd.api import WorldHip
B, fs, sec = (int(sys.argv[1]) if len(sys.argv) > 1 else 64), 16000, 5.0
x = torch.stack([synth.vowel(fs, sec, seed=100 + i, base_f0=90.0 + (i % 32) * 8.0) for i in range(B)])
x = (torch.round(x * 32768.0) / 32768.0).cuda().contiguous()
wh = WorldHip()
run = lambda: wh.analyze(x, fs, f0_method="dio")
for _ in range(3): run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): out = run()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 10 * 1e3
frames = int(out[4].sum())
prof = wh.profile(lambda: [run() for _ in range(3)])
k = sorted(((n, sum(v) / 3) for n, v in prof.items()), key=lambda kv: -kv[1])
print(json.dumps({"workload": f"{B} x ({fs} Hz, {sec:g} s) Dio+StoneMask+CheapTrick+D4C", "ms_per_call": ms,
                  "frames_per_s": frames / ms * 1e3, "kernels_ms": {n: round(v, 4) for n, v in k[:16]}}))
