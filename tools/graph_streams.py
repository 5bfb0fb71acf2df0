"""Development aid: the headline mode (S independent configs[1] jobs in flight, one library context + stream each)
with every slot's job replayed as ONE HIP graph instead of ~45 launches -- does the GPU schedule twelve graphs better
than twelve streams of eager launches?   python tools/graph_streams.py [streams] [seconds]"""
import os as _os, sys as _sys
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import sys
import time

import torch

from world_amd import synth
from world_amd.api import WorldHip, frame_count

S = int(sys.argv[1]) if len(sys.argv) > 1 else 12
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
FS, FFT = 48000, 2048
dev = torch.device("cuda", 0)
xs = [synth.vowel(FS, seconds, seed=12345 + 977 * k, base_f0=140.0 + 7.0 * ((5 * k) % 12), device=dev)[None].contiguous()
      for k in range(S)]
nf = frame_count(FS, xs[0].shape[1], 5.0)
streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
whs = [WorldHip(device=0) for _ in range(S)]
blks = [torch.zeros((nf, 2 + 2 * (FFT // 2 + 1)), dtype=torch.float64, device=dev) for _ in range(S)]


def eager(k):
    with torch.cuda.stream(streams[k]):
        whs[k].analyze_packed(xs[k], FS, blks[k], frame_period=5.0)


for k in range(S):
    eager(k); eager(k)
torch.cuda.synchronize()
want = [b.clone() for b in blks]
graphs = []
for k in range(S):
    with torch.cuda.stream(streams[k]):
        graphs.append(whs[k].capture(lambda k=k: whs[k].analyze_packed(xs[k], FS, blks[k], frame_period=5.0)))
torch.cuda.synchronize()


def replay(k):
    with torch.cuda.stream(streams[k]):
        graphs[k].launch()


def rate(fn, jobs):
    for k in range(2 * S):
        fn(k % S)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for j in range(jobs):
        fn(j % S)
    torch.cuda.synchronize()
    return nf * jobs / (time.perf_counter() - t0)


for rep in range(2):
    print(f"{S} jobs in flight: eager {rate(eager, 600) / 1e6:.3f} M frames/s, graph replay {rate(replay, 600) / 1e6:.3f} M frames/s", flush=True)
for b in blks:
    b.fill_(-1.0)
for k in range(S):
    replay(k)
torch.cuda.synchronize()
print("replays bit-identical to eager:", all(torch.equal(a, b) for a, b in zip(blks, want)))
