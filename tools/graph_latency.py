"""Development aid / experiment: can one analysis job -- Harvest + CheapTrick + D4C through the batched
C API, ~45 kernel launches and a few pinned-memory uploads -- be captured in a HIP graph as the library stands, and
what does replaying it do to a lone job's latency?  The library calls made while torch captures the stream must not
allocate, synchronise or wait on events recorded inside the capture; contexts and workspace are warmed up first.
    python tools/graph_latency.py [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
from world_amd import synth
from world_amd.api import WorldHip

FS, FP = 48000, 5.0
sec = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
dev = torch.device("cuda", 0)
x = synth.vowel(FS, sec, seed=12345, device=dev)[None].contiguous()
s = torch.cuda.Stream(device=dev)
wh = WorldHip(device=0)
with torch.cuda.stream(s):
    for _ in range(3):
        ref = wh.analyze(x, FS, frame_period=FP)
torch.cuda.synchronize()


def timed(fn, n=20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def plain():
    with torch.cuda.stream(s):
        wh.analyze(x, FS, frame_period=FP)


print("plain launches: %.3f ms per job" % timed(plain), flush=True)
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g, stream=s):
        out = wh.analyze(x, FS, frame_period=FP)
except Exception as e:                                   # noqa: BLE001 -- the experiment's negative outcome
    print("capture failed:", type(e).__name__, str(e)[:300])
    sys.exit(0)
g.replay()
torch.cuda.synchronize()
same = all(torch.equal(a, b) for a, b in zip(out[:4], ref[:4]))
print("graph replay:   %.3f ms per job, outputs bit-identical to plain launches: %s" % (timed(g.replay), same))
