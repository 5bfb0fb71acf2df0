import time, torch
from world_amd import synth
from world_amd.api import WorldHip
dev = torch.device("cuda", 0)
x = synth.vowel(48000, 10.0, seed=12345, device=dev)[None]
wh = WorldHip()
s = torch.cuda.Stream()
sp = torch.empty((1, 2001, 1025), dtype=torch.float64, device=dev); ap = torch.empty_like(sp)
with torch.cuda.stream(s):
    for _ in range(3): wh.analyze(x, 48000, sp_out=sp, ap_out=ap)
torch.cuda.synchronize()
for n in (1, 1, 1, 4):
    t0 = time.perf_counter()
    with torch.cuda.stream(s):
        for _ in range(n): wh.analyze(x, 48000, sp_out=sp, ap_out=ap)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{n} jobs: submit {1e3*(t1-t0)/n:.3f} ms/job, total {1e3*(t2-t0)/n:.3f} ms/job")
