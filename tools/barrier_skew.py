"""Development aid: per-WAVEFRONT arrival times at every workgroup barrier of one d4c_frame workgroup in the middle of a
batch launch (every CU loaded) -- VERDICT r05 task 1(a): is one wavefront the one the others wait for?
Build the traced variant here (no GPU needed), run on the GPU box:
    python tools/ab.py build bartrace="-DWH_BARTRACE -DWH_TRACE_FRAME=500 -DWH_TRACE_UTT=40"
    python tools/barrier_skew.py [variant name = bartrace] [runs = 5]
Prints, per barrier ordinal: the spread (last arrival - first), last - median, which wavefront came last; then, over the
whole kernel, how often each wavefront was last and how many cycles the OTHERS spent waiting for it."""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import ctypes as C
import os
import sys

name = sys.argv[1] if len(sys.argv) > 1 else "bartrace"
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 5
os.environ["WORLD_HIP_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "world_amd", "variants",
                                           f"libworld_hip_{name}.so")
import torch
from world_amd import synth
from world_amd.api import WorldHip
wh = WorldHip()
x = torch.stack([synth.vowel(48000, 5.0, seed=100 + u, device=torch.device("cuda", 0)) for u in range(64)])
NW = 4


def read():
    buf = (C.c_longlong * (128 * 16))()
    fn = wh.lib.world_hip_bartrace_read_d4c
    fn.argtypes = [C.POINTER(C.c_longlong), C.c_int]
    assert fn(buf, 128 * 16) == 0
    rows = []
    for i in range(128):
        r = [buf[i * 16 + w] for w in range(NW)]
        if not all(r):
            break
        rows.append(r)
    return rows


tot_last = [0] * NW
tot_wait = [0] * NW
for run in range(runs):
    wh.analyze(x, 48000)
    torch.cuda.synchronize()
    rows = read()
    if not rows:
        print("no stamps: is the traced frame voiced and the variant built with -DWH_BARTRACE?")
        sys.exit(1)
    t0 = min(rows[0])
    span = max(rows[-1]) - t0
    last_count = [0] * NW
    wait_for = [0] * NW
    skew_sum = 0
    if run == runs - 1:
        print(f"run {run}: {len(rows)} barriers, {span} cycles from the first arrival at barrier 0 to the last at barrier {len(rows) - 1}")
        print(" ord   release@   spread  last-median  last  arrivals relative to the first")
    for i, r in enumerate(rows):
        lo, hi = min(r), max(r)
        srt = sorted(r)
        med = (srt[NW // 2 - 1] + srt[NW // 2]) // 2
        lw = r.index(hi)
        last_count[lw] += 1
        wait_for[lw] += sum(hi - v for v in r)               # wavefront-cycles the others idled at this barrier
        skew_sum += hi - med
        if run == runs - 1:
            print(f" {i:3d} {hi - t0:10d} {hi - lo:8d} {hi - med:10d}    w{lw}   " + " ".join(f"{v - lo:6d}" for v in r))
    print(f"run {run}: barriers {len(rows)}  span {span}  sum(last - median) {skew_sum} = {100.0 * skew_sum / span:.1f} % of the span;  "
          f"last-arrival counts per wavefront {last_count};  wavefront-cycles the others waited for w0..w3: {wait_for} "
          f"({100.0 * sum(wait_for) / (NW * span):.1f} % of all wavefront-cycles)")
    for w in range(NW):
        tot_last[w] += last_count[w]
        tot_wait[w] += wait_for[w]
print("all runs: last-arrival counts", tot_last, " waited-for cycles", tot_wait)
