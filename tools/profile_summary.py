"""Condense rocprofv3 output (kernel stats + PMC counter CSVs) into one text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = sys.argv[1]


def short(name):
    name = name.split("(")[0]
    return name.replace("world_hip::", "").replace("void ", "")[:48]


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r.get("TotalDurationNs", 0) or 0))
    print(f"{'kernel':50s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>6s}")
    for r in rows[:40]:
        print(f"{short(r['Name']):50s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/1e6:10.3f} "
              f"{float(r['AverageNs'])/1e3:10.2f} {float(r['Percentage']):6.2f}")

import json
traffic = {}
print()
print("== PMC counters, per-dispatch average by kernel ==")
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            a = acc[k][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    print(f"-- {os.path.basename(d)}")
    for k in sorted(acc, key=lambda k: -sum(v[0] for v in acc[k].values()))[:12]:
        vals = "  ".join(f"{c}={v[0]/max(v[1],1):.4g}" for c, v in sorted(acc[k].items()))
        n = max(v[1] for v in acc[k].values())
        print(f"   {k:48s} n={n:<5d} {vals}")
    for k in acc:
        for c, v in acc[k].items():
            if c in ("FETCH_SIZE", "WRITE_SIZE") or c.endswith("_F64"):
                traffic.setdefault(k, {})[c] = v[0] / max(v[1], 1)
import bench
json.dump({"csrc_hash": bench.csrc_hash(), "config": os.environ.get("PROFILE_CONFIG", "1"),
           "frames_per_launch": int(os.environ.get("PROFILE_FRAMES", "2001")), "unit": "FETCH_SIZE / WRITE_SIZE: KB per dispatch; *_F64: wave-level instructions per dispatch (rocprofv3)",
           "kernels": traffic}, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
