"""Copy the condensed rocprofv3 evidence of gpurun_out/prof_<tag>/ into profiles/<round>/ (tracked):
kernel stats, the text summary, per-kernel averages of every PMC pass; and merge the run's PMC traffic /
FP64 counters into profiles/pmc_traffic.json under its BASELINE config, stamped with the hash of the kernel
sources they were measured on (bench.py reports `traffic_stale` when the tree has moved on).
    python tools/profile_install.py <tag> <round> [suffix]"""
import csv, glob, json, os, shutil, sys
from collections import defaultdict
tag, rnd = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "r02")
suffix = sys.argv[3] if len(sys.argv) > 3 else ""
src, dst = f"gpurun_out/prof_{tag}", f"profiles/{rnd}"
os.makedirs(dst, exist_ok=True)
for f in glob.glob(src + "/keep/pmc_*_by_kernel.csv"):
    shutil.copy(f, os.path.join(dst, os.path.basename(f).replace("_by_kernel.csv", suffix + "_by_kernel.csv")))
shutil.copy(src + "/keep/trace_kernel_stats.csv", dst + f"/kernel_stats{suffix}.csv")
shutil.copy(src + "/summary.txt", dst + f"/rocprofv3_summary{suffix}.txt")
new = json.load(open(src + "/pmc_traffic.json"))
path = "profiles/pmc_traffic.json"
try:
    old = json.load(open(path))
except (OSError, ValueError):
    old = {}
if "configs" not in old or old.get("csrc_hash") != new["csrc_hash"]:
    old = {"csrc_hash": new["csrc_hash"], "unit": new["unit"], "configs": {}}   # counters of another tree are dropped, not mixed
old["configs"][str(new.get("config", "1"))] = {"frames_per_launch": new["frames_per_launch"], "kernels": new["kernels"],
                                               "source": f"{dst}/rocprofv3_summary{suffix}.txt"}
json.dump(old, open(path, "w"), indent=1)
print(open(dst + f"/rocprofv3_summary{suffix}.txt").read().split("\n== PMC")[0][:1600])
