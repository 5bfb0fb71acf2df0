"""Copy the condensed rocprofv3 evidence of gpurun_out/prof_<tag>/ into profiles/ (tracked):
kernel stats, the text summary, per-kernel averages of every PMC pass, pmc_traffic.json."""
import csv, glob, os, shutil, sys
from collections import defaultdict
tag, rnd = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "r01")
src, dst = f"gpurun_out/prof_{tag}", f"profiles/{rnd}"
os.makedirs(dst, exist_ok=True)
for f in glob.glob(src + "/keep/pmc_*.csv"):
    acc = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("world_hip::", "").replace("void ", "")[:60]
        a = acc[(k, r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
    with open(os.path.join(dst, os.path.basename(f)[:-4] + "_by_kernel.csv"), "w") as o:
        o.write("kernel,counter,dispatches,avg_per_dispatch\n")
        for (k, c), (s, n) in sorted(acc.items()):
            o.write(f'"{k}",{c},{n},{s / n:.6g}\n')
shutil.copy(src + "/keep/trace_kernel_stats.csv", dst + "/kernel_stats.csv")
shutil.copy(src + "/summary.txt", dst + "/rocprofv3_summary.txt")
shutil.copy(src + "/pmc_traffic.json", "profiles/pmc_traffic.json")
print(open(dst + "/rocprofv3_summary.txt").read().split("\n== PMC")[0][:1600])
