"""Per-kernel averages of every rocprofv3 PMC pass under <dir>/pmc_*/ -> <dir>/keep/pmc_*_by_kernel.csv
(kernel, counter, dispatches, average per dispatch): what profiles/<round>/ keeps instead of the per-dispatch rows."""
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("world_hip::", "").replace("void ", "")[:60]
            a = acc[(k, r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
    with open(os.path.join(out, "keep", os.path.basename(d) + "_by_kernel.csv"), "w") as o:
        o.write("kernel,counter,dispatches,avg_per_dispatch\n")
        for (k, c), (s, n) in sorted(acc.items()):
            o.write(f'"{k}",{c},{n},{s / n:.6g}\n')
