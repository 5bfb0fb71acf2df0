#!/bin/bash
# Development aid: the two fabric-traffic PMC passes (FETCH_SIZE, WRITE_SIZE) of one bench config, per kernel --
# a fifth of tools/profile.sh's time, for A/B runs of a traffic change.
#   tools/traffic.sh <tag> <config 1..4>     (WORLD_HIP_LIB selects a library variant, tools/ab.py)
# -> gpurun_out/traffic_<tag>_c<config>.txt : MB per step and kernel (2 x FETCH_SIZE + WRITE_SIZE, MI355X_MICROARCH.md)
set -u
TAG=${1:-t}
CFG=${2:-3}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/traffic_${TAG}_c$CFG
mkdir -p $OUT/keep
cd /tmp && export TMPDIR=/tmp
if [ "$CFG" = "1" ]; then
  CMD="python $ROOT/bench.py --steps 5 --warmup 2 --streams 1 --min-wall 0 --no-cpu-baseline --no-extras --no-configs"
else
  CMD="python $ROOT/bench.py --only-config $CFG --steps 2 --min-wall 0 --contexts 1"
fi
for pmc in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/pmc_$pmc -o pmc -- $CMD > $OUT/pmc_$pmc.log 2>&1
done
python $ROOT/tools/profile_aggregate.py $OUT
python - $OUT $CFG > $OUT.txt <<'EOF'
import csv, sys, os
out, cfg = sys.argv[1], sys.argv[2]
alg = {"1": 2001 * 18336, "2": 256256 * 1936, "3": 128128 * 18336, "4": 64064 * 8864}[cfg]
def load(c):
    return {r["kernel"]: (int(r["dispatches"]), float(r["avg_per_dispatch"]))
            for r in csv.DictReader(open(os.path.join(out, "keep", f"pmc_{c}_by_kernel.csv")))}
f, w = load("FETCH_SIZE"), load("WRITE_SIZE")
ours = [k for k in f if k and not k.startswith(("at::", "rocprim", "__amd", "hipcub", "Cijk"))]
steps = max(f[k][0] for k in ours if k.startswith(("d4c_finish", "hv_detect", "hc_output", "dio_"))) if any(k.startswith(("d4c_finish", "hv_detect", "hc_output", "dio_")) for k in ours) else 1
rows = []
for k in ours:
    n, fe = f[k]
    wr = w.get(k, (0, 0.0))[1]
    rows.append(((2 * fe + wr) * 1024 * n / steps / 1e6, 2 * fe * 1024 * n / steps / 1e6, wr * 1024 * n / steps / 1e6, k))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"config {cfg}: {steps} steps; counted traffic {tot:.0f} MB per step = {tot * 1e6 / alg:.2f} x algorithmic ({alg / 1e6:.1f} MB)")
for t, fe, wr, k in rows[:24]:
    print(f"  {k:40s} {t:9.1f} MB   fetch {fe:9.1f}  write {wr:9.1f}")
EOF
rm -rf $OUT/pmc_*/ $OUT/*.log
cat $OUT.txt
