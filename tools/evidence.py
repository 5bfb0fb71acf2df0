"""The round's rocprofv3 evidence in ONE gpurun call (tools/round_evidence.sh drives this on the GPU box).

    python tools/evidence.py plan [--all]            which PMC passes are needed: only kernels of units whose sources changed
                                                     since profiles/pmc_traffic.json was stamped (per-unit hashes) -> plan.json
    python tools/evidence.py install <dir> <round>   condense the passes under <dir> into profiles/<round>/ and merge the
                                                     counters into profiles/pmc_traffic.json (kernel by kernel, each stamped
                                                     with the hash of its unit; counters of unchanged units are kept)

A one-line change to one .hip unit therefore costs that unit's kernels a re-profile, not the tree's (VERDICT r04 item 8:
finished patches were parked because "every change needs a re-profile")."""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

PMC_PATH = os.path.join(ROOT, "profiles", "pmc_traffic.json")
FRAMES = {"1": 2001, "2": 256256, "3": 128128, "4": 64064}
# which units' kernels run in which BASELINE config's leg
CONFIG_UNITS = {"1": {"harvest.hip", "harvest_contour.hip", "cheaptrick.hip", "d4c.hip"},
                "2": {"harvest.hip", "harvest_contour.hip"},
                "3": {"harvest.hip", "harvest_contour.hip", "cheaptrick.hip", "d4c.hip"},
                "4": {"dio.hip", "stonemask.hip", "cheaptrick.hip", "d4c.hip"}}


def short(name):
    return name.split("(")[0].replace("world_hip::", "").replace("void ", "").strip()


def load():
    try:
        with open(PMC_PATH) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def plan(everything):
    units, kernels = bench.unit_hashes()
    old = load()
    stamped = old.get("unit_hashes", {})
    changed = sorted(u for u in units if everything or stamped.get(u) != units[u])
    configs = [c for c in sorted(CONFIG_UNITS) if CONFIG_UNITS[c] & set(changed) or c not in old.get("configs", {})]
    names = sorted(k for k, u in kernels.items() if u in changed)
    # rocprofv3 --kernel-include-regex: the kernels of the changed units only (instrumenting fewer kernels also runs faster)
    regex = "|".join(names) if names and len(changed) < len(units) else ""
    out = {"changed_units": changed, "configs": configs, "kernel_regex": regex}
    print(json.dumps(out))
    return out


def prune(table, units, kernels):
    """drop what can no longer be looked up honestly: entries of library kernels that the sources no longer have (hc_base,
    d4c_prepare2 ...), and older spellings of a kernel whose template arguments changed ("ct_frame<8, 11, 256>" beside
    "ct_frame<8, 11, 256, false>": bench.py summed both into the pipeline's flop per frame once) -- a base name keeps its
    entries with the CURRENT unit hash only, if it has any"""
    every_base = set()
    for entry in table.get("configs", {}).values():
        for name, e in entry.get("kernels", {}).items():
            if "unit_hash" in e:
                every_base.add(name.split("<")[0])
    for entry in table.get("configs", {}).values():
        ks = entry.get("kernels", {})
        for name in list(ks):
            e, b = ks[name], name.split("<")[0]
            if "unit_hash" not in e:
                if b.startswith(("hv_", "hc_", "ct_", "d4c_", "dio_", "sm_", "rng_", "sy_", "codec_")) and b not in kernels:
                    del ks[name]                                # a library kernel of an earlier round, never re-stamped
                continue
            if b not in kernels:
                del ks[name]
                continue
            fresh = [n for n in ks if n.split("<")[0] == b and ks[n].get("unit_hash") == units[kernels[b]]]
            if fresh and name not in fresh:
                del ks[name]


def install(src, rnd):
    dst = os.path.join(ROOT, "profiles", rnd)
    os.makedirs(dst, exist_ok=True)
    units, kernels = bench.unit_hashes()
    old = load()
    if "configs" not in old:
        old = {"configs": {}}
    old["unit"] = "FETCH_SIZE / WRITE_SIZE: KB per dispatch; *_F64: wave-level instructions per dispatch (rocprofv3)"
    for cfg in sorted(FRAMES):
        base = os.path.join(src, f"config{cfg}")
        if not os.path.isdir(base):
            continue
        for f in glob.glob(os.path.join(base, "stats", "**", "*kernel_stats.csv"), recursive=True):
            shutil.copy(f, os.path.join(dst, f"kernel_stats_config{cfg}.csv"))
        entry = old["configs"].setdefault(cfg, {"frames_per_launch": FRAMES[cfg], "kernels": {}})
        entry["frames_per_launch"] = FRAMES[cfg]
        entry["source"] = f"profiles/{rnd}/pmc_*_config{cfg}_by_kernel.csv"
        for d in sorted(glob.glob(os.path.join(base, "pmc_*"))):
            acc = defaultdict(lambda: [0.0, 0])
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    a = acc[(short(r["Kernel_Name"])[:60], r["Counter_Name"])]
                    a[0] += float(r["Counter_Value"]); a[1] += 1
            if not acc:
                continue
            with open(os.path.join(dst, f"{os.path.basename(d)}_config{cfg}_by_kernel.csv"), "w") as o:
                o.write("kernel,counter,dispatches,avg_per_dispatch\n")
                for (k, c), (s, n) in sorted(acc.items()):
                    o.write(f'"{k}",{c},{n},{s / n:.6g}\n')
            for (k, c), (s, n) in acc.items():
                if c in ("FETCH_SIZE", "WRITE_SIZE") or c.endswith("_F64"):
                    kk = k[:48]
                    e = entry["kernels"].setdefault(kk, {})
                    e[c] = s / n
                    b = kk.split("<")[0]
                    if b in kernels:
                        e["unit_hash"] = units[kernels[b]]
    prune(old, units, kernels)
    old["csrc_hash"] = bench.csrc_hash()
    old["unit_hashes"] = units
    with open(PMC_PATH, "w") as f:
        json.dump(old, f, indent=1)
    print("installed", sorted(old["configs"]), "->", dst)


if __name__ == "__main__":
    if sys.argv[1] == "plan":
        plan("--all" in sys.argv)
    elif sys.argv[1] == "install":
        install(sys.argv[2], sys.argv[3])
