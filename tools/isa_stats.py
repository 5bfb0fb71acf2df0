"""Development aid (no GPU needed): compile one unit of world_amd/csrc to gfx950 assembly and count, per kernel,
the instructions by class -- FP64 arithmetic, other VALU, SALU, LDS, memory -- plus what the register allocator
added (SGPR spills through v_writelane/v_readlane, scratch traffic) and the resource lines of the metadata.
With --trace the unit is built with -DWH_TRACE and the counts are split at the cycle-stamp markers (s_memtime),
which gives instructions per PHASE of a kernel next to the cycles tools/trace.py measures for the same phases.

    python tools/isa_stats.py d4c.hip d4c_frameILi4096 [--trace] [--flags="-DD4C_MIN_WAVES=4"]
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "world_amd", "csrc")


def classify(op):
    if "f64" in op and not op.startswith(("v_cvt", "v_cmp")):
        return "fp64"
    if op.startswith("ds_"):
        return "lds"
    if op in ("v_readlane_b32", "v_writelane_b32"):
        return "sgpr_spill"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_"):
        return "salu"
    return "mem"


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    unit, pattern = args[0], (args[1] if len(args) > 1 else "")
    trace = "--trace" in sys.argv
    extra = []
    for a in sys.argv[1:]:
        if a.startswith("--flags="):
            extra += a[len("--flags="):].split()
    out = os.path.join(tempfile.gettempdir(), "isa_stats_" + unit.replace(".", "_") + ".s")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-gpu-rdc", "-I", CSRC,
           "-x", "hip", "--cuda-device-only", "-S", os.path.join(CSRC, unit), "-o", out] + (["-DWH_TRACE"] if trace else []) + extra
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    lines = open(out).read().splitlines()
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and pattern in l]
    for i0, name in starts:
        i1 = next(i for i in range(i0, len(lines)) if ".Lfunc_end" in lines[i])
        segs, cur = [], collections.Counter()
        for l in lines[i0:i1]:
            m = re.match(r"\s+([a-z_0-9]+)(\s|$)", l)
            if not m:
                continue
            if m.group(1) == "s_memtime":
                segs.append(cur)
                cur = collections.Counter()
                continue
            cur[classify(m.group(1))] += 1
            if m.group(1) == "s_barrier":
                cur["barriers"] += 1
        segs.append(cur)
        total = sum(segs, collections.Counter())
        print(name)
        meta = [l.strip() for l in lines if False]
        k = next((i for i, l in enumerate(lines) if l.strip() == ".name:           " + name or l.strip().endswith(".name: " + name)), None)
        for j, l in enumerate(lines):
            if l.strip().startswith(".name:") and l.strip().endswith(name):
                for l2 in lines[max(0, j - 25):j + 25]:
                    if re.search(r"\.(vgpr_count|sgpr_count|private_segment_fixed_size|sgpr_spill_count|vgpr_spill_count):", l2):
                        meta.append(l2.strip())
        print("  ", "  ".join(dict.fromkeys(meta)))
        print("   total", dict(total))
        if trace:
            for n, s in enumerate(segs):
                print(f"   phase {n:2d}", dict(s))


if __name__ == "__main__":
    main()
