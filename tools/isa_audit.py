"""Development aid (no GPU needed): compile the kernel sources to gfx950 assembly and flag the two code shapes that cost
round 3 the most lone-job latency:

  1. chained load-wait pairs -- a global load, an `s_waitcnt vmcnt(0)` within a few instructions, then the next load:
     items that should be in flight together queue up one trip to memory each.  Typical sources: a load under a
     lane-varying condition (`cond ? p[i] : 0`: the value is waited for where the branch rejoins), a prefetch under `if`
     (the wait-count bookkeeping gives up at the join), a loop `dst[i] = f(src[i])` the compiler may not reorder;
  2. innermost loops that wait for a load they issued in the same trip with at most three loads in flight (harmless for
     short trip counts -- twiddle tables -- and a trip to memory per iteration otherwise);
  3. the same for LDS: innermost loops that wait for the one or two ds_reads of the trip (waiting_lds_loops).

    python tools/isa_audit.py [unit ...]        # default: every .hip under world_amd/csrc
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "world_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I", SRC, "-S", "--cuda-device-only"]


def demangle(name):
    return subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()[:88]


def assembly(unit, out_dir):
    out = os.path.join(out_dir, unit + ".s")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    r = subprocess.run([hipcc, *FLAGS, "-o", out, os.path.join(SRC, unit + ".hip")], capture_output=True, text=True)
    if r.returncode != 0:
        raise SystemExit(r.stderr[-2000:])
    return [l for l in open(out).read().split("\n") if l.strip() and not l.strip().startswith(";")]


def chained_pairs(lines):
    found, kern, run, start, last_load = {}, None, 0, None, None
    for n, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            kern, run, last_load = m.group(1), 0, None
        if re.search(r"\b(global_load|flat_load|buffer_load)", l):
            last_load = n
        if re.search(r"s_waitcnt.*vmcnt\(0\)", l) and last_load is not None and n - last_load <= 6:
            run = run + 1 if start is not None and n - start <= 40 else 1
            start, last_load = n, None
            if run >= 3:
                found[kern] = max(found.get(kern, 0), run)
    return found


def waiting_loops(lines):
    found, kern = [], None
    for n, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            kern = m.group(1)
        m = re.match(r"^(\.LBB\d+_\d+):.*Inner Loop Header", l)
        if not m:
            continue
        lab, end = m.group(1), None
        for k in range(n + 1, min(n + 4000, len(lines))):
            if re.search(r"s_c?branch\w*\s+" + re.escape(lab) + r"\b", lines[k]):
                end = k
                break
            if re.match(r"^_Z\w+:", lines[k]):
                break
        if end is None:
            continue
        body = lines[n:end + 1]
        loads = sum(1 for b in body if re.search(r"\b(global_load|flat_load|buffer_load)", b))
        waits = sum(1 for b in body if re.search(r"s_waitcnt.*vmcnt\(0\)", b))
        if loads and waits and loads <= 3:
            found.append((kern, lab, len(body), loads))
    return found


def waiting_lds_loops(lines):
    """innermost loops that read LDS and wait for it (lgkmcnt(0)) every trip with at most two reads in flight: a trip
    through the LDS pipe per iteration -- ~100 cycles on an idle CU, ~750 in one whose other workgroups run transforms
    (tools/trace_batch.py).  Round 4 found hv_band_events_fft's mirror-store loop this way (10.5 k of a block's 47 k cycles)."""
    found, kern = [], None
    for n, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            kern = m.group(1)
        m = re.match(r"^(\.LBB\d+_\d+):.*Inner Loop Header", l)
        if not m:
            continue
        lab, end = m.group(1), None
        for k in range(n + 1, min(n + 4000, len(lines))):
            if re.search(r"s_c?branch\w*\s+" + re.escape(lab) + r"\b", lines[k]):
                end = k
                break
            if re.match(r"^_Z\w+:", lines[k]):
                break
        if end is None:
            continue
        body = lines[n:end + 1]
        reads = sum(1 for b in body if re.search(r"\bds_read", b))
        waits = sum(1 for b in body if re.search(r"s_waitcnt.*lgkmcnt\(0\)", b))
        if reads and waits and reads <= 2:
            found.append((kern, lab, len(body), reads))
    return found


def main():
    units = sys.argv[1:] or sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(SRC, "*.hip")))
    with tempfile.TemporaryDirectory() as tmp:
        for unit in units:
            lines = assembly(unit, tmp)
            for kern, run in chained_pairs(lines).items():
                print(f"{unit:18s} {demangle(kern):88s} chained load-wait pairs: {run}")
            for kern, lab, size, loads in waiting_loops(lines):
                print(f"{unit:18s} {demangle(kern):88s} loop {lab} ({size} instructions) waits for its {loads} load(s) every trip")
            for kern, lab, size, reads in waiting_lds_loops(lines):
                print(f"{unit:18s} {demangle(kern):88s} loop {lab} ({size} instructions) waits for its {reads} LDS read(s) every trip")


if __name__ == "__main__":
    main()
