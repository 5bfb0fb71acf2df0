"""Condense a rocprofv3 --kernel-trace of the HEADLINE mode (12 independent jobs in flight on 12 streams: the mode that
produces bench.py's `value`) into what the next optimisation is chosen from: per kernel -- launches, total and average
duration, share -- and for the whole timed span the concurrency the GPU actually saw: sum of kernel durations over the
wall span (how many kernels ran at once on average), the fraction of the span with 0 / 1 / 2-3 / >= 4 kernels resident,
launches per queue, and the gaps with nothing resident.

    python tools/trace_overlap.py <dir with *kernel_trace.csv> [out.txt]
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    return name.split("(")[0].replace("world_hip::", "").replace("void ", "")[:44]


def main():
    d = sys.argv[1]
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        sys.exit("no *kernel_trace.csv under " + d)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            try:
                def dims(prefix):
                    if prefix in r:
                        return max(1, int(r[prefix] or 1))
                    v = 1
                    for ax in ("_X", "_Y", "_Z"):
                        v *= max(1, int(r.get(prefix + ax, 1) or 1))
                    return v
                wg, grid = dims("Workgroup_Size"), dims("Grid_Size")
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?"),
                             max(1, grid // wg)))
            except (KeyError, ValueError):
                continue
    rows = [r for r in rows if not r[2].startswith(("rng_", "hv_band_spectra"))]          # set-up kernels of the first call
    rows.sort()
    # the steady part: drop the first and last 10 % of the launches (warm-up, tail)
    n = len(rows)
    rows = rows[n // 10: n - n // 10]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    span = t1 - t0
    out = []
    per = defaultdict(lambda: [0, 0])
    for s, e, k, q, _ in rows:
        per[k][0] += 1
        per[k][1] += e - s
    busy = sum(v[1] for v in per.values())
    out.append(f"{len(rows)} kernel launches over {span / 1e6:.2f} ms of steady state; sum of kernel durations {busy / 1e6:.2f} ms "
               f"= {busy / span:.2f} kernels resident on average")
    out.append(f"{'kernel':46s} {'calls':>7s} {'total_ms':>9s} {'avg_us':>8s} {'share':>6s}")
    for k, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        out.append(f"{k:46s} {c:7d} {t / 1e6:9.3f} {t / c / 1e3:8.2f} {100.0 * t / busy:6.2f}")
    # residency histogram by sweeping the start / end events
    ev = sorted([(s, 1, w) for s, e, _, _, w in rows] + [(e, -1, -w) for s, e, _, _, w in rows])
    level, width, last, hist, gaps, narrow = 0, 0, t0, defaultdict(int), [], defaultdict(int)
    for t, dlt, dw in ev:
        if t > last:
            hist[level] += t - last
            if level == 0 and t - last > 2000:
                gaps.append(t - last)
            # workgroups the resident kernels were launched with, against the 256 CUs: < 256 means that not even one
            # workgroup per CU exists among everything resident -- the chip is mostly idle whatever the kernel count says
            narrow["< 64" if width < 64 else "64-255" if width < 256 else "256-1023" if width < 1024 else ">= 1024"] += t - last
        last = t
        level += dlt
        width += dw
    buckets = {"0": hist[0], "1": hist[1], "2-3": hist[2] + hist[3], "4-7": sum(hist[i] for i in range(4, 8)),
               ">=8": sum(v for i, v in hist.items() if i >= 8)}
    out.append("fraction of the span with N kernels resident: " + ", ".join(f"{k}: {100.0 * v / span:.1f} %" for k, v in buckets.items()))
    out.append("fraction of the span by the total number of workgroups the resident kernels were launched with: "
               + ", ".join(f"{k}: {100.0 * narrow[k] / span:.1f} %" for k in ("< 64", "64-255", "256-1023", ">= 1024")))
    out.append(f"gaps with nothing resident longer than 2 us: {len(gaps)}, {sum(gaps) / 1e3:.1f} us in all ({100.0 * sum(gaps) / span:.2f} % of the span)")
    qs = defaultdict(int)
    for _, _, _, q, _ in rows:
        qs[q] += 1
    out.append(f"hardware queues used: {len(qs)}; launches per queue min / max: {min(qs.values())} / {max(qs.values())}")
    text = "\n".join(out)
    print(text)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text + "\n")


if __name__ == "__main__":
    main()
