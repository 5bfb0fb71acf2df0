"""Recompute every bench leg's roofline figures from the committed rocprofv3 CSVs ALONE:
    python tools/roofline_check.py profiles/r03 > profiles/r03/roofline_check.txt
For each BASELINE config profiled (kernel_stats_config<n>.csv + pmc_*_config<n>_by_kernel.csv, one job in flight):
the dominant kernel, its average duration, algorithmic GB/s and fraction of the 8 TB/s HBM peak; its FP64 work
(SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 x 64 lanes, FMA = 2) and fraction of the 78.6 TFLOP/s vector peak; its counted
fabric traffic (2 x FETCH_SIZE + WRITE_SIZE: the gfx950 correction of MI355X_MICROARCH.md) -- and the same three for the
whole pipeline (sum over the library's kernels of one step)."""
import csv, os, sys

CFG = {"1": ("configs[1]: 1 x 48 kHz x 10 s, Harvest+CheapTrick+D4C", 2001, 18336),
       "2": ("configs[2]: 256 x 48 kHz x 5 s, Harvest only", 256256, 1936),
       "3": ("configs[3] per-GPU share: 128 x 48 kHz x 5 s, Harvest+CheapTrick+D4C", 128128, 18336),
       "4": ("configs[4]: 64 x 16 kHz x 5 s, DIO+StoneMask+CheapTrick+D4C", 64064, 8864)}
HBM, FP64 = 8000.0, 78.6
d = sys.argv[1] if len(sys.argv) > 1 else "profiles/r03"


def short(n):
    return n.split("(")[0].replace("world_hip::", "").replace("void ", "").strip()


def ours(n):
    return not n.startswith(("at::", "rocprim", "__amd", "hipcub")) and n != "" and not n.startswith("Cijk")


for c, (title, frames, bpf) in CFG.items():
    f = os.path.join(d, f"kernel_stats_config{c}.csv")
    if not os.path.exists(f):
        continue
    stats = {short(r["Name"]): r for r in csv.DictReader(open(f))
             if "world_hip" in r["Name"] and not short(r["Name"]).startswith(("mp_", "rng_", "hv_band_spectra"))}    # (not the microprobe, not one-time set-up)
    pmc = {}
    for fn in os.listdir(d):
        if fn.startswith("pmc_") and fn.endswith(f"_config{c}_by_kernel.csv"):
            for r in csv.DictReader(open(os.path.join(d, fn))):
                pmc.setdefault(r["kernel"], {})[r["counter"]] = (float(r["avg_per_dispatch"]), int(r["dispatches"]))
    steps = None
    dom = max(stats, key=lambda k: float(stats[k]["TotalDurationNs"]))
    # steps profiled = launches of a once-per-step kernel
    once = [k for k in stats if k.startswith(("d4c_finish", "hv_detect", "hc_output", "dio_"))]
    steps = int(stats[once[0]]["Calls"]) if once else int(stats[dom]["Calls"])
    alg = frames * bpf
    avg_us = float(stats[dom]["AverageNs"]) / 1e3
    print(f"== {title}  ({frames} frames per step, {bpf} algorithmic B per frame = {alg / 1e6:.2f} MB per step; {steps} steps profiled)")
    print(f"   dominant kernel {dom}: {int(stats[dom]['Calls'])} launches, avg {avg_us:.2f} us")
    ach = alg / (avg_us * 1e-6) / 1e9
    print(f"   roofline (hbm): achieved {ach:.1f} GB/s of {HBM:.0f} = frac {ach / HBM:.4f}")

    def cnt(k, name):
        for kk, v in pmc.items():
            if kk == k or kk.startswith(k.split("<")[0]) and k.split("<")[0] == kk.split("<")[0]:
                if name in v:
                    return v[name][0]
        return None
    def flop(k):
        a, m, fm, t = (cnt(k, "SQ_INSTS_VALU_" + x + "_F64") for x in ("ADD", "MUL", "FMA", "TRANS"))
        if None in (a, m, fm):
            return None
        return 64.0 * (a + m + 2.0 * fm + (t or 0.0))
    fl = flop(dom)
    if fl:
        tf = fl / (avg_us * 1e-6) / 1e12
        print(f"   fp64: {fl / 1e9:.2f} GFLOP per launch -> {tf:.2f} TFLOP/s = {100 * tf / FP64:.1f} % of {FP64}")
    fe, wr = cnt(dom, "FETCH_SIZE"), cnt(dom, "WRITE_SIZE")
    if fe is not None and wr is not None:
        tr = (2.0 * fe + wr) * 1024.0
        print(f"   traffic: 2 x FETCH_SIZE {2 * fe * 1024 / 1e6:.1f} MB + WRITE_SIZE {wr * 1024 / 1e6:.1f} MB = {tr / 1e6:.1f} MB per launch "
              f"({tr / alg:.2f} x the pipeline's algorithmic bytes)")
    tot_t = tot_f = tot_ms = 0.0
    for k, r in stats.items():
        per_step = int(r["Calls"]) / steps
        tot_ms += float(r["TotalDurationNs"]) / steps / 1e6
        fe, wr = cnt(k, "FETCH_SIZE"), cnt(k, "WRITE_SIZE")
        if fe is not None and wr is not None:
            tot_t += (2.0 * fe + wr) * 1024.0 * per_step
        fl = flop(k)
        if fl:
            tot_f += fl * per_step
    print(f"   whole pipeline: sum of kernel durations {tot_ms:.3f} ms per step -> {frames / tot_ms / 1e3:.3f} M frames/s with one job in flight; "
          f"{tot_f / frames / 1e6:.2f} MFLOP per frame = {tot_f / (tot_ms * 1e-3) / 1e12:.2f} TFLOP/s ({100 * tot_f / (tot_ms * 1e-3) / 1e12 / FP64:.1f} %); "
          f"counted traffic {tot_t / 1e6:.0f} MB per step = {tot_t / alg:.1f} x algorithmic")
    top = sorted(stats, key=lambda k: -float(stats[k]["TotalDurationNs"]))[:8]
    for k in top:
        fe, wr = cnt(k, "FETCH_SIZE"), cnt(k, "WRITE_SIZE")
        lc, la = cnt(k, "SQ_LDS_BANK_CONFLICT"), cnt(k, "SQ_LDS_IDX_ACTIVE")
        print(f"      {k:34s} {float(stats[k]['TotalDurationNs']) / steps / 1e6:8.3f} ms/step"
              + (f"  traffic {(2 * fe + wr) * 1024 * int(stats[k]['Calls']) / steps / 1e6:9.1f} MB" if fe is not None and wr is not None else "")
              + (f"  lds conflict ratio {lc / la:.2f}" if lc is not None and la else ""))
    print()
