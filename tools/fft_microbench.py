"""csrc/fft.h in isolation: per transform length, GB/s of the block-cooperative
real FFT straight from and to HBM through the probe entry points (include/world_hip.h: world_hip_probe_rfft / _irfft).

    python tools/fft_microbench.py [--out profiles/r02/fft_microbench.json] [--pmc-csv counter_collection.csv]

Traffic model per transform: 8 N bytes in + 16 (N/2 + 1) bytes out, 2.5 N log2 N flop.
With --pmc-csv (the counter_collection.csv of `rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
-- python tools/fft_microbench.py --reps 1`) the LDS bank-conflict ratio of every (length, plan) is added: dispatches are
told apart by kernel name (plan) and launch shape (length)."""
import argparse
import csv
import json
import math
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def lds_bytes(lg):                       # fft_probe_lds_bytes(): N doubles + quarter-wave table of the inner transform
    n = 1 << lg
    return 8 * (n + (1 << (lg - 3)) + 2)


def conflicts(path):
    acc = defaultdict(lambda: defaultdict(float))
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        if "fft_probe" not in name:
            continue
        # dynamic LDS is not reported: the length follows from the launch shape (batch = 2^26 / N workgroups)
        lg = 26 - int(round(math.log2(int(r["Grid_Size"]) / int(r["Workgroup_Size"]))))
        key = (("irfft" if "irfft" in name else "rfft"), name.split("<")[1].split(">")[0].replace(" ", ""), lg)
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
    return acc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--pmc-csv", default="")
    a = ap.parse_args()
    import torch
    from world_amd.api import WorldHip
    wh = WorldHip()
    pmc = conflicts(a.pmc_csv) if a.pmc_csv else {}
    rows = []
    for lg in range(8, 14):
        n = 1 << lg
        batch = max(4096, (1 << 26) // n)                    # 512 MB of input: far beyond the Infinity Cache
        x = torch.randn((batch, n), dtype=torch.float64, device="cuda")
        spec = torch.empty((batch, n // 2 + 1, 2), dtype=torch.float64, device="cuda")
        y = torch.empty_like(x)
        for max_lr in (3, 4):
            for static in ((False, True) if (max_lr == 3 and 10 <= lg <= 12) else (False,)):
                rec = {"n": n, "plan": "radix-8" if max_lr == 3 else "radix-16", "length_known_at_compile_time": static,
                       "batch": batch, "threads": max(64, n >> (max_lr + 1))}
                for direction, fn, args in (("rfft", wh.probe_rfft, (x,)), ("irfft", wh.probe_irfft, (spec,))):
                    out = spec if direction == "rfft" else y
                    fn(*args, max_lr=max_lr, out=out, static_plan=static)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(a.reps):
                        fn(*args, max_lr=max_lr, out=out, static_plan=static)
                    e1.record()
                    torch.cuda.synchronize()
                    ms = e0.elapsed_time(e1) / a.reps
                    nbytes = batch * (8 * n + 16 * (n // 2 + 1))
                    rec[direction] = {"ms": ms, "GB_per_s": nbytes / ms / 1e6, "frac_of_8TB_s": nbytes / ms / 1e6 / 8000.0,
                                      "TFLOP_per_s": batch * 2.5 * n * lg / ms / 1e9,
                                      "transforms_per_s": batch / ms * 1e3}
                    tmpl = f"{max_lr},{lg if static else 0}"
                    c = pmc.get((direction, tmpl, lg))
                    if c and c.get("SQ_LDS_IDX_ACTIVE"):
                        rec[direction]["lds_bank_conflict_ratio"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]
                rows.append(rec)
                print(json.dumps(rec), flush=True)
        del x, spec, y
    if a.out:
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        json.dump({"model": "bytes = batch * (8 N + 16 (N/2+1)), flop = batch * 2.5 N log2 N; HIP events around `reps` launches",
                   "hbm_peak_GB_per_s": 8000.0, "rows": rows}, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
