"""Non-finite and hostile inputs through the drop-in entry points (run by tests/test_hostile.py in a subprocess, under a
timeout: a hang or a device fault is the failure being looked for).

    python tests/hostile_inputs.py <library path | "ref"> <case> <fs> <out.npz>

The reference never traps on such input (src/d4c.cpp:342-407, src/cheaptrick.cpp:191-240: no checks, garbage in / garbage
out) and some of it is undefined behaviour there (a NaN F0 becomes a window length through a double -> int conversion,
src/cheaptrick.cpp:95, src/stonemask.cpp:129).  What the drop-in promises instead: every call RETURNS, the process survives,
`temporal_positions` do not depend on the samples, and the library is unharmed -- a clean call afterwards gives the bits a
fresh process gives.  Cases:
    x_*   Harvest, Dio + StoneMask, CheapTrick and D4C on a signal with a hostile stretch
    f0_*  StoneMask, CheapTrick, D4C on a clean signal with a hostile caller-made F0 track
    clean the same calls on the clean signal (the "fresh process" answer)
"""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import sys

import numpy as np

X_CASES = ("x_nan", "x_pinf", "x_ninf", "x_huge", "x_denormal", "x_nan_run")
F0_CASES = ("f0_nan", "f0_negative", "f0_inf", "f0_huge_tiny")


def signal(fs, seconds=0.8):
    from world_amd import synth
    return np.ascontiguousarray(synth.vowel(fs, seconds, seed=4242, base_f0=150.0).numpy())


def hostile_x(case, fs):
    x = signal(fs)
    n = len(x)
    if case == "x_nan":
        x[n // 3] = np.nan
    elif case == "x_pinf":
        x[n // 2] = np.inf
    elif case == "x_ninf":
        x[n // 4] = -np.inf
    elif case == "x_huge":
        x[n // 2:n // 2 + 200] = 1e308
    elif case == "x_denormal":
        x = x * 1e-310
    elif case == "x_nan_run":
        x[n // 2:n // 2 + fs // 20] = np.nan
        x[5] = np.inf
    return x


def hostile_f0(case, f0):
    f0 = f0.copy()
    n = len(f0)
    if case == "f0_nan":
        f0[n // 5] = np.nan
        f0[n // 2:n // 2 + 6] = np.nan
        f0[-1] = np.nan
    elif case == "f0_negative":
        f0[n // 4:n // 4 + 5] = -120.0
        f0[0] = -1e300
    elif case == "f0_inf":
        f0[n // 3] = np.inf
        f0[n // 3 + 4] = -np.inf
    elif case == "f0_huge_tiny":
        f0[n // 3] = 1e300
        f0[n // 3 + 2] = 1e-300
        f0[n // 3 + 4] = 5e-324
        f0[n // 3 + 6] = 1e6
    return f0


def run(H, case, fs):
    out = {}
    fft = H.cheaptrick_fft_size(fs)
    x_clean = signal(fs)
    if case.startswith("x_") or case == "clean":
        x = hostile_x(case, fs) if case != "clean" else x_clean
        tp, f0 = H.harvest(x, fs)
        out["harvest.tp"], out["harvest.f0"] = tp, f0
        tpd, f0d = H.dio(x, fs)
        out["dio.tp"], out["dio.f0"] = tpd, f0d
        out["stonemask.f0"] = H.stonemask(x, fs, tpd, f0d)
        out["cheaptrick.sp"] = H.cheaptrick(x, fs, tp, f0, fft_size=fft)
        out["d4c.ap"] = H.d4c(x, fs, tp, f0, fft)
    else:
        tp, f0 = H.harvest(x_clean, fs)
        f0h = hostile_f0(case, f0)
        out["harvest.tp"] = tp
        out["stonemask.f0"] = H.stonemask(x_clean, fs, tp, f0h)
        out["cheaptrick.sp"] = H.cheaptrick(x_clean, fs, tp, f0h, fft_size=fft)
        out["d4c.ap"] = H.d4c(x_clean, fs, tp, f0h, fft)
    return out


def main():
    lib, case, fs, path = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    if lib == "ref":
        from oracle.loader import RefOracle
        H = RefOracle()
    else:
        from world_amd.api import HostAPI
        H = HostAPI(lib)
    arrays = {}
    for k, v in run(H, case, fs).items():
        arrays["hostile." + k] = v
    if case != "clean" and lib != "ref":
        for k, v in run(H, "clean", fs).items():              # the library after the hostile calls
            arrays["after." + k] = v
    np.savez(path, **arrays)
    print("hostile_inputs done", case)


if __name__ == "__main__":
    main()
