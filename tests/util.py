"""Shared helpers for the parity tests."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# north_star tolerance: F0, spectral envelope and aperiodicity within 1e-4 relative;
# frame counts and temporal positions bit-exact.
RTOL = 1e-4


def load_golden(name):
    g = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    g["x"] = g["q"].astype(np.float64) / 32768.0
    for k in ("fs", "fft_size"):
        g[k] = int(g[k])
    for k in ("f0_floor_est", "frame_period", "q1", "threshold"):
        g[k] = float(g[k])
    g["f0_method"] = str(g["f0_method"])
    return g


def max_rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300))) if a.size else 0.0


def assert_f0_close(f0, ref, rtol=RTOL, what="f0"):
    f0 = np.asarray(f0); ref = np.asarray(ref)
    assert f0.shape == ref.shape, what
    flips = int(np.sum((f0 > 0) != (ref > 0)))
    assert flips == 0, f"{what}: {flips} voiced/unvoiced flips"
    v = ref > 0
    assert max_rel(f0[v], ref[v]) <= rtol, f"{what}: rel err {max_rel(f0[v], ref[v])}"
    assert np.all(f0[~v] == 0.0)


def analyse(backend, g):
    """Run the golden fixture's pipeline on any backend exposing the loader API."""
    x, fs = g["x"], g["fs"]
    if g["f0_method"] == "harvest":
        tp, f0 = backend.harvest(x, fs, f0_floor=g["f0_floor_est"], frame_period=g["frame_period"])
        f0_est = None
    else:
        tp, f0_est = backend.dio(x, fs, f0_floor=g["f0_floor_est"], frame_period=g["frame_period"])
        f0 = backend.stonemask(x, fs, tp, f0_est)
    sp = backend.cheaptrick(x, fs, tp, f0, q1=g["q1"], f0_floor=71.0, fft_size=g["fft_size"])
    ap = backend.d4c(x, fs, tp, f0, g["fft_size"], threshold=g["threshold"])
    return tp, f0_est, f0, sp, ap


def check_against_golden(backend, g, rtol=RTOL, given_f0=False):
    """Full-pipeline check.  With given_f0 the spectral stages are fed the golden
    F0 (isolates CheapTrick/D4C from F0 differences)."""
    x, fs = g["x"], g["fs"]
    if given_f0:
        tp, f0 = g["tp"], g["f0"]
        sp = backend.cheaptrick(x, fs, tp, f0, q1=g["q1"], f0_floor=71.0, fft_size=g["fft_size"])
        ap = backend.d4c(x, fs, tp, f0, g["fft_size"], threshold=g["threshold"])
    else:
        tp, f0_est, f0, sp, ap = analyse(backend, g)
        assert np.array_equal(tp, g["tp"]), "temporal_positions must be bit-exact"
        if f0_est is not None:
            assert_f0_close(f0_est, g["f0_dio"], rtol, "dio f0")
        assert_f0_close(f0, g["f0"], rtol)
    rows = g["rows"]
    assert sp.shape == (len(g["f0"]), g["fft_size"] // 2 + 1)
    assert max_rel(sp[rows], g["sp_rows"]) <= rtol, f"spectrogram rel err {max_rel(sp[rows], g['sp_rows'])}"
    assert max_rel(ap[rows], g["ap_rows"]) <= rtol, f"aperiodicity rel err {max_rel(ap[rows], g['ap_rows'])}"
    assert max_rel(np.log(sp).sum(axis=1), g["sp_row_sums"]) <= rtol
    assert max_rel(ap.sum(axis=1), g["ap_row_sums"]) <= rtol


def synth_params(fs, nf, fft_size, seed=0):
    """Deterministic analysis-like parameters for the synthesis tests: an f0 contour with unvoiced
    gaps, a formant-shaped envelope, a rising aperiodicity (plain numpy arithmetic only)."""
    nb = fft_size // 2 + 1
    i = np.arange(nf, dtype=np.float64)
    f0 = 130.0 + 45.0 * np.sin(2 * np.pi * i / 83.0 + seed) + 8.0 * np.sin(2 * np.pi * i / 11.0)
    f0[(i % 67) < 12] = 0.0                                   # unvoiced stretches
    f0[-5:] = 0.0
    k = np.arange(nb, dtype=np.float64) * fs / fft_size
    env = np.zeros((nf, nb))
    for c, bw, a in ((700.0, 130.0, 1.0), (1220.0, 170.0, 0.5), (2600.0, 240.0, 0.25), (3500.0, 300.0, 0.1)):
        centre = c * (1.0 + 0.1 * np.sin(2 * np.pi * i / 140.0 + seed))[:, None]
        env += a / (1.0 + ((k[None, :] - centre) / bw) ** 2)
    sp = 1e-3 * env ** 2 + 1e-9
    ap = np.clip(0.02 + 0.9 * (k[None, :] / (fs / 2.0)) ** 1.5 * (1.0 + 0.2 * np.sin(i / 9.0))[:, None], 0.0, 1.0)
    ap[f0 == 0.0] = 1.0 - 1e-12
    return f0, sp, ap
