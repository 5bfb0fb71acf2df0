"""Parity at BASELINE.json's sizes and in the bench's timed mode, all against the
unmodified reference (oracle/_ref, test/test.cpp:89-219 plumbing):

  configs[1]  48 kHz x 10 s, Harvest + CheapTrick + D4C                      -- every frame checked
  configs[2]  256 x (48 kHz, 5 s), Harvest only, one batched call            -- 16 utterances checked in full
  configs[3]  per-GPU share, 128 x (48 kHz, 5 s), full pipeline              -- 8 utterances checked in full
  configs[4]  64 x (16 kHz, 5 s), DIO + StoneMask + CheapTrick + D4C         -- 8 utterances checked in full
  the bench's mode: 8 contexts on 8 streams concurrently                     -- bit-identical to serial runs
  randn table growth under interleaved CheapTrick / D4C calls on 2 contexts  -- D4C bit-stable, table intact
  slices of the randomised sweeps (tests/fuzz_*.py) and one 60 s utterance

Tolerance (north_star): frame counts / temporal positions bit-exact, no voiced/unvoiced flip; F0, spectral
envelope and aperiodicity within 1e-4 relative.  Reference calls run on a few host threads (the reference is
re-entrant; ctypes releases the GIL)."""
import json
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from util import RTOL, assert_f0_close, max_rel

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
FS = 48000


@pytest.fixture(scope="module")
def wh():
    from world_amd.api import WorldHip
    return WorldHip()


@pytest.fixture(scope="module")
def ref():
    from oracle.loader import best_oracle
    return best_oracle()


@pytest.fixture(scope="module")
def pool():
    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as p:
        yield p


def _ref_full(ref, x, fs, fft):
    tp, f0 = ref.harvest(x, fs)
    return tp, f0, ref.cheaptrick(x, fs, tp, f0, fft_size=fft), ref.d4c(x, fs, tp, f0, fft)


def _check_full(tag, got, want):
    (tp, f0, sp, ap), (tp_r, f0_r, sp_r, ap_r) = got, want
    n = len(f0_r)
    assert np.array_equal(tp[:n], tp_r), tag + ": temporal positions must be bit-exact"
    assert_f0_close(f0[:n], f0_r, what=tag + " f0")
    assert max_rel(sp[:n], sp_r) <= RTOL, f"{tag}: spectrogram rel err {max_rel(sp[:n], sp_r)}"
    assert max_rel(ap[:n], ap_r) <= RTOL, f"{tag}: aperiodicity rel err {max_rel(ap[:n], ap_r)}"


def _lone_context():
    from world_amd.api import WorldHip
    return WorldHip()


def test_config1_full_size_against_the_reference(wh, ref, pool):
    """BASELINE configs[1] at the size the metric is quoted on: 480 000 samples, 2001 frames, every frame"""
    import torch
    from world_amd import synth
    x = synth.vowel(FS, 10.0, seed=12345)
    job = pool.submit(_ref_full, ref, x.numpy(), FS, 2048)
    tpos, f0, sp, ap, nf = wh.analyze(x.cuda().unsqueeze(0), FS)
    torch.cuda.synchronize()
    assert int(nf[0]) == 2001
    _check_full("configs[1]", (tpos[0].cpu().numpy(), f0[0].cpu().numpy(), sp[0].cpu().numpy(), ap[0].cpu().numpy()), job.result())


def test_config2_harvest_batch_of_256(wh, ref, pool):
    import torch
    from world_amd import synth
    B = 256
    xs = [synth.utterance(i, FS, 5.0, device="cuda") for i in range(B)]
    picks = list(range(3, B, 16))                                 # 16 utterances, vowels and chirps
    jobs = {i: pool.submit(ref.harvest, xs[i].cpu().numpy(), FS) for i in picks}
    tpos, f0, nf = wh.harvest(torch.stack(xs).contiguous(), FS)
    torch.cuda.synchronize()
    assert np.all(nf == 1001) and tpos.shape == (B, 1001)
    assert torch.equal(tpos, tpos[0].expand_as(tpos))            # same length => the same time axis everywhere
    assert bool(torch.isfinite(f0).all()) and float(f0.min()) >= 0.0
    voiced_share = float((f0 > 0).double().mean())
    assert 0.5 < voiced_share < 1.0, voiced_share
    for i in picks:
        tp_r, f0_r = jobs[i].result()
        assert np.array_equal(tpos[i].cpu().numpy(), tp_r)
        assert_f0_close(f0[i].cpu().numpy(), f0_r, what=f"configs[2] utterance {i}")
    # ... and EVERY utterance of the batch is bit-identical to its own single call: with the reference-checked ones above
    # that covers the whole batch (batched == single is the library's invariant)
    lone = _lone_context()
    for i in range(B):
        tp1, f01, _ = lone.harvest(xs[i][None].contiguous(), FS)
        assert torch.equal(f01[0], f0[i]) and torch.equal(tp1[0], tpos[i]), f"configs[2] utterance {i}: batched != single"
    lone.close()


def test_config3_per_gpu_share_128(wh, ref, pool):
    import torch
    from world_amd import synth
    B = 128
    xs = [synth.utterance(i, FS, 5.0, device="cuda") for i in range(B)]
    picks = [0, 1, 30, 47, 64, 93, 110, 127]
    jobs = {i: pool.submit(_ref_full, ref, xs[i].cpu().numpy(), FS, 2048) for i in picks}
    tpos, f0, sp, ap, nf = wh.analyze(torch.stack(xs).contiguous(), FS)
    torch.cuda.synchronize()
    assert np.all(nf == 1001)
    assert bool(torch.isfinite(sp).all()) and bool((sp > 0).all()) and bool(((ap > 0) & (ap <= 1)).all())
    for i in picks:
        _check_full(f"configs[3] utterance {i}", (tpos[i].cpu().numpy(), f0[i].cpu().numpy(), sp[i].cpu().numpy(),
                                                  ap[i].cpu().numpy()), jobs[i].result())
    lone = _lone_context()                                        # the whole batch: every utterance == its own single call
    for i in range(B):
        tp1, f01, sp1, ap1, _ = lone.analyze(xs[i][None].contiguous(), FS)
        assert torch.equal(f01[0], f0[i]) and torch.equal(sp1[0], sp[i]) and torch.equal(ap1[0], ap[i]), \
            f"configs[3] utterance {i}: batched != single"
    lone.close()
    wh.close()                                                    # the batch's workspace back to the pool


def test_config4_dio_path_64_x_16k(wh, ref, pool):
    import torch
    from world_amd import synth
    B, fs = 64, 16000
    xs = [synth.vowel(fs, 5.0, seed=100 + i, base_f0=90.0 + (i % 32) * 8.0, device="cuda") for i in range(B)]
    picks = [0, 9, 18, 27, 36, 45, 54, 63]

    def ref_dio_path(x):
        tp, f0_raw = ref.dio(x, fs)
        f0 = ref.stonemask(x, fs, tp, f0_raw)
        return tp, f0, ref.cheaptrick(x, fs, tp, f0, q1=-0.15, fft_size=1024), ref.d4c(x, fs, tp, f0, 1024, threshold=0.85)
    jobs = {i: pool.submit(ref_dio_path, xs[i].cpu().numpy()) for i in picks}
    tpos, f0, sp, ap, nf = wh.analyze(torch.stack(xs).contiguous(), fs, f0_method="dio", q1=-0.15, threshold=0.85)
    torch.cuda.synchronize()
    assert np.all(nf == 1001) and sp.shape[-1] == 513
    for i in picks:
        _check_full(f"configs[4] utterance {i}", (tpos[i].cpu().numpy(), f0[i].cpu().numpy(), sp[i].cpu().numpy(),
                                                  ap[i].cpu().numpy()), jobs[i].result())
    lone = _lone_context()                                        # the whole batch: every utterance == its own single call
    for i in range(B):
        tp1, f01, sp1, ap1, _ = lone.analyze(xs[i][None].contiguous(), fs, f0_method="dio", q1=-0.15, threshold=0.85)
        assert torch.equal(f01[0], f0[i]) and torch.equal(sp1[0], sp[i]) and torch.equal(ap1[0], ap[i]), \
            f"configs[4] utterance {i}: batched != single"
    lone.close()


def test_eight_contexts_on_eight_streams_equal_serial_runs():
    """bench.py's timed mode: 8 library contexts, one HIP stream each, jobs interleaved from one host thread with
    nothing synchronised in between.  Every slot gets its OWN utterance; after three rounds every slot's outputs
    must be bit-identical to a serial analysis of that utterance on a fresh context."""
    import torch
    from world_amd import synth
    from world_amd.api import WorldHip
    S, seconds = 8, 3.0
    xs = [(synth.vowel(FS, seconds, seed=12345) if k == 0 else synth.utterance(k, FS, seconds)).cuda().unsqueeze(0) for k in range(S)]
    serial = []
    for k in range(S):
        w = WorldHip()
        serial.append(w.analyze(xs[k], FS))
        torch.cuda.synchronize()
        w.close()
    streams = [torch.cuda.Stream() for _ in range(S)]
    whs = [WorldHip() for _ in range(S)]
    out = [None] * S
    for _ in range(3):
        for k in range(S):
            with torch.cuda.stream(streams[k]):
                out[k] = whs[k].analyze(xs[k], FS)
    torch.cuda.synchronize()
    for k in range(S):
        for name, a, b in zip(("tpos", "f0", "sp", "ap"), out[k][:4], serial[k][:4]):
            assert torch.equal(a, b), f"slot {k}: {name} differs from the serial run by {float((a - b).abs().max()):.3e}"
    assert all(w.verify_tables() for w in whs[:1])
    for w in whs:
        w.close()


def test_one_worldhip_keeps_a_context_per_stream():
    """alternating torch streams on ONE WorldHip must not destroy / recreate its context"""
    import torch
    from world_amd import synth
    from world_amd.api import WorldHip
    w = WorldHip()
    x = synth.vowel(16000, 0.5, seed=4).cuda().unsqueeze(0)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    for _ in range(2):
        for s in (s1, s2):
            with torch.cuda.stream(s):
                outs.append(w.analyze(x, 16000))
    torch.cuda.synchronize()
    assert len(w._ctxs) == 2
    for o in outs[1:]:
        assert all(torch.equal(a, b) for a, b in zip(o[:4], outs[0][:4]))
    w.close()


def test_randn_table_growth_stress():
    """the shared randn table is rebuilt (and verified word for word) while two contexts
    interleave CheapTrick / D4C calls on their own streams; D4C of a probe utterance must be bit-stable through
    every generation of the table and over 50 repetitions.  Runs in its own process: the table is per process."""
    r = subprocess.run([sys.executable, os.path.join(HERE, "randn_table_stress.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert rep["generations"] >= 4 and rep["d4c_repetitions"] >= 50
    assert rep["d4c_bit_stable"] and rep["cheaptrick_bit_stable"] and rep["table_intact"], rep
    assert rep["d4c_vs_reference"] <= RTOL and rep["cheaptrick_vs_reference"] <= RTOL, rep


def test_randomised_parity_sweep_slice(ref):
    """100 cases of tests/fuzz_parity.py: eight sampling rates, nine signal kinds, random options, every entry
    point (Harvest, DIO, StoneMask, CheapTrick, D4C, Synthesis, coders) against the reference at 1e-6"""
    import fuzz_parity
    from world_amd.api import HostAPI
    failures = fuzz_parity.run(seed=2026, n_cases=100, hip=HostAPI(), orc=ref, verbose=False)
    assert not failures, "\n".join(failures[:10])


def test_given_f0_sweep_slice(ref, wh):
    """150 cases of tests/fuzz_given_f0.py (the sweep that once showed the unexplained 1e-4 D4C deviations):
    arbitrary caller-made F0 tracks through StoneMask / CheapTrick / D4C / Synthesis; on any failure the shared
    randn table is re-verified so that a corrupted table and a wrong kernel can be told apart"""
    import fuzz_given_f0
    from world_amd.api import HostAPI
    failures = fuzz_given_f0.run(seed=77, n_cases=150, hip=HostAPI(), orc=ref, verbose=False)
    assert not failures, f"randn table intact: {wh.verify_tables()}\n" + "\n".join(failures[:10])


def test_batched_equals_single_sweep_slice(wh):
    """40 random ragged batches (tests/fuzz_batched.py): batched calls are bit-identical to single-utterance calls"""
    import fuzz_batched
    failures = fuzz_batched.run(seed=5, n_cases=40, wh=wh, verbose=False)
    assert not failures, "\n".join(failures[:10])


def test_sixty_second_utterance(wh, ref, pool):
    """sizes well beyond the bench: 2 880 000 samples, 12 001 frames, every frame against the reference"""
    import torch
    from world_amd import synth
    x = torch.cat([synth.utterance(100 + i, FS, 10.0) for i in range(6)])
    x = torch.round(x * 32768.0) / 32768.0
    xn = x.numpy()
    tp_r, f0_r = ref.harvest(xn, FS)
    jobs = [pool.submit(ref.cheaptrick, xn, FS, tp_r, f0_r, fft_size=2048), pool.submit(ref.d4c, xn, FS, tp_r, f0_r, 2048)]
    tpos, f0, sp, ap, nf = wh.analyze(x.cuda().unsqueeze(0), FS)
    torch.cuda.synchronize()
    assert int(nf[0]) == 12001
    _check_full("60 s", (tpos[0].cpu().numpy(), f0[0].cpu().numpy(), sp[0].cpu().numpy(), ap[0].cpu().numpy()),
                (tp_r, f0_r, jobs[0].result(), jobs[1].result()))
    wh.close()


def test_voiced_section_longer_than_the_lds_is_smoothed_out_of_hbm(wh, ref):
    """SmoothF0Contour filters a voiced section in LDS when it fits (up to 18 540 base frames); a steady 24 s tone is one
    section of ~24 000 and takes the HBM route of hc_smooth -- same contour as the reference either way"""
    import torch
    fs, dur = 16000, 24.0
    t = np.arange(int(fs * dur)) / fs
    f0_true = 150.0 + 6.0 * np.sin(2 * np.pi * 0.31 * t)
    ph = 2 * np.pi * np.cumsum(f0_true) / fs
    x = sum(np.sin(k * ph) / k for k in range(1, 9)) * 0.2
    x = np.round(x * 32768.0) / 32768.0
    tp_r, f0_r = ref.harvest(x, fs)
    voiced = f0_r > 0
    runs = np.diff(np.flatnonzero(np.diff(np.concatenate([[0], voiced.astype(np.int8), [0]]))))[::2]
    assert runs.max() * 5 > 18540, "the tone must stay voiced long enough to leave the LDS route"
    tpos, f0, nf = wh.harvest(torch.from_numpy(x).cuda().unsqueeze(0), fs)
    torch.cuda.synchronize()
    n = len(f0_r)
    assert int(nf[0]) == n and np.array_equal(tpos[0, :n].cpu().numpy(), tp_r)
    assert_f0_close(f0[0, :n].cpu().numpy(), f0_r, what="24 s tone f0")


def test_bench_multi_rank_path_on_one_gpu():
    """bench.py --gpus 2 end to end (the SCALE run's code path): two ranks share this box's one GPU, the all-gather
    goes through gloo with host staging (a functional check, never a number of record): partition, batched analysis
    in sub-batches, pack, ONE all-gather, per-utterance views -- and the run's own check that utterances received
    from the other rank are bit-identical to a lone analysis"""
    env = dict(os.environ, WORLD_HIP_BENCH_BACKEND="gloo")
    port = 29600 + os.getpid() % 300
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--steps", "1",
           "--warmup", "1", "--job-utterances", "10", "--sub-batch", "3", "--min-wall", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["config"]["workload"].startswith("configs[3]")
    assert line["config"]["frames_per_step"] == 10 * 1001 and line["value"] > 0
    p = line["parity_in_run"]
    assert p["every_rank_bit_identical_to_lone_analysis"] and p["randn_table_intact"], p
    assert line["phases"]["compute_ms_per_step_max_over_ranks"] > 0 and line["roofline"]["kernel"]


def test_bare_bench_command_spawns_its_own_ranks():
    """the driver types `python bench.py --gpus N ...` with NO launcher.  The bare command must become N
    ranks by itself (here: 2 ranks sharing the one GPU over gloo) and print a complete line -- n_gpus == N as the
    communicator reports it, `cpu_baseline` timed on rank 0, every rank's device named."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["WORLD_HIP_BENCH_BACKEND"] = "gloo"
    cmd = [sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
           "--job-utterances", "8", "--sub-batch", "2", "--min-wall", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["rccl_ranks_seen"] == 2 and len(line["ranks"]) == 2
    assert sorted(d["rank"] for d in line["ranks"]) == [0, 1] and all(d["device"] for d in line["ranks"])
    cpu = line["cpu_baseline"]
    assert cpu and cpu["value"] > 0 and cpu["cores"] == 1 and cpu["kind"] in ("reference", "port") and cpu["sample"]
    assert line["environment"]["launcher"].startswith("bench.py itself")
    assert line["environment"]["hsa_ipc"]["HSA_ENABLE_IPC_MODE_LEGACY"] is not None
    assert line["parity_in_run"]["every_rank_bit_identical_to_lone_analysis"]


@pytest.mark.parametrize("wire", ["f64", "f32"])
def test_rccl_collective_path_executes_with_a_world_of_one(wire):
    """the RCCL path had never executed.  torch.distributed backend "nccl" (= RCCL), world_size 1: the
    communicator initialises and every sub-batch's in-place all_gather_into_tensor (input aliasing out[rank]) runs on
    the lanes' streams; every utterance's records are bit-identical to a lone analysis (f32 wire: rounded once)."""
    port = 29200 + (os.getpid() + (31 if wire == "f32" else 0)) % 250
    r = subprocess.run([sys.executable, os.path.join(HERE, "nccl_single_rank.py"), wire, str(port)], capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0 and "nccl single rank ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
