"""The N > 1 path on CPU: world_size-2 gloo processes exercise the sharding and the
all-gather reassembly used by bench.py --gpus N (same code, backend "nccl" = RCCL there)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from world_amd import distributed as wd


def test_partition_is_balanced_and_complete():
    lengths = [240000, 120000, 480000, 100, 240000, 240000, 10, 480000]
    parts = wd.partition(lengths, 3)
    assert sorted(i for p in parts for i in p) == list(range(len(lengths)))
    loads = [sum(lengths[i] for i in p) for p in parts]
    assert max(loads) - min(loads) <= max(lengths)
    assert parts == wd.partition(lengths, 3)            # deterministic on every rank
    assert wd.partition([5, 5], 4) == [[0], [1], [], []]


def _worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lengths = [300, 100, 200, 400, 150]
        parts = wd.partition(lengths, world)
        rows = max(len(p) for p in parts)
        mine = parts[rank]
        # stand-in for the per-utterance analysis: a result that identifies (utterance, frame, bin)
        def fake(i):
            return (torch.arange(6 * 4, dtype=torch.float64).reshape(6, 4) + 1000.0 * i)
        local = torch.stack([fake(i) for i in mine]) if mine else torch.zeros((0, 6, 4), dtype=torch.float64)
        f0_local = torch.stack([fake(i)[:, 0] for i in mine]) if mine else torch.zeros((0, 6), dtype=torch.float64)
        def pad(t):                                   # ranks may own different numbers of utterances
            return torch.cat([t, torch.zeros((rows - t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype)], 0)

        def assemble(g):                              # undo partition(): [world, rows, ...] -> utterance order
            out = torch.empty((len(lengths),) + tuple(g.shape[2:]), dtype=g.dtype)
            for r, idx in enumerate(parts):
                for j, i in enumerate(idx):
                    out[i] = g[r, j]
            return out
        (g_sp, g_f0), works = wd.all_gather_results([pad(local), pad(f0_local)], async_op=True)
        wd.wait_all(works)
        full = assemble(g_sp)
        full_f0 = assemble(g_f0)
        for i in range(len(lengths)):
            assert torch.equal(full[i], fake(i))
            assert torch.equal(full_f0[i], fake(i)[:, 0])
        np.save(os.path.join(tmp, f"ok{rank}.npy"), np.array([1]))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_gather_roundtrip(tmp_path):
    world = 2
    port = 29500 + os.getpid() % 400
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f"ok{r}.npy") for r in range(world))


def _fake_analyze(x, fs, x_len=None, frame_period=5.0, **_):
    """stand-in for WorldHip.analyze with the same shapes: results that identify (utterance content, frame, bin)"""
    from world_amd.api import frame_count
    nf = [frame_count(fs, int(n), frame_period) for n in x_len]
    F, hop, nb = max(nf), int(fs * frame_period / 1000.0), 5
    f0 = torch.zeros((x.shape[0], F), dtype=torch.float64)
    for u, n in enumerate(nf):
        idx = torch.clamp(torch.arange(n) * hop, max=int(x_len[u]) - 1)
        f0[u, :n] = x[u, idx]
    sp = f0[:, :, None] + torch.arange(nb, dtype=torch.float64)
    return None, f0, sp, -sp, torch.tensor(nf)


def _sharded_worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(5)
        lengths = [1600, 400, 2400, 801, 1203] if world == 2 else [700]      # world 3: two ranks own nothing
        xs = [torch.rand(n, generator=g, dtype=torch.float64) + i for i, n in enumerate(lengths)]
        res = wd.analyze_sharded(xs, 16000, analyze=_fake_analyze, bins=5, sub_batch=2)
        f0, sp, ap, nf = res.dense()
        for i, x in enumerate(xs):
            _, f0_v, sp_v, ap_v = res.utterance(i)            # views into the gathered block
            assert torch.equal(f0_v, f0[i, :f0_v.shape[0]]) and torch.equal(sp_v, sp[i, :f0_v.shape[0]])
            _, f0_i, sp_i, ap_i, nf_i = _fake_analyze(x[None], 16000, x_len=[len(x)])
            n = int(nf_i[0])
            assert int(nf[i]) == n
            assert torch.equal(f0[i, :n], f0_i[0]) and torch.equal(sp[i, :n], sp_i[0]) and torch.equal(ap[i, :n], ap_i[0])
            assert torch.all(f0[i, n:] == 0)
        np.save(os.path.join(tmp, f"sharded{world}_{rank}.npy"), np.array([1]))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
@pytest.mark.parametrize("world", [2, 3])
def test_analyze_sharded_reassembles_every_utterance_on_every_rank(tmp_path, world):
    port = 29900 + (os.getpid() + 7 * world) % 90
    mp.spawn(_sharded_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f"sharded{world}_{r}.npy") for r in range(world))


def test_analyze_sharded_without_a_process_group():
    xs = [torch.rand(900, dtype=torch.float64), torch.rand(300, dtype=torch.float64)]
    f0, sp, ap, nf = wd.analyze_sharded(xs, 16000, analyze=_fake_analyze, bins=5).dense()
    assert len(wd.analyze_sharded([], 16000, analyze=_fake_analyze)) == 0
    assert f0.shape == (2, int(nf.max())) and sp.shape == (2, int(nf.max()), 5) and torch.equal(ap, -sp)


def test_analyze_sharded_refuses_a_dense_analysis_with_another_bin_count():
    """every rank sizes its receive buffers from `bins`; an analysis that returns rows of another width must fail loudly
    (on a multi-rank job the ranks that own no utterance of the chunk would otherwise wait in the all-gather for ever)"""
    xs = [torch.rand(900, dtype=torch.float64)]
    with pytest.raises(ValueError, match="bins"):
        wd.analyze_sharded(xs, 16000, analyze=_fake_analyze, bins=7)       # _fake_analyze returns 5 bins
    f0, sp, ap, nf = wd.analyze_sharded(xs, 16000, analyze=_fake_analyze, bins=5).dense()
    assert sp.shape[-1] == 5


# ---- the real analysis on 2 ranks: the kernel sources compiled for the host (tests/emu) ----------------
def _emu_analyze(x, fs, x_len=None, frame_period=5.0, **_):
    """WorldHip.analyze's contract on CPU tensors, computed by the emulated kernels (tests/emu/libworld_emu.so)
    through the drop-in C ABI, one utterance at a time"""
    import subprocess
    from world_amd.api import HostAPI, cheaptrick_fft_size, frame_count
    emu_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu")
    subprocess.run(["make", "-s", "-f", os.path.join(emu_dir, "Makefile")], check=True)
    H = HostAPI(os.path.join(emu_dir, "libworld_emu.so"))
    fft = cheaptrick_fft_size(fs)
    nf = [frame_count(fs, int(n), frame_period) for n in x_len]
    B, F, nb = x.shape[0], max(nf), fft // 2 + 1
    tpos, f0 = torch.zeros((B, F), dtype=torch.float64), torch.zeros((B, F), dtype=torch.float64)
    sp, ap = torch.zeros((B, F, nb), dtype=torch.float64), torch.zeros((B, F, nb), dtype=torch.float64)
    for u in range(B):
        xu = x[u, :int(x_len[u])].numpy()
        tp, f = H.harvest(xu, fs, frame_period=frame_period)
        tpos[u, :nf[u]], f0[u, :nf[u]] = torch.from_numpy(tp), torch.from_numpy(f)
        sp[u, :nf[u]] = torch.from_numpy(H.cheaptrick(xu, fs, tp, f, fft_size=fft))
        ap[u, :nf[u]] = torch.from_numpy(H.d4c(xu, fs, tp, f, fft))
    return tpos, f0, sp, ap, torch.tensor(nf)


def _emu_worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from world_amd import synth
        fs = 16000
        lengths = [4000, 2600, 3300]
        parts = wd.partition(lengths, world)
        # every rank materialises only its own share (dict form of x_list)
        xs = {i: synth.utterance(i, fs, lengths[i] / fs) for i in parts[rank]}
        res = wd.analyze_sharded(xs, fs, lengths=lengths, analyze=_emu_analyze, sub_batch=1)
        for i, n in enumerate(lengths):                      # every utterance, on every rank, equals a lone analysis
            tp_i, f0_i, sp_i, ap_i, nf_i = _emu_analyze(synth.utterance(i, fs, n / fs)[None], fs, x_len=[n])
            tp, f0, sp, ap = res.utterance(i)
            k = int(nf_i[0])
            assert res.n_frames[i] == k and tp.shape[0] == k
            assert torch.equal(tp, tp_i[0]) and torch.equal(f0, f0_i[0]) and torch.equal(sp, sp_i[0]) and torch.equal(ap, ap_i[0])
        np.save(os.path.join(tmp, f"emu{rank}.npy"), np.array([1]))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_run_the_real_analysis_and_exchange_packed_blocks(tmp_path):
    """SURVEY.md 8e end to end on CPU: partition -> (emulated) Harvest + CheapTrick + D4C per rank -> one packed
    block per rank -> one all-gather -> per-utterance views, bit-identical to analysing each utterance alone"""
    world = 2
    port = 29700 + os.getpid() % 200
    mp.spawn(_emu_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f"emu{r}.npy") for r in range(world))


def test_chunk_sizes_taper_the_last_sub_batch():
    """only the last chunk's all-gather is exposed: a 128-utterance share in sub-batches of 32 is cut 32, 32, 32, 16, 8, 8"""
    assert wd.chunk_sizes(128, 32) == [32, 32, 32, 16, 8, 8]
    assert wd.chunk_sizes(128, 32, taper=False) == [32, 32, 32, 32]
    assert wd.chunk_sizes(40, 32) == [32, 4, 2, 2]
    assert wd.chunk_sizes(37, 32) == [32, 5]                      # tails of fewer than 8 stay whole
    assert wd.chunk_sizes(5, 32) == [5] and wd.chunk_sizes(0, 32) == []
    for n in range(0, 200):
        for sb in (1, 3, 8, 32):
            s = wd.chunk_sizes(n, sb)
            assert sum(s) == n and all(0 < v <= sb for v in s)
    parts = [list(range(0, 24)), list(range(24, 33))]             # ranks with shares of 24 and 9
    ch = wd.chunks_of(parts, 8)
    assert [len(c[0]) for c in ch] == [8, 8, 4, 2, 2] and [len(c[1]) for c in ch] == [8, 1, 0, 0, 0]
    assert sorted(i for c in ch for p in c for i in p) == list(range(33))


# ---- the packed analysis (the stage kernels write the records) on 2 ranks, both wire formats -----------
def _emu_packed(wire_cols):
    """WorldHip.analyze_packed's contract on CPU tensors: world_hip_analyze_packed of the host-compiled library"""
    import ctypes as C
    import subprocess
    from world_amd.api import (CheapTrickOption, D4COption, HarvestOption, cheaptrick_fft_size, frame_count, load_library)
    emu_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu")
    subprocess.run(["make", "-s", "-f", os.path.join(emu_dir, "Makefile")], check=True)
    L = load_library(os.path.join(emu_dir, "libworld_emu.so"))
    ctx = L.world_hip_create(0, None)

    def run(x, fs, block, first_row=0, x_len=None, frame_period=5.0, **_):
        fft = cheaptrick_fft_size(fs)
        assert block.shape[-1] == wire_cols(fft // 2 + 1) and block.is_contiguous() and x.is_contiguous()
        xl = np.ascontiguousarray(x_len, dtype=np.int32)
        h, c, d = HarvestOption(71.0, 800.0, frame_period), CheapTrickOption(-0.15, 71.0, fft), D4COption(0.85)
        rc = L.world_hip_analyze_packed(ctx, x.shape[0], fs, x.data_ptr(), x.shape[1], xl.ctypes.data_as(C.POINTER(C.c_int)),
                                        C.byref(h), C.byref(c), C.byref(d), first_row, block.data_ptr(), block.shape[-1])
        assert rc == 0, L.world_hip_last_error().decode()
        return [frame_count(fs, int(n), frame_period) for n in xl]
    return run


def _emu_coded(dims):
    """WorldHip.analyze_coded's contract on CPU tensors: world_hip_analyze_coded of the host-compiled library; also returns a
    coder of dense rows (world_hip_code_*) for the comparison"""
    import ctypes as C
    import subprocess
    from world_amd.api import (CheapTrickOption, D4COption, HarvestOption, cheaptrick_fft_size, frame_count, load_library)
    emu_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu")
    subprocess.run(["make", "-s", "-f", os.path.join(emu_dir, "Makefile")], check=True)
    L = load_library(os.path.join(emu_dir, "libworld_emu.so"))
    ctx = L.world_hip_create(0, None)

    def run(x, fs, block, first_row=0, x_len=None, frame_period=5.0, **_):
        fft = cheaptrick_fft_size(fs)
        assert block.shape[-1] == L.world_hip_coded_columns(fs, dims) and block.is_contiguous() and x.is_contiguous()
        xl = np.ascontiguousarray(x_len, dtype=np.int32)
        h, c, d = HarvestOption(71.0, 800.0, frame_period), CheapTrickOption(-0.15, 71.0, fft), D4COption(0.85)
        rc = L.world_hip_analyze_coded(ctx, x.shape[0], fs, x.data_ptr(), x.shape[1], xl.ctypes.data_as(C.POINTER(C.c_int)),
                                       C.byref(h), C.byref(c), C.byref(d), dims, first_row, block.data_ptr(), block.shape[-1])
        assert rc == 0, L.world_hip_last_error().decode()
        return [frame_count(fs, int(n), frame_period) for n in xl]

    def code(sp, ap, fs):
        fft = cheaptrick_fft_size(fs)
        n = sp.shape[0]
        nap = L.GetNumberOfAperiodicities(fs)
        mc, bap = torch.zeros((n, dims), dtype=torch.float64), torch.zeros((n, nap), dtype=torch.float64)
        sp, ap = sp.contiguous(), ap.contiguous()
        assert L.world_hip_code_spectral_envelope(ctx, n, fs, fft, dims, sp.data_ptr(), mc.data_ptr()) == 0
        assert L.world_hip_code_aperiodicity(ctx, n, fs, fft, ap.data_ptr(), bap.data_ptr()) == 0
        return mc, bap
    return run, code


def _coded_worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from world_amd import synth
        fs, dims = 16000, 24
        lengths = [4000, 2600, 3300, 2000]
        xs = [synth.utterance(i, fs, n / fs) for i, n in enumerate(lengths)]
        run, code = _emu_coded(dims)
        res = wd.analyze_sharded(xs, fs, analyze_packed=run, sub_batch=1, wire="coded", coded_dimensions=dims)
        nap = wd.number_of_aperiodicities(fs)
        assert res.blocks[0].shape[-1] == 2 + dims + nap == wd.wire_columns("coded", 513, fs, dims)
        for i, n in enumerate(lengths):
            tp_i, f0_i, sp_i, ap_i, nf_i = _emu_analyze(xs[i][None], fs, x_len=[n])
            k = int(nf_i[0])
            mc_want, bap_want = code(sp_i[0, :k], ap_i[0, :k], fs)
            tp, f0, mc, bap = res.utterance(i)
            assert torch.equal(tp, tp_i[0, :k]) and torch.equal(f0, f0_i[0, :k])
            # the coders applied to the rows inside the records == the coders applied to a lone analysis' dense rows
            assert mc.shape == (k, dims) and bap.shape == (k, nap)
            assert torch.equal(mc, mc_want) and torch.equal(bap, bap_want)
        np.save(os.path.join(tmp, f"coded_{rank}.npy"), np.array([1]))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_exchange_coded_records(tmp_path):
    """the coded wire format: [tpos, f0, mel-cepstrum, band aperiodicity] records written on the 'device'
    before the all-gather, on 2 gloo ranks with the host-compiled kernels: every utterance on every rank equals
    CodeSpectralEnvelope / CodeAperiodicity of a lone analysis, bit for bit"""
    world = 2
    port = 29500 + os.getpid() % 180
    mp.spawn(_coded_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f"coded_{r}.npy") for r in range(world))


def _packed_worker(rank, world, port, tmp, wire):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from world_amd import synth
        fs = 16000
        lengths = [4000, 2600, 3300, 2000]
        xs = [synth.utterance(i, fs, n / fs) for i, n in enumerate(lengths)]
        res = wd.analyze_sharded(xs, fs, analyze_packed=_emu_packed(wd.WIRE_COLS[wire]), sub_batch=1, wire=wire)
        assert res.wire == wire
        for i, n in enumerate(lengths):
            tp_i, f0_i, sp_i, ap_i, nf_i = _emu_analyze(xs[i][None], fs, x_len=[n])
            tp, f0, sp, ap = res.utterance(i)
            assert torch.equal(tp, tp_i[0]) and torch.equal(f0, f0_i[0])          # the record's head stays float64
            if wire == "f64":
                assert torch.equal(sp, sp_i[0]) and torch.equal(ap, ap_i[0])
            else:                                                              # the f64 results rounded ONCE to float32
                assert sp.dtype == torch.float32 and ap.dtype == torch.float32
                assert torch.equal(sp, sp_i[0].to(torch.float32)) and torch.equal(ap, ap_i[0].to(torch.float32))
        f0_d, sp_d, ap_d, _ = res.dense()
        assert sp_d.dtype == torch.float64 and float(sp_d[0, :res.n_frames[0]].min()) > 0      # dense(): float64 again
        np.save(os.path.join(tmp, f"packed_{wire}_{rank}.npy"), np.array([1]))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("wire", ["f64", "f32"])
def test_two_ranks_exchange_records_written_by_the_stage_kernels(tmp_path, wire):
    """the path bench.py --gpus N runs (analyze_packed -> in-place all-gather), both wire formats, on 2 gloo ranks with
    the host-compiled kernels: every utterance on every rank equals a lone analysis (f32: rounded once to float)"""
    world = 2
    port = 29300 + (os.getpid() + (17 if wire == "f32" else 0)) % 180
    mp.spawn(_packed_worker, args=(world, port, str(tmp_path), wire), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f"packed_{wire}_{r}.npy") for r in range(world))


def test_own_buffers_survive_the_next_call():
    """the default result views a per-shape cache that the next call overwrites; own_buffers=True gives a
    result that stays valid while the next step runs"""
    xs = [torch.rand(900, dtype=torch.float64), torch.rand(300, dtype=torch.float64)]
    ys = [x + 5.0 for x in xs]
    kept = wd.analyze_sharded(xs, 16000, analyze=_fake_analyze, bins=5, own_buffers=True)
    snap = kept.utterance(0)[1].clone()
    nxt = wd.analyze_sharded(ys, 16000, analyze=_fake_analyze, bins=5, own_buffers=True)
    assert torch.equal(kept.utterance(0)[1], snap) and not torch.equal(nxt.utterance(0)[1], snap)
    a = wd.analyze_sharded(xs, 16000, analyze=_fake_analyze, bins=5)
    b = wd.analyze_sharded(ys, 16000, analyze=_fake_analyze, bins=5)          # same shape: the cached buffers are reused
    assert a.blocks[0].data_ptr() == b.blocks[0].data_ptr()


# ---- frame-level sharding of ONE long utterance (last sentence) -----------------------------------
def _emu_range_backend(fs):
    """harvest / spectral_range of analyze_long_sharded on CPU tensors: the host-compiled library's batched entry points"""
    import ctypes as C
    import subprocess
    from world_amd.api import (CheapTrickOption, D4COption, HarvestOption, cheaptrick_fft_size, frame_count, load_library)
    emu_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu")
    subprocess.run(["make", "-s", "-f", os.path.join(emu_dir, "Makefile")], check=True)
    L = load_library(os.path.join(emu_dir, "libworld_emu.so"))
    ctx = L.world_hip_create(0, None)
    fft = cheaptrick_fft_size(fs)
    ip = C.POINTER(C.c_int)

    def harvest(xb):
        n = xb.shape[1]
        nf = frame_count(fs, n, 5.0)
        tp, f0 = torch.zeros((1, nf), dtype=torch.float64), torch.zeros((1, nf), dtype=torch.float64)
        xl = np.array([n], dtype=np.int32)
        h = HarvestOption(71.0, 800.0, 5.0)
        assert L.world_hip_harvest_batch(ctx, 1, fs, xb.data_ptr(), n, xl.ctypes.data_as(ip), C.byref(h), nf, tp.data_ptr(), f0.data_ptr()) == 0
        return tp, f0

    def spectral_range(xb, tp, f0, block, lo, hi, reuse=False):
        n, nf = xb.shape[1], tp.shape[1]
        xl, nfa = np.array([n], dtype=np.int32), np.array([nf], dtype=np.int32)
        c, d = CheapTrickOption(-0.15, 71.0, fft), D4COption(0.85)
        rc = L.world_hip_spectral_packed_range(ctx, 1, fs, xb.data_ptr(), n, xl.ctypes.data_as(ip), nfa.ctypes.data_as(ip), nf,
                                               tp.data_ptr(), f0.data_ptr(), C.byref(c), C.byref(d), lo, hi, 1 if reuse else 0,
                                               0, block.data_ptr(), block.shape[-1])
        assert rc == 0, L.world_hip_last_error().decode()
    return harvest, spectral_range


def test_frame_ranges_cover_the_utterance():
    assert wd.frame_ranges(1001, 1) == [(0, 1001)]
    assert wd.frame_ranges(1001, 4) == [(0, 256), (256, 512), (512, 768), (768, 1001)]
    assert wd.frame_ranges(100, 8)[0] == (0, 64) and wd.frame_ranges(100, 8)[1] == (64, 100) and wd.frame_ranges(100, 8)[2] == (100, 100)
    for nf in (1, 63, 64, 65, 2001, 48001):
        for w in (1, 2, 3, 8):
            r = wd.frame_ranges(nf, w)
            assert r[0][0] == 0 and r[-1][1] == nf and all(r[i][1] == r[i + 1][0] for i in range(w - 1))


def _long_worker(rank, world, port, tmp, wire):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from world_amd import synth
        fs = 16000
        x = synth.utterance(4, fs, 0.9)                                  # 181 frames: ranges 0-128 and 128-181
        harvest, spectral_range = _emu_range_backend(fs)
        tp, f0, sp, ap = wd.analyze_long_sharded(x, fs, wire=wire, harvest=harvest, spectral_range=spectral_range, sub_frames=64)
        tp_i, f0_i, sp_i, ap_i, nf_i = _emu_analyze(x[None], fs, x_len=[x.numel()])
        k = int(nf_i[0])
        assert tp.shape[0] == k == sp.shape[0] == ap.shape[0]
        assert torch.equal(tp, tp_i[0]) and torch.equal(f0, f0_i[0])
        if wire == "f64":
            assert torch.equal(sp, sp_i[0]) and torch.equal(ap, ap_i[0])
        else:
            assert torch.equal(sp, sp_i[0].to(torch.float32)) and torch.equal(ap, ap_i[0].to(torch.float32))
        np.save(os.path.join(tmp, f"long_{wire}_{rank}.npy"), np.array([1]))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("wire", ["f64", "f32"])
def test_two_ranks_share_the_frames_of_one_utterance(tmp_path, wire):
    """every rank runs Harvest, rank r the spectral stages of its frame range only (sub-ranges of 64 frames, records
    written by the host-compiled stage kernels, all-gathered): bit-identical to analysing the utterance alone -- the
    randn() stream positions of a range are those of the whole utterance"""
    world = 2
    port = 29100 + (os.getpid() + (13 if wire == "f32" else 0)) % 90
    mp.spawn(_long_worker, args=(world, port, str(tmp_path), wire), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f"long_{wire}_{r}.npy") for r in range(world))
