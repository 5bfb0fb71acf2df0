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
        (g_sp, g_f0), works = wd.all_gather_results([wd.pad_shard(local, rows), wd.pad_shard(f0_local, rows)],
                                                   async_op=True)
        wd.wait_all(works)
        full = wd.assemble(g_sp, parts, len(lengths))
        full_f0 = wd.assemble(g_f0, parts, len(lengths))
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
        f0, sp, ap, nf = wd.analyze_sharded(xs, 16000, analyze=_fake_analyze)
        for i, x in enumerate(xs):
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
    f0, sp, ap, nf = wd.analyze_sharded(xs, 16000, analyze=_fake_analyze)
    assert f0.shape == (2, int(nf.max())) and sp.shape == (2, int(nf.max()), 5) and torch.equal(ap, -sp)
