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
