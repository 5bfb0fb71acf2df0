"""Helper of tests/test_gpu_fullsize.py: the collective path of bench.py --gpus N on ONE GPU with the REAL backend --
torch.distributed "nccl" (= RCCL on ROCm), world_size 1.  RCCL loads, the communicator initialises, and every
sub-batch's in-place all_gather_into_tensor (input aliasing out[rank]) executes on the lanes' streams; the records that
come back must be bit-identical to lone analyses.  Usage: nccl_single_rank.py <wire> <port>"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from world_amd import distributed as wd, synth                       # noqa: E402
from world_amd.api import WorldHip                                   # noqa: E402

wire, port = sys.argv[1], sys.argv[2]
os.environ["MASTER_ADDR"] = "127.0.0.1"
os.environ["MASTER_PORT"] = port
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
try:
    assert dist.get_backend() == "nccl"
    fs = 48000
    secs = [0.5, 0.21, 0.37, 0.44, 0.3, 0.26, 0.33, 0.4, 0.29, 0.31, 0.22]
    xs = [synth.utterance(i, fs, d, device="cuda") for i, d in enumerate(secs)]
    timings = {}
    for _ in range(2):                                               # second pass: cached buffers, communicator warm
        res = wd.analyze_sharded(xs, fs, sub_batch=8, wire=wire, exchange_single_rank=True, timings=timings)
    torch.cuda.synchronize()
    assert timings["gathered_bytes"] == 0 and timings["steps"] == 2   # a world of one receives nothing, but the collectives ran
    assert len(res.blocks) == len(wd.chunk_sizes(len(xs), 8)) == 2
    wh = WorldHip()
    for i, x in enumerate(xs):
        tp1, f01, sp1, ap1, nf1 = wh.analyze(x[None].contiguous(), fs)
        k = int(nf1[0])
        tp, f0, sp, ap = res.utterance(i)
        assert tp.shape[0] == k and torch.equal(tp, tp1[0, :k]) and torch.equal(f0, f01[0, :k]), i
        if wire == "f64":
            assert torch.equal(sp, sp1[0, :k]) and torch.equal(ap, ap1[0, :k]), i
        else:
            assert torch.equal(sp, sp1[0, :k].to(torch.float32)) and torch.equal(ap, ap1[0, :k].to(torch.float32)), i
    print("nccl single rank ok:", wire, len(xs), "utterances,", len(res.blocks), "all-gathers per step")
finally:
    dist.destroy_process_group()
