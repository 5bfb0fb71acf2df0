"""Helper of the route tests: Harvest's F0 of a few utterances to an .npz, in a process of its own because
WORLD_HIP_MERGE_LDS_SECTIONS is read once per process.   python merge_routes.py emu|gpu out.npz"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from world_amd import synth  # noqa: E402

CASES = [(0, 16000, 1.0), (3, 16000, 2.5), (11, 16000, 3.0), (21, 44100, 1.5)]


def main(kind, path):
    out = {}
    if kind == "emu":
        from world_amd.api import HostAPI
        H = HostAPI(os.path.join(ROOT, "tests", "emu", "libworld_emu.so"))
        for seed, fs, dur in CASES[:3]:
            out[f"{seed}_{fs}"] = H.harvest(synth.utterance(seed, fs, dur).numpy(), fs)[1]
    else:
        import torch
        from world_amd.api import WorldHip
        wh = WorldHip()
        for seed, fs, dur in CASES:
            x = synth.utterance(seed, fs, dur)
            out[f"{seed}_{fs}"] = wh.harvest(x.cuda().unsqueeze(0), fs)[1][0].cpu().numpy()
        # a batch: utterances of different lengths side by side
        xs = [synth.utterance(40 + i, 16000, 1.0 + 0.4 * i) for i in range(6)]
        L = max(len(x) for x in xs)
        xb = torch.zeros((len(xs), L), dtype=torch.float64)
        for i, x in enumerate(xs):
            xb[i, :len(x)] = x
        out["batch"] = wh.harvest(xb.cuda(), 16000, x_len=np.array([len(x) for x in xs], dtype=np.int32))[1].cpu().numpy()
        torch.cuda.synchronize()
    np.savez(path, **out)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
