"""The race-shaking builds (world_amd/csrc/devrt.h: -DWH_JITTER, -DWH_LDS_POISON; built by __graft_entry__.build() into
world_amd/variants/) must compute the SAME BITS as the product library on the analysis paths.  The frame kernels alias LDS
regions across phases and place their barriers by hand; a missing barrier or a read of never-written LDS is invisible in
normal runs (the wavefronts of a workgroup arrive together; a fresh workgroup inherits plausible data) and used to be found
by reading the code (d4c_frame's select zeroed histograms other wavefronts were still reading).  With random
multi-thousand-cycle stalls behind every barrier and wave fence, or NaN patterns in every LDS byte a kernel did not write,
such a bug changes the result."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANTS = os.path.join(ROOT, "world_amd", "variants")


def _lib(name):
    path = os.path.join(VARIANTS, f"libworld_hip_{name}.so")
    assert os.path.exists(path), f"{path} is missing: python -m world_amd.build --checked (or __graft_entry__.build()) builds it"
    return path


def _workloads():
    import torch
    from world_amd import synth
    out = []
    # configs[1]'s shape (48 kHz, Harvest + CheapTrick + D4C; 4096-point D4C transforms, 2048-point CheapTrick), ragged batch
    xs = [synth.vowel(48000, 1.3, seed=21, base_f0=120.0), synth.utterance(5, 48000, 0.9), synth.vowel(48000, 0.6, seed=4, base_f0=310.0)]
    L = max(x.numel() for x in xs)
    xb = torch.zeros((len(xs), L), dtype=torch.float64)
    for i, x in enumerate(xs):
        xb[i, :x.numel()] = x
    out.append(("48k harvest ragged", xb, 48000, [x.numel() for x in xs], dict(f0_method="harvest")))
    # configs[4]'s shape (16 kHz, DIO + StoneMask + CheapTrick 1024 + D4C 2048-point)
    xs = [synth.utterance(3, 16000, 1.1), synth.vowel(16000, 0.8, seed=9, base_f0=180.0)]
    L = max(x.numel() for x in xs)
    xb = torch.zeros((len(xs), L), dtype=torch.float64)
    for i, x in enumerate(xs):
        xb[i, :x.numel()] = x
    out.append(("16k dio ragged", xb, 16000, [x.numel() for x in xs], dict(f0_method="dio", q1=-0.15, threshold=0.85)))
    # a low voice at 48 kHz: D4C windows longer than half the transform, CheapTrick's longest smoothing segments
    out.append(("48k low voice", synth.vowel(48000, 0.8, seed=2, base_f0=62.0)[None].contiguous(), 48000, None, dict(f0_method="harvest", f0_floor=50.0)))
    # 22.05 kHz: Harvest's decimation ratio 3 (windows by rotation, not from the table), 2048-point D4C
    out.append(("22k harvest", synth.utterance(8, 22050, 0.8)[None].contiguous(), 22050, None, dict(f0_method="harvest")))
    # 176.4 kHz: the 16384-point D4C shape (sixteen wavefronts per workgroup, the group delay parked in global memory), windows
    # longer than half its transform, LoveTrain's run-time-length 16384-point transform, CheapTrick's 8192-point shape
    out.append(("176k dio low voice", synth.vowel(176400, 0.25, seed=6, base_f0=65.0)[None].contiguous(), 176400, None, dict(f0_method="dio", f0_floor=50.0)))
    return out


def _run(lib_path, workloads, repeats=1):
    import torch
    from world_amd.api import WorldHip
    wh = WorldHip(device=0, lib_path=lib_path)
    results = []
    for _ in range(repeats):
        res = []
        for name, x, fs, x_len, kw in workloads:
            tpos, f0, sp, ap, nf = wh.analyze(x.cuda(), fs, x_len=x_len, **kw)
            torch.cuda.synchronize()
            keep = []
            for u, n in enumerate(nf):                       # the valid frames only: padding is the caller's
                n = int(n)
                keep.append((tpos[u, :n].cpu().numpy(), f0[u, :n].cpu().numpy(), sp[u, :n].cpu().numpy(), ap[u, :n].cpu().numpy()))
            res.append(keep)
        results.append(res)
    wh.close()
    return results


def _same(a, b):
    return all(np.array_equal(x, y) for ua, ub in zip(a, b) for x, y in zip(ua, ub))


def test_jitter_build_is_bit_identical():
    from world_amd.api import LIB_PATH
    work = _workloads()
    want = _run(LIB_PATH, work)[0]
    assert all(np.isfinite(arr).all() for wl in want for utt in wl for arr in utt)
    got = _run(_lib("jitter"), work, repeats=3)              # three passes: the stalls fall differently every time
    for rep, res in enumerate(got):
        for (name, *_), a, b in zip(work, want, res):
            assert _same(a, b), f"jitter build, pass {rep}, {name}: results differ from the product build's"


def test_lds_poison_build_is_bit_identical():
    from world_amd.api import LIB_PATH
    work = _workloads()
    want = _run(LIB_PATH, work)[0]
    got = _run(_lib("poison"), work)[0]
    for (name, *_), a, b in zip(work, want, got):
        assert all(np.isfinite(arr).all() for utt in b for arr in utt), f"poison build, {name}: a NaN pattern reached the output"
        assert _same(a, b), f"poison build, {name}: results differ from the product build's"
