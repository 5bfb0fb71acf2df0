"""Test infrastructure (python tests/fuzz_batched.py <seed> <cases> on the GPU box; a slice runs in the -m gpu suite): random ragged batches through the device-resident
API must reproduce the single-utterance calls bit for bit, whatever shares the batch."""
import os, sys, time
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
import torch
from world_amd import synth
from world_amd.api import WorldHip


def run(seed=0, n_cases=30, wh=None, verbose=True):
    wh = wh or WorldHip()
    rng = np.random.default_rng(seed)
    bad = 0
    failures = []
    t0 = time.time()
    for case in range(n_cases):
        fs = int(rng.choice([16000, 22050, 44100, 48000]))
        B = int(rng.integers(2, 7))
        lens = [int(fs * float(rng.uniform(0.05, 0.9))) for _ in range(B)]
        L = max(lens)
        x = torch.zeros((B, L), dtype=torch.float64)
        for b in range(B):
            kind = rng.choice(['vowel', 'utt', 'noise', 'silence'])
            if kind == 'vowel': v = synth.vowel(fs, lens[b] / fs + 0.01, seed=int(rng.integers(1, 10**6)), base_f0=float(rng.uniform(80, 400)))
            elif kind == 'utt': v = synth.utterance(int(rng.integers(1, 10**6)), fs, lens[b] / fs + 0.01)
            elif kind == 'noise': v = torch.from_numpy(np.round(rng.normal(size=lens[b] + 8) * 0.05 * 32768) / 32768)
            else: v = torch.zeros(lens[b] + 8, dtype=torch.float64)
            x[b, :lens[b]] = v[:lens[b]]
        x = x.cuda(); xl = np.array(lens, dtype=np.int32)
        method = str(rng.choice(['harvest', 'dio']))
        tpos, f0, sp, ap, nf = wh.analyze(x, fs, x_len=xl, f0_method=method)
        y = wh.synthesis(f0, sp, ap, nf, sp.shape[-1] * 2 - 2, 5.0, fs, xl)
        msg = []
        for b in range(B):
            t1, f1, s1, a1, n1 = wh.analyze(x[b:b + 1, :lens[b]].contiguous(), fs, f0_method=method)
            n = int(nf[b])
            if int(n1[0]) != n: msg.append(f'utt {b}: frame count'); continue
            for name, u, v in (('tpos', tpos[b, :n], t1[0, :n]), ('f0', f0[b, :n], f1[0, :n]), ('sp', sp[b, :n], s1[0, :n]), ('ap', ap[b, :n], a1[0, :n])):
                if not torch.equal(u, v): msg.append(f'utt {b}: {name} differs by {float((u - v).abs().max()):.1e}')
            y1 = wh.synthesis(f1, s1, a1, n1, sp.shape[-1] * 2 - 2, 5.0, fs, xl[b:b + 1])
            if not torch.equal(y[b, :lens[b]], y1[0, :lens[b]]): msg.append(f'utt {b}: synthesis differs')
        if msg:
            bad += 1
            failures.append(f'case {case}: fs={fs} B={B} lens={lens} {method}: ' + '; '.join(msg[:4]))
            if verbose: print(failures[-1], flush=True)
    if verbose: print(f'{n_cases} batches, {bad} with differences, {time.time() - t0:.0f} s')
    return failures


if __name__ == '__main__':
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 30)
