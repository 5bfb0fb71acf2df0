"""Test infrastructure (python tests/fuzz_given_f0.py <seed> <cases>; a slice runs in the -m gpu suite): the stages that take
F0 from the caller -- StoneMask, CheapTrick, D4C, Synthesis -- on arbitrary F0 tracks (zeros, values
below the floors, up to 1 kHz, jumps), HIP path against the CPU oracle."""
import os, sys, time
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
from oracle.loader import best_oracle
from world_amd import synth
from world_amd.api import HostAPI
from util import max_rel


def run(seed=0, n_cases=40, hip=None, orc=None, verbose=True):
    hip, orc = hip or HostAPI(), orc or best_oracle()
    rng = np.random.default_rng(seed)
    bad = 0
    failures = []
    t0 = time.time()
    for case in range(n_cases):
        # (round 5: 96 / 128 / 192 kHz -- D4C's 8192- and 16384-point shapes, CheapTrick's and Synthesis()' 8192-point ones;
        # WORLD_FUZZ_RATES=a,b,.. narrows the sweep to given rates)
        rates = [16000, 22050, 32000, 44100, 48000, 96000, 128000, 192000]
        if os.environ.get("WORLD_FUZZ_RATES"): rates = [int(r) for r in os.environ["WORLD_FUZZ_RATES"].split(",")]
        fs = int(rng.choice(rates))
        dur = float(rng.uniform(0.2, 0.8))
        x = synth.utterance(int(rng.integers(1, 10**6)), fs, dur).numpy()
        fp = float(rng.choice([2.5, 5.0, 10.0]))
        nf = int(1000.0 * len(x) / fs / fp) + 1
        tp = np.arange(nf) * fp / 1000.0
        style = rng.choice(['random', 'steps', 'low', 'high', 'sparse'])
        if style == 'random': f0 = rng.uniform(30.0, 1000.0, nf)
        elif style == 'steps': f0 = np.repeat(rng.uniform(60.0, 600.0, nf // 7 + 1), 7)[:nf]
        elif style == 'low': f0 = rng.uniform(20.0, 90.0, nf)
        elif style == 'high': f0 = rng.uniform(500.0, 1000.0, nf)
        else: f0 = np.where(rng.random(nf) < 0.15, rng.uniform(80.0, 400.0, nf), 0.0)
        f0[rng.random(nf) < 0.2] = 0.0
        msg = []
        sm_o, sm = orc.stonemask(x, fs, tp, f0), hip.stonemask(x, fs, tp, f0)
        if np.sum((sm > 0) != (sm_o > 0)) or max_rel(sm[sm_o > 0], sm_o[sm_o > 0]) > 1e-6: msg.append('stonemask')
        fft = hip.cheaptrick_fft_size(fs)
        sp_o, sp = orc.cheaptrick(x, fs, tp, f0, fft_size=fft), hip.cheaptrick(x, fs, tp, f0, fft_size=fft)
        e = max_rel(sp, sp_o)
        if e > 1e-6: msg.append(f'cheaptrick rel={e:.1e}')
        ap_o, ap = orc.d4c(x, fs, tp, f0, fft), hip.d4c(x, fs, tp, f0, fft)
        e = max_rel(ap, ap_o)
        if e > 1e-5: msg.append(f'd4c rel={e:.1e}')
        y_o, y = orc.synthesis(f0, sp_o, ap_o, fft, fp, fs, len(x)), hip.synthesis(f0, sp_o, ap_o, fft, fp, fs, len(x))
        e = float(np.max(np.abs(y - y_o)) / max(np.max(np.abs(y_o)), 1e-9))
        if e > 1e-7: msg.append(f'synthesis peak-rel={e:.1e}')
        if msg:
            bad += 1
            failures.append(f'case {case}: fs={fs} dur={dur:.2f} fp={fp} style={style}: ' + '; '.join(msg))
            if verbose: print(failures[-1], flush=True)
            if os.path.isdir('gpurun_out'): np.savez(f'gpurun_out/fuzz_f0_case_{case}.npz', x=x, fs=fs, fp=fp, f0=f0)
    if verbose: print(f'{n_cases} cases, {bad} with divergences, {time.time() - t0:.0f} s')
    return failures


if __name__ == '__main__':
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 40)
