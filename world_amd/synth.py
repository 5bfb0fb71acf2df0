"""Deterministic synthetic utterances for the BASELINE.json configs.

No dataset can be fetched, so every benchmark/parity input is generated here:
"vowels" (vibrato harmonic complex, gated on/off so there are unvoiced gaps)
and harmonic log-chirps, both with an LCG noise floor and int16 quantisation
(x = q / 32768, exactly what the reference's wavread produces for 16-bit PCM,
tools/audioio.cpp:236-249).  Pure torch (float64), runs on CPU or on the GPU.
"""
import math

import torch


def _lcg_noise(n, seed, device):
    """uniform(-1,1) from the 32-bit LCG (1664525, 1013904223), vectorised:
    x_k = a^k x_0 + c * sum_{j<k} a^j  (mod 2^32), all in wrapping int64."""
    a = torch.full((n,), 1664525, dtype=torch.int64)
    a[0] = 1
    apow = torch.cumprod(a, 0)                      # a^k mod 2^64 (wraps)
    geo = torch.cumsum(apow, 0) - apow              # sum_{j<k} a^j
    x = (apow * int(seed) + geo * 1013904223) & 0xFFFFFFFF
    return (x.to(torch.float64) / 2147483648.0 - 1.0).to(device)


def _finish(s, fs, seed, noise, amp, gate, device):
    n = s.numel()
    s = s * (amp / s.abs().max().clamp_min(1e-30))
    if gate is not None:
        on, off = gate
        t = torch.arange(n, dtype=torch.float64, device=device) / fs
        ph = torch.remainder(t, on + off)
        ramp = 0.005
        g = torch.clamp(ph / ramp, 0, 1) * torch.clamp((on - ph) / ramp, 0, 1)
        g = 0.5 - 0.5 * torch.cos(math.pi * g)
        s = s * g
    s = s + noise * _lcg_noise(n, seed, device)
    q = torch.clamp(torch.round(s * 32768.0), -32768, 32767)
    return q / 32768.0


def vowel(fs, seconds, seed=12345, base_f0=140.0, device="cpu"):
    """f0(t) = base + 40 sin(2 pi 0.7 t) + 3 sin(2 pi 5.5 t); 30 harmonics with
    1/h * 1/(1+(h f0/2500)^2) tilt; gated 1.6 s on / 0.4 s off."""
    n = int(round(fs * seconds))
    t = torch.arange(n, dtype=torch.float64, device=device) / fs
    f0 = base_f0 + 40.0 * torch.sin(2 * math.pi * 0.7 * t) + 3.0 * torch.sin(2 * math.pi * 5.5 * t)
    f0 = f0.clamp_min(60.0)
    phase = 2 * math.pi * torch.cumsum(f0, 0) / fs
    s = torch.zeros(n, dtype=torch.float64, device=device)
    for h in range(1, 31):
        hf = h * f0
        w = (1.0 / h) / (1.0 + (hf / 2500.0) ** 2)
        w = torch.where(hf < 0.45 * fs, w, torch.zeros_like(w))
        s += w * torch.sin(h * phase)
    return _finish(s, fs, seed, 1e-3, 0.3, (1.6, 0.4), device)


def chirp(fs, seconds, seed=1, f_lo=80.0, f_hi=400.0, device="cpu"):
    """harmonic log-chirp f_lo -> f_hi over the utterance, 20 harmonics."""
    n = int(round(fs * seconds))
    t = torch.arange(n, dtype=torch.float64, device=device) / fs
    f0 = f_lo * (f_hi / f_lo) ** (t / seconds)
    phase = 2 * math.pi * torch.cumsum(f0, 0) / fs
    s = torch.zeros(n, dtype=torch.float64, device=device)
    for h in range(1, 21):
        hf = h * f0
        w = torch.where(hf < 0.45 * fs, torch.full_like(hf, 1.0 / h), torch.zeros_like(hf))
        s += w * torch.sin(h * phase)
    return _finish(s, fs, seed, 1e-3, 0.3, None, device)


def utterance(index, fs, seconds, device="cpu"):
    """Config 3/4 batch member: even index = vowel (seed = index, base f0
    90 + (index mod 32) * 8 Hz), odd index = chirp."""
    if index % 2 == 0:
        return vowel(fs, seconds, seed=index, base_f0=90.0 + (index % 32) * 8.0, device=device)
    return chirp(fs, seconds, seed=index, device=device)
