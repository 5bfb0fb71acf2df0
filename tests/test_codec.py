"""Coders for the analysis outputs (reference src/codec.cpp): the oracle
restatement, the host-emulated kernels and the HIP path against golden vectors generated
from the unmodified reference (tests/golden/make_golden.py: codec.npz)."""
import os
import subprocess

import numpy as np
import pytest

from util import GOLDEN, RTOL, load_golden, max_rel

NAMES = ["vaiueo2d_harvest", "vowel48k_harvest", "vowel16k_dio"]


@pytest.fixture(scope="module")
def codec_golden():
    return dict(np.load(os.path.join(GOLDEN, "codec.npz")))


def check_codec(backend, cg, name, rtol, atol):
    g = load_golden(name)
    fs, fft = g["fs"], g["fft_size"]
    sp, ap = g["sp_rows"][:12], g["ap_rows"][:12]
    assert backend.number_of_aperiodicities(fs) == cg[f"{name}.bap"].shape[1]
    for nd in (24, 60):
        want = cg[f"{name}.mcep{nd}"]
        got = backend.code_spectral_envelope(sp, fs, fft, nd)
        assert got.shape == want.shape
        # cepstral coefficients cross zero: absolute tolerance scaled by the row's c0
        assert np.max(np.abs(got - want)) <= atol * np.max(np.abs(want))
        back = backend.decode_spectral_envelope(want, fs, fft)
        assert max_rel(back[:4], cg[f"{name}.sp_from_mcep{nd}"]) <= rtol
    got = backend.code_aperiodicity(ap, fs, fft)
    assert np.max(np.abs(got - cg[f"{name}.bap"])) <= atol * 60.0          # dB values in [-60, 0]
    back = backend.decode_aperiodicity(cg[f"{name}.bap_in"], fs, fft)
    assert max_rel(back, cg[f"{name}.ap_from_bap"]) <= rtol
    assert np.all(back[::5] == 1.0 - 1e-12)                                 # CheckVUV's aperiodic frames


@pytest.mark.parametrize("name", NAMES)
def test_port_codec_matches_golden(port_oracle, codec_golden, name):
    check_codec(port_oracle, codec_golden, name, rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("name", NAMES[:2])
def test_reference_codec_reproduces_golden(ref_oracle, codec_golden, name):
    check_codec(ref_oracle, codec_golden, name, rtol=1e-10, atol=1e-12)


@pytest.fixture(scope="module")
def emu():
    emu_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu")
    subprocess.run(["make", "-s", "-f", os.path.join(emu_dir, "Makefile")], check=True)
    from world_amd.api import HostAPI
    return HostAPI(os.path.join(emu_dir, "libworld_emu.so"))


@pytest.mark.parametrize("name", NAMES)
def test_emulated_codec_matches_golden(emu, codec_golden, name):
    check_codec(emu, codec_golden, name, rtol=1e-9, atol=1e-11)


def test_emulated_codec_edge_cases(emu, port_oracle):
    """one dimension, every bin the DCT's real FFT has, the smallest transform, zero rows"""
    rng = np.random.default_rng(5)
    for fs, fft in ((16000, 128), (22050, 512), (48000, 4096)):
        sp = np.exp(rng.normal(size=(3, fft // 2 + 1)) * 2.0)
        for nd in (1, fft // 4 + 1):
            want = port_oracle.code_spectral_envelope(sp, fs, fft, nd)
            got = emu.code_spectral_envelope(sp, fs, fft, nd)
            assert np.max(np.abs(got - want)) <= 1e-11 * np.max(np.abs(want))
            assert max_rel(emu.decode_spectral_envelope(want, fs, fft),
                           port_oracle.decode_spectral_envelope(want, fs, fft)) <= 1e-9
    assert emu.code_spectral_envelope(np.zeros((0, 513)), 22050, 1024, 24).shape == (0, 24)


# ---------------------------------------------------------------------------------------
# GPU
# ---------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def hip():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from world_amd.api import HostAPI
    return HostAPI()


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_hip_codec_matches_golden(hip, codec_golden, name):
    check_codec(hip, codec_golden, name, rtol=RTOL, atol=1e-9)


@pytest.mark.gpu
def test_hip_codec_matches_oracle_random_rows(hip):
    from oracle.loader import best_oracle
    oracle = best_oracle()
    rng = np.random.default_rng(11)
    for fs, fft in ((16000, 1024), (44100, 2048), (48000, 4096), (16000, 128)):
        sp = np.exp(rng.normal(size=(37, fft // 2 + 1)) * 3.0)
        ap = np.clip(rng.uniform(size=(37, fft // 2 + 1)), 1e-3, 1 - 1e-12)
        for nd in (1, 25, fft // 4 + 1):
            want = oracle.code_spectral_envelope(sp, fs, fft, nd)
            got = hip.code_spectral_envelope(sp, fs, fft, nd)
            assert np.max(np.abs(got - want)) <= 1e-9 * np.max(np.abs(want))
            assert max_rel(hip.decode_spectral_envelope(want, fs, fft),
                           oracle.decode_spectral_envelope(want, fs, fft)) <= RTOL
        bap = oracle.code_aperiodicity(ap, fs, fft)
        assert np.max(np.abs(hip.code_aperiodicity(ap, fs, fft) - bap)) <= 1e-9
        bap[::4] = -0.3
        assert max_rel(hip.decode_aperiodicity(bap, fs, fft), oracle.decode_aperiodicity(bap, fs, fft)) <= RTOL


@pytest.mark.gpu
def test_device_resident_codec_full_size_properties(hip):
    """configs[1] shapes (2001 frames, fft 2048): the batched device API equals the drop-in
    calls row for row, the envelope coder is linear in the log domain, and band values are
    the dB values of the bins that sit exactly on 3 kHz multiples (3000 / (48000/2048) = 128)."""
    import torch
    from world_amd.api import WorldHip
    wh = WorldHip()
    fs, fft, nf, nd = 48000, 2048, 2001, 60
    gen = torch.Generator().manual_seed(3)
    sp1 = torch.exp(torch.randn((nf, fft // 2 + 1), generator=gen, dtype=torch.float64) * 2.0).cuda()
    sp2 = torch.exp(torch.randn((nf, fft // 2 + 1), generator=gen, dtype=torch.float64)).cuda()
    ap = torch.rand((nf, fft // 2 + 1), generator=gen, dtype=torch.float64).clamp(1e-3, 1 - 1e-12).cuda()
    c1 = wh.code_spectral_envelope(sp1, fs, fft, nd)
    c2 = wh.code_spectral_envelope(sp2, fs, fft, nd)
    c12 = wh.code_spectral_envelope(sp1 * sp2, fs, fft, nd)
    scale = float(c12.abs().max())
    assert float((c12 - (c1 + c2)).abs().max()) <= 1e-11 * scale
    rows = slice(0, nf, 97)
    assert np.max(np.abs(c1[rows].cpu().numpy() - hip.code_spectral_envelope(sp1[rows].cpu().numpy(), fs, fft, nd))) \
        <= 1e-12 * scale
    back = wh.decode_spectral_envelope(c1, fs, fft)
    assert back.shape == sp1.shape and bool(torch.isfinite(back).all()) and bool((back > 0).all())
    assert max_rel(back[rows].cpu().numpy(), hip.decode_spectral_envelope(c1[rows].cpu().numpy(), fs, fft)) <= 1e-12
    bap = wh.code_aperiodicity(ap, fs, fft)
    assert bap.shape == (nf, 5)
    want = 20 * torch.log10(ap[:, 128:128 * 6:128])
    assert float((bap - want).abs().max()) <= 1e-11
    dec = wh.decode_aperiodicity(bap, fs, fft)
    # decoded values at the band centres reproduce the inputs (interp1 hits the knots)
    assert float((dec[:, 128:128 * 6:128] - ap[:, 128:128 * 6:128]).abs().max()) <= 1e-11
    # [batch, frames, bins] layout: leading dimensions are just rows
    c3 = wh.code_spectral_envelope(sp1[:2000].reshape(4, 500, -1), fs, fft, nd)
    assert torch.equal(c3.reshape(2000, nd), c1[:2000])


@pytest.mark.gpu
def test_coded_records_equal_the_reference_coders_of_the_reference_analysis():
    """world_hip_analyze_coded (the coded wire format): [tpos, f0, mel-cepstrum[60], band aperiodicity[5]]
    of a ragged 48 kHz batch == CodeSpectralEnvelope / CodeAperiodicity (reference src/codec.cpp:268-297, :217-236) applied
    by the REFERENCE to the REFERENCE's own Harvest + CheapTrick + D4C, and bit-identical to this library's coders applied
    to its own dense analysis."""
    import torch
    from oracle.loader import best_oracle
    from world_amd import synth
    from world_amd.api import WorldHip, frame_count
    oracle = best_oracle()
    wh = WorldHip()
    fs, fft, nd = 48000, 2048, 60
    xs = [synth.vowel(fs, 0.7, seed=31, base_f0=130.0), synth.utterance(4, fs, 0.45)]
    L = max(x.numel() for x in xs)
    xb = torch.zeros((2, L), dtype=torch.float64)
    for i, x in enumerate(xs):
        xb[i, :x.numel()] = x
    lens = [x.numel() for x in xs]
    nf = [frame_count(fs, n, 5.0) for n in lens]
    cols = wh.lib.world_hip_coded_columns(fs, nd)
    assert cols == 2 + nd + 5
    block = torch.full((sum(nf) + 2, cols), float("nan"), dtype=torch.float64, device="cuda")
    assert wh.analyze_coded(xb.cuda(), fs, block, first_row=1, x_len=lens, number_of_dimensions=nd) == nf
    tpos, f0, sp, ap, _ = wh.analyze(xb.cuda(), fs, x_len=lens)
    torch.cuda.synchronize()
    rec = block.cpu().numpy()
    assert np.isnan(rec[0]).all() and np.isnan(rec[-1]).all()
    row = 1
    for u, x in enumerate(xs):
        k = nf[u]
        r = rec[row:row + k]
        row += k
        # this library's coders on its own dense rows: the same kernels reading other strides -> the same bits
        mc = wh.code_spectral_envelope(sp[u, :k], fs, fft, nd).cpu().numpy()
        bap = wh.code_aperiodicity(ap[u, :k], fs, fft).cpu().numpy()
        assert np.array_equal(r[:, 0], tpos[u, :k].cpu().numpy()) and np.array_equal(r[:, 1], f0[u, :k].cpu().numpy())
        assert np.array_equal(r[:, 2:2 + nd], mc) and np.array_equal(r[:, 2 + nd:], bap)
        # the reference, end to end
        xn = x.numpy()
        tp_o, f0_o = oracle.harvest(xn, fs)
        sp_o = oracle.cheaptrick(xn, fs, tp_o, f0_o, fft_size=fft)
        ap_o = oracle.d4c(xn, fs, tp_o, f0_o, fft)
        mc_o = oracle.code_spectral_envelope(sp_o, fs, fft, nd)
        bap_o = oracle.code_aperiodicity(ap_o, fs, fft)
        assert np.array_equal(r[:, 0], tp_o)
        assert np.max(np.abs(r[:, 2:2 + nd] - mc_o)) <= 1e-4 * np.max(np.abs(mc_o))          # cepstra: relative to the row scale
        assert np.max(np.abs(r[:, 2 + nd:] - bap_o)) <= 1e-4 * max(1.0, np.max(np.abs(bap_o)))   # dB values
