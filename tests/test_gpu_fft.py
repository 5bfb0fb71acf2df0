"""csrc/fft.h in isolation on the GPU (reference src/world/fft.h:22-44, src/fft.cpp:143-212):
block_rfft / block_irfft through the probe entry points against numpy.fft, every size the path uses
(256 .. 16384 points), both plans (radix-8 and radix-16 butterflies), several workgroup sizes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def wh():
    from world_amd.api import WorldHip
    return WorldHip()


def _signals(batch, n, seed):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((batch, n))
    x[0] = 0.0; x[0, 0] = 1.0                                   # impulse: flat spectrum
    x[1] = 1.0                                                  # DC only
    x[2] = np.cos(np.pi * np.arange(n))                         # Nyquist only
    x[3] = np.cos(2 * np.pi * 5 * np.arange(n) / n) * 1e-12     # tiny amplitudes keep their relative accuracy
    return x


@pytest.mark.parametrize("lg", [8, 9, 10, 11, 12, 13, 14])
@pytest.mark.parametrize("max_lr", [3, 4])
def test_block_rfft_matches_numpy(wh, lg, max_lr):
    import torch
    n = 1 << lg
    x = _signals(37, n, lg * 10 + max_lr)
    ref = np.fft.rfft(x, axis=1)
    scale = np.abs(ref).max(axis=1, keepdims=True)
    for threads in (0, 64, 256):
        got = wh.probe_rfft(torch.from_numpy(x).cuda(), max_lr=max_lr, threads=threads).cpu().numpy()
        got = got[..., 0] + 1j * got[..., 1]
        err = np.abs(got - ref) / scale
        assert err.max() < 2e-15 * lg, (lg, max_lr, threads, err.max())
        assert np.all(got[:, 0].imag == 0) and np.all(got[:, -1].imag == 0)      # r2c: DC / Nyquist are real


@pytest.mark.parametrize("lg", [8, 9, 10, 11, 12, 13, 14])
@pytest.mark.parametrize("max_lr", [3, 4])
def test_block_irfft_matches_numpy(wh, lg, max_lr):
    import torch
    n = 1 << lg
    rng = np.random.default_rng(lg + 100 * max_lr)
    spec = rng.standard_normal((21, n // 2 + 1)) + 1j * rng.standard_normal((21, n // 2 + 1))
    ref_in = spec.copy()
    ref_in[:, 0] = ref_in[:, 0].real                             # c2r ignores Im of DC / Nyquist (src/fft.cpp:28-29)
    ref_in[:, -1] = ref_in[:, -1].real
    ref = np.fft.irfft(ref_in, n=n, axis=1) * n                  # the reference's c2r is unscaled
    packed = np.stack([spec.real, spec.imag], axis=-1).copy()
    for threads in (0, 128):
        got = wh.probe_irfft(torch.from_numpy(packed).cuda(), max_lr=max_lr, threads=threads).cpu().numpy()
        err = np.abs(got - ref).max() / np.abs(ref).max()
        assert err < 2e-15 * lg, (lg, max_lr, threads, err)


@pytest.mark.parametrize("lg", [10, 11, 12])
def test_compile_time_plan_instantiations_match_numpy(wh, lg):
    """the instantiations the frame kernels run (length a compile-time constant: static stages, constexpr digit
    reversal) against numpy and, bit for bit, against the run-time-length instantiation of the same plan"""
    import torch
    n = 1 << lg
    x = torch.from_numpy(_signals(29, n, 7 * lg)).cuda()
    ref = np.fft.rfft(x.cpu().numpy(), axis=1)
    got = wh.probe_rfft(x, max_lr=3, static_plan=True)
    g = got.cpu().numpy()
    assert (np.abs(g[..., 0] + 1j * g[..., 1] - ref) / np.abs(ref).max(axis=1, keepdims=True)).max() < 2e-15 * lg
    assert torch.equal(got, wh.probe_rfft(x, max_lr=3, static_plan=False))
    back = wh.probe_irfft(got, max_lr=3, static_plan=True)
    assert torch.equal(back, wh.probe_irfft(got, max_lr=3, static_plan=False))
    assert float((back / n - x).abs().max()) < 1e-14 * lg


def test_roundtrip_is_identity_at_path_sizes(wh):
    """c2r(r2c(x)) = N x: the pair the CheapTrick liftering runs (cheaptrick.cpp:22-57) loses nothing"""
    import torch
    for lg in (10, 11, 12):
        n = 1 << lg
        x = torch.from_numpy(np.random.default_rng(lg).standard_normal((64, n))).cuda()
        y = wh.probe_irfft(wh.probe_rfft(x, max_lr=3), max_lr=3) / n
        assert float((y - x).abs().max()) < 1e-14 * lg
