"""The C-ABI library: builds for gfx950 (hipcc cross-compiles without a GPU), loads,
and exports every symbol include/world_hip.h declares.  No compute calls here."""
import ctypes
import os
import re
import shutil

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib_path():
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("hipcc not available")
    from world_amd import build
    return build.build()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "world_hip.h")).read()
    return sorted(set(re.findall(r"WORLD_HIP_API\s+[\w\s\*]+?\b(\w+)\s*\(", text)))


def test_header_declares_the_reference_api():
    names = declared_symbols()
    for must in ["Dio", "InitializeDioOption", "GetSamplesForDIO", "Harvest", "InitializeHarvestOption",
                 "GetSamplesForHarvest", "StoneMask", "CheapTrick", "InitializeCheapTrickOption",
                 "GetFFTSizeForCheapTrick", "GetF0FloorForCheapTrick", "D4C", "InitializeD4COption"]:
        assert must in names          # the 13 symbols of SURVEY.md 8b
    for must in ["GetNumberOfAperiodicities", "CodeAperiodicity", "DecodeAperiodicity", "CodeSpectralEnvelope",
                 "DecodeSpectralEnvelope", "Synthesis"]:
        assert must in names          # every public symbol of codec.o


def test_library_exports_every_declared_symbol(lib_path):
    lib = ctypes.CDLL(lib_path)
    for name in declared_symbols():
        assert hasattr(lib, name), name


def test_option_helpers_are_host_arithmetic(lib_path):
    """these never touch the GPU, so they can be checked here"""
    from world_amd.api import CheapTrickOption, DioOption, HarvestOption, load_library
    lib = load_library(lib_path)
    h = HarvestOption(); lib.InitializeHarvestOption(ctypes.byref(h))
    assert (h.f0_floor, h.f0_ceil, h.frame_period) == (71.0, 800.0, 5.0)
    d = DioOption(); lib.InitializeDioOption(ctypes.byref(d))
    assert (d.f0_floor, d.f0_ceil, d.channels_in_octave, d.frame_period, d.speed, d.allowed_range) == \
        (71.0, 800.0, 2.0, 5.0, 1, 0.1)
    c = CheapTrickOption(); lib.InitializeCheapTrickOption.argtypes = [ctypes.c_int, ctypes.POINTER(CheapTrickOption)]
    lib.InitializeCheapTrickOption(48000, ctypes.byref(c))
    assert (c.q1, c.f0_floor, c.fft_size) == (-0.15, 71.0, 2048)
    lib.InitializeCheapTrickOption(16000, ctypes.byref(c)); assert c.fft_size == 1024
    lib.GetSamplesForHarvest.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_double]
    assert lib.GetSamplesForHarvest(22050, 17500, 5.0) == 159
    lib.GetF0FloorForCheapTrick.restype = ctypes.c_double
    lib.GetF0FloorForCheapTrick.argtypes = [ctypes.c_int, ctypes.c_int]
    assert lib.GetF0FloorForCheapTrick(48000, 2048) == 3.0 * 48000 / (2048 - 3.0)


def test_missing_library_fails_loudly(tmp_path):
    from world_amd.api import HostAPI, load_library
    with pytest.raises(ImportError):
        load_library(str(tmp_path / "nope.so"))
    with pytest.raises(ImportError):
        HostAPI(str(tmp_path / "nope.so"))


def test_batch_tools_refuse_to_run_without_a_gpu(tmp_path):
    """The command-line tools have no CPU path either: without a GPU they stop with an error
    before touching any file (here: CPU-only container; skipped where a GPU is present)."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "world_amd.tools", "analysis", "absent.wav", "--outdir", str(tmp_path / "o")],
                       cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "needs a GPU" in r.stderr
    assert not (tmp_path / "o").exists()


def test_option_and_size_helpers_agree_with_the_reference_build(lib_path):
    """The host-arithmetic helpers (frame counts, FFT sizes, floors, band counts, option defaults) of
    libworld_hip.so against the same symbols of the unmodified reference built in place -- a sweep of
    sampling rates, lengths and frame periods; exact equality (they size every caller-owned buffer)."""
    from oracle.loader import ref_available
    if not ref_available():
        pytest.skip("oracle/_ref not built")
    from world_amd.api import CheapTrickOption, D4COption, DioOption, HarvestOption
    ours = ctypes.CDLL(lib_path)
    ref = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref",
                                   "libworld_ref.so"))
    for L in (ours, ref):
        L.GetSamplesForHarvest.argtypes = L.GetSamplesForDIO.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_double]
        L.GetFFTSizeForCheapTrick.argtypes = [ctypes.c_int, ctypes.POINTER(CheapTrickOption)]
        L.GetF0FloorForCheapTrick.argtypes = [ctypes.c_int, ctypes.c_int]
        L.GetF0FloorForCheapTrick.restype = ctypes.c_double
        L.InitializeCheapTrickOption.argtypes = [ctypes.c_int, ctypes.POINTER(CheapTrickOption)]
        L.GetNumberOfAperiodicities.argtypes = [ctypes.c_int]
    rng = np.random.default_rng(8)
    rates = [8000, 11025, 12000, 16000, 22050, 24000, 32000, 44100, 48000, 88200, 96000, 192000]
    for _ in range(2000):
        fs = int(rng.choice(rates)) if rng.random() < 0.8 else int(rng.integers(4000, 200000))
        n = int(rng.integers(1, 2_000_000))
        fp = float(rng.choice([1.0, 2.5, 5.0, 10.0, 12.5])) if rng.random() < 0.8 else float(rng.uniform(0.5, 20.0))
        assert ours.GetSamplesForHarvest(fs, n, fp) == ref.GetSamplesForHarvest(fs, n, fp)
        assert ours.GetSamplesForDIO(fs, n, fp) == ref.GetSamplesForDIO(fs, n, fp)
        a, b = CheapTrickOption(), CheapTrickOption()
        ours.InitializeCheapTrickOption(fs, ctypes.byref(a)); ref.InitializeCheapTrickOption(fs, ctypes.byref(b))
        assert (a.q1, a.f0_floor, a.fft_size) == (b.q1, b.f0_floor, b.fft_size)
        a.f0_floor = b.f0_floor = float(rng.uniform(20.0, 300.0))
        size = ours.GetFFTSizeForCheapTrick(fs, ctypes.byref(a))
        assert size == ref.GetFFTSizeForCheapTrick(fs, ctypes.byref(b))
        assert ours.GetF0FloorForCheapTrick(fs, size) == ref.GetF0FloorForCheapTrick(fs, size)
        assert ours.GetNumberOfAperiodicities(fs) == ref.GetNumberOfAperiodicities(fs)
    for cls, init in ((DioOption, "InitializeDioOption"), (HarvestOption, "InitializeHarvestOption"), (D4COption, "InitializeD4COption")):
        a, b = cls(), cls()
        getattr(ours, init)(ctypes.byref(a)); getattr(ref, init)(ctypes.byref(b))
        assert bytes(a) == bytes(b), init


def test_forwarding_headers_compile_like_the_reference_tree(tmp_path):
    """include/world/*.h (and the tools/ headers): a caller written against the reference's header tree --
    #include "world/dio.h" ... -- compiles against this repository's include/ alone, as C and as C++, and sees
    every drop-in symbol with the reference's signature (reference src/world/*.h)"""
    import subprocess
    src = tmp_path / "caller.c"
    src.write_text('''
#include "world/dio.h"
#include "world/harvest.h"
#include "world/stonemask.h"
#include "world/cheaptrick.h"
#include "world/d4c.h"
#include "world/codec.h"
#include "world/synthesis.h"
#include "world/macrodefinitions.h"
#include "audioio.h"
#include "parameterio.h"
WORLD_BEGIN_C_DECLS
int uses_everything(const double *x, int n, int fs, double *tp, double *f0, double **sp, double **ap, double *y);
WORLD_END_C_DECLS
int uses_everything(const double *x, int n, int fs, double *tp, double *f0, double **sp, double **ap, double *y) {
  DioOption d; HarvestOption h; CheapTrickOption c; D4COption a;
  InitializeDioOption(&d); InitializeHarvestOption(&h); InitializeCheapTrickOption(fs, &c); InitializeD4COption(&a);
  int nf = GetSamplesForHarvest(fs, n, h.frame_period) + 0 * GetSamplesForDIO(fs, n, d.frame_period);
  Harvest(x, n, fs, &h, tp, f0); Dio(x, n, fs, &d, tp, f0); StoneMask(x, n, fs, tp, f0, nf, f0);
  c.fft_size = GetFFTSizeForCheapTrick(fs, &c); (void)GetF0FloorForCheapTrick(fs, c.fft_size);
  CheapTrick(x, n, fs, tp, f0, nf, &c, sp); D4C(x, n, fs, tp, f0, nf, c.fft_size, &a, ap);
  CodeAperiodicity((const double *const *)ap, nf, fs, c.fft_size, ap); DecodeAperiodicity((const double *const *)ap, nf, fs, c.fft_size, ap);
  CodeSpectralEnvelope((const double *const *)sp, nf, fs, c.fft_size, 4, sp); DecodeSpectralEnvelope((const double *const *)sp, nf, fs, c.fft_size, 4, sp);
  Synthesis(f0, nf, (const double *const *)sp, (const double *const *)ap, c.fft_size, h.frame_period, fs, n, y);
  wavwrite(y, n, fs, 16, "o.wav"); (void)GetAudioLength("o.wav"); WriteF0("o.f0", nf, 5.0, tp, f0, 0);
  return GetNumberOfAperiodicities(fs);
}
''')
    inc = os.path.join(ROOT, "include")
    for cc, std in (("gcc", "-std=c99"), ("g++", "-std=c++11")):
        r = subprocess.run([cc, std, "-x", "c" if cc == "gcc" else "c++", "-Wall", "-Werror", "-fsyntax-only", "-I", inc, str(src)],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def test_shape_limits_are_answered_on_the_host():
    """world_hip_check_shape: the limits of the GPU path as pure host arithmetic -- what the drop-in symbols
    check before any GPU work"""
    import ctypes as C
    from world_amd.api import load_library
    lib = load_library()
    why = C.create_string_buffer(256)
    assert lib.world_hip_check_shape(48000, 2048, why, 256) == 0 and why.value == b""
    assert lib.world_hip_check_shape(16000, 1024, why, 256) == 0
    assert lib.world_hip_check_shape(192000, 8192, why, 256) == 0                 # round 5: StoneMask's indices as bytes, D4C's 16384-point shape
    assert lib.world_hip_check_shape(250000, 8192, why, 256) == 1 and b"StoneMask" in why.value     # the first limit met
    assert lib.world_hip_check_shape(96000, 8192, why, 256) == 0                  # round 4: CheapTrick runs 8192 points
    assert lib.world_hip_check_shape(96000, 16384, why, 256) == 1 and b"CheapTrick" in why.value
    assert lib.world_hip_check_shape(48000, 1000, why, 256) == 1 and b"power of two" in why.value
    assert lib.world_hip_check_shape(120000, 4096, why, 256) == 0
    assert lib.world_hip_check_shape(200000, 8192, why, 256) == 1 and b"D4C" in why.value and b"192 kHz" in why.value
    assert lib.world_hip_check_shape(8000, 512, why, 256) == 1 and b"15.8" in why.value
    assert lib.world_hip_check_shape(96000, 4096, why, 256) == 0


def test_abi_version_and_hint_are_exported_and_consistent(lib_path):
    """round 6: the batched ABI carries a version a binding checks before it binds prototypes; the header's macro
    and the library agree; the launch-geometry hint is part of the ABI and refuses a null context"""
    lib = ctypes.CDLL(lib_path)
    text = open(os.path.join(ROOT, "include", "world_hip.h")).read()
    want = int(re.search(r"#define\s+WORLD_HIP_ABI_VERSION\s+(\d+)", text).group(1))
    assert lib.world_hip_abi_version() == want == 6
    assert int(re.search(r"#define\s+WORLD_HIP_HINT_SHARED_DEVICE\s+(\d+)", text).group(1)) == 1
    lib.world_hip_set_hint.argtypes = [ctypes.c_void_p, ctypes.c_int]
    assert lib.world_hip_set_hint(None, 1) == -1
