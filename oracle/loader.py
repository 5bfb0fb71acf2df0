"""ctypes loaders for the CPU oracles -- TEST INFRASTRUCTURE ONLY.

Two interchangeable back ends with one Python surface:

* ``PortOracle``  -- oracle/libworld_oracle.so, this repo's plain-C restatement.
* ``RefOracle``   -- oracle/_ref/libworld_ref*.so, the UNMODIFIED reference
  compiled in place from /root/reference by oracle/Makefile (present whenever
  it was built in the container; the .so travels to the GPU box).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module.  Nothing under world_amd/ does.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

HERE = os.path.dirname(os.path.abspath(__file__))
_dp = C.POINTER(C.c_double)


def _p(a):
    return a.ctypes.data_as(_dp)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def build(quiet=True):
    """Compile the restatement (and the in-place reference when available)."""
    subprocess.run(["make", "-s", "-f", os.path.join(HERE, "Makefile")],
                   check=True, stdout=subprocess.DEVNULL if quiet else None)


class _Base:
    kind = "?"

    def frame_count(self, fs, n, frame_period):
        return int(1000.0 * n / fs / frame_period) + 1

    def cheaptrick_fft_size(self, fs, f0_floor=71.0):
        import math
        return int(2.0 ** (1.0 + int(math.log(3.0 * fs / f0_floor + 1) / 0.69314718055994529)))


class PortOracle(_Base):
    kind = "port"

    def __init__(self, path=None):
        path = path or os.path.join(HERE, "libworld_oracle.so")
        if not os.path.exists(path):
            build()
        self.lib = L = C.CDLL(path)
        L.wo_randn.restype = C.c_double
        L.wo_frame_count.restype = C.c_int
        L.wo_frame_count.argtypes = [C.c_int, C.c_int, C.c_double]
        L.wo_harvest.argtypes = [_dp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, _dp, _dp]
        L.wo_dio.argtypes = [_dp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double,
                             C.c_int, C.c_double, _dp, _dp]
        L.wo_stonemask.argtypes = [_dp, C.c_int, C.c_int, _dp, _dp, C.c_int, _dp]
        L.wo_cheaptrick.argtypes = [_dp, C.c_int, C.c_int, _dp, _dp, C.c_int, C.c_double, C.c_int, _dp]
        L.wo_d4c.argtypes = [_dp, C.c_int, C.c_int, _dp, _dp, C.c_int, C.c_int, C.c_double, _dp]
        L.wo_rfft.argtypes = [C.c_int, _dp, _dp, _dp]
        L.wo_irfft_unscaled.argtypes = [C.c_int, _dp, _dp, _dp]
        L.wo_interp1.argtypes = [_dp, _dp, C.c_int, _dp, C.c_int, _dp]
        L.wo_decimate.argtypes = [_dp, C.c_int, C.c_int, _dp]
        L.wo_linear_smoothing.argtypes = [_dp, C.c_double, C.c_int, C.c_int, _dp]
        L.wo_dc_correction.argtypes = [_dp, C.c_double, C.c_int, C.c_int, _dp]
        L.wo_round.argtypes = [C.c_double]
        L.wo_nuttall.argtypes = [C.c_int, _dp]
        L.wo_synthesis.argtypes = [_dp, C.c_int, _dp, _dp, C.c_int, C.c_double, C.c_int, C.c_int, _dp]
        L.wo_number_of_aperiodicities.argtypes = [C.c_int]
        L.wo_code_aperiodicity.argtypes = [_dp, C.c_int, C.c_int, C.c_int, _dp]
        L.wo_decode_aperiodicity.argtypes = [_dp, C.c_int, C.c_int, C.c_int, _dp]
        L.wo_code_spectral_envelope.argtypes = [_dp, C.c_int, C.c_int, C.c_int, C.c_int, _dp]
        L.wo_decode_spectral_envelope.argtypes = [_dp, C.c_int, C.c_int, C.c_int, C.c_int, _dp]

    # -- primitives --
    def randn(self, count):
        st = (C.c_uint32 * 4)()
        self.lib.wo_randn_seed(st)
        return np.array([self.lib.wo_randn(st) for _ in range(count)])

    def rfft(self, x):
        x = _f64(x); n = len(x)
        re = np.zeros(n // 2 + 1); im = np.zeros(n // 2 + 1)
        self.lib.wo_rfft(n, _p(x), _p(re), _p(im))
        return re + 1j * im

    def irfft_unscaled(self, X):
        n = 2 * (len(X) - 1)
        re = _f64(X.real); im = _f64(X.imag); out = np.zeros(n)
        self.lib.wo_irfft_unscaled(n, _p(re), _p(im), _p(out))
        return out

    def interp1(self, x, y, xi):
        x, y, xi = _f64(x), _f64(y), _f64(xi)
        yi = np.zeros(len(xi))
        self.lib.wo_interp1(_p(x), _p(y), len(x), _p(xi), len(xi), _p(yi))
        return yi

    def decimate(self, x, r):
        x = _f64(x)
        y = np.zeros(len(x))
        self.lib.wo_decimate(_p(x), len(x), r, _p(y))
        return y[: (len(x) - 1) // r + 1]

    def linear_smoothing(self, spec, width, fs, fft_size):
        spec = _f64(spec); out = np.zeros(fft_size // 2 + 1)
        self.lib.wo_linear_smoothing(_p(spec), width, fs, fft_size, _p(out))
        return out

    # -- analysis path --
    def harvest(self, x, fs, f0_floor=71.0, f0_ceil=800.0, frame_period=5.0):
        x = _f64(x); nf = self.frame_count(fs, len(x), frame_period)
        tp = np.zeros(nf); f0 = np.zeros(nf)
        self.lib.wo_harvest(_p(x), len(x), fs, f0_floor, f0_ceil, frame_period, _p(tp), _p(f0))
        return tp, f0

    def dio(self, x, fs, f0_floor=71.0, f0_ceil=800.0, channels_in_octave=2.0, frame_period=5.0,
            speed=1, allowed_range=0.1):
        x = _f64(x); nf = self.frame_count(fs, len(x), frame_period)
        tp = np.zeros(nf); f0 = np.zeros(nf)
        self.lib.wo_dio(_p(x), len(x), fs, f0_floor, f0_ceil, channels_in_octave, frame_period,
                        speed, allowed_range, _p(tp), _p(f0))
        return tp, f0

    def stonemask(self, x, fs, tp, f0):
        x, tp, f0 = _f64(x), _f64(tp), _f64(f0)
        out = np.zeros(len(f0))
        self.lib.wo_stonemask(_p(x), len(x), fs, _p(tp), _p(f0), len(f0), _p(out))
        return out

    def cheaptrick(self, x, fs, tp, f0, q1=-0.15, f0_floor=71.0, fft_size=None):
        x, tp, f0 = _f64(x), _f64(tp), _f64(f0)
        fft_size = fft_size or self.cheaptrick_fft_size(fs, f0_floor)
        sp = np.zeros((len(f0), fft_size // 2 + 1))
        self.lib.wo_cheaptrick(_p(x), len(x), fs, _p(tp), _p(f0), len(f0), q1, fft_size, _p(sp))
        return sp

    def d4c(self, x, fs, tp, f0, fft_size, threshold=0.85):
        x, tp, f0 = _f64(x), _f64(tp), _f64(f0)
        ap = np.zeros((len(f0), fft_size // 2 + 1))
        self.lib.wo_d4c(_p(x), len(x), fs, _p(tp), _p(f0), len(f0), fft_size, threshold, _p(ap))
        return ap

    # -- synthesis --
    def synthesis(self, f0, sp, ap, fft_size, frame_period, fs, y_length):
        f0, sp, ap = _f64(f0), _f64(sp), _f64(ap)
        y = np.zeros(y_length)
        self.lib.wo_synthesis(_p(f0), len(f0), _p(sp), _p(ap), fft_size, frame_period, fs, y_length, _p(y))
        return y

    # -- codec --
    def number_of_aperiodicities(self, fs):
        return self.lib.wo_number_of_aperiodicities(fs)

    def code_aperiodicity(self, ap, fs, fft_size):
        ap = _f64(ap); out = np.zeros((ap.shape[0], self.number_of_aperiodicities(fs)))
        self.lib.wo_code_aperiodicity(_p(ap), ap.shape[0], fs, fft_size, _p(out))
        return out

    def decode_aperiodicity(self, coded, fs, fft_size):
        coded = _f64(coded); out = np.zeros((coded.shape[0], fft_size // 2 + 1))
        self.lib.wo_decode_aperiodicity(_p(coded), coded.shape[0], fs, fft_size, _p(out))
        return out

    def code_spectral_envelope(self, sp, fs, fft_size, ndim):
        sp = _f64(sp); out = np.zeros((sp.shape[0], ndim))
        self.lib.wo_code_spectral_envelope(_p(sp), sp.shape[0], fs, fft_size, ndim, _p(out))
        return out

    def decode_spectral_envelope(self, coded, fs, fft_size):
        coded = _f64(coded); out = np.zeros((coded.shape[0], fft_size // 2 + 1))
        self.lib.wo_decode_spectral_envelope(_p(coded), coded.shape[0], fs, fft_size, coded.shape[1], _p(out))
        return out


# The generic binding of the reference's 13-symbol C ABI lives in the product's
# Python mirror (world_amd/api.py: HostAPI); the oracle side only points it at
# the in-place build of the reference.
from world_amd.api import HostAPI as WorldCABI  # noqa: E402


def _cpu_has(flag):
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    return flag in line.split()
    except OSError:
        pass
    return False


class RefOracle(WorldCABI):
    kind = "reference"

    def __init__(self, optimized=False):
        name = "libworld_ref_o3.so" if optimized and _cpu_has("avx2") and _cpu_has("fma") else "libworld_ref.so"
        path = os.path.join(HERE, "_ref", name)
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.flags = "-O3 -march=x86-64-v3" if name.endswith("_o3.so") else "-O1 (reference makefile:6)"
        super().__init__(path, hip_runtime=False)


def ref_available():
    return os.path.exists(os.path.join(HERE, "_ref", "libworld_ref.so"))


def best_oracle():
    """The real reference when its prebuilt .so is present, else the port."""
    return RefOracle() if ref_available() else PortOracle()


# ---- whole-box CPU baseline: one analysis per worker process ---------------------------
def parallel_analyses(x, fs, frame_period, fft_size, procs, timeout=300, optimized=False):
    """Run `procs` full analyses concurrently, each in its own `python oracle/cpu_worker.py`
    process (no GPU runtime, own heap), all starting at the same wall-clock instant.
    Returns (total frames, wall seconds from the common start to the last finish, build flags)."""
    import tempfile
    import time as _t
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "x.npy")
        np.save(path, np.ascontiguousarray(x, dtype=np.float64))
        lib = "libworld_ref_o3.so" if optimized and _cpu_has("avx2") and _cpu_has("fma") else "libworld_ref.so"
        flags = "-O3 -march=x86-64-v3" if lib.endswith("_o3.so") else "-O1 (reference makefile:6)"
        if not os.path.exists(os.path.join(HERE, "_ref", lib)):
            flags = "gcc -O2 restatement"
        cmd = [sys.executable, os.path.join(HERE, "cpu_worker.py"), path, str(fs), str(frame_period), str(fft_size),
               "-", lib]
        ps = [subprocess.Popen(cmd, stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True) for _ in range(procs)]
        for p in ps:                                   # every worker has its library and input loaded ...
            if p.stdout.readline().strip() != "ready":
                raise RuntimeError("cpu_worker did not come up")
        start_at = _t.time() + 0.25                    # ... before the common start is named
        for p in ps:
            p.stdin.write(repr(start_at) + "\n")
            p.stdin.flush()
        frames, t_end = 0, start_at
        for p in ps:
            out, _ = p.communicate(timeout=timeout)
            n, t0, t1 = out.split()[-3:]
            frames += int(n)
            t_end = max(t_end, float(t1))
    return frames, t_end - start_at, flags
