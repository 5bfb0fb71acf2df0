"""Python face of libworld_hip.so (ctypes; the library is the product, this file is
plumbing).

Two levels, mirroring include/world_hip.h:

* ``WorldHip``  -- the batched, device-resident API.  Inputs/outputs are torch
  CUDA(=HIP) tensors of dtype float64; work is enqueued on torch's current
  stream; nothing synchronises.
* module-level ``harvest / dio / stonemask / cheaptrick / d4c`` -- numpy in, numpy
  out, same argument meaning as the reference's C functions of the same names
  (they call the library's drop-in host-pointer entry points).

The extension must exist: importing this module on a machine where
``libworld_hip.so`` has not been built raises immediately (no fallback path).
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# WORLD_HIP_LIB: another build of the library (tools/ab.py times variants built with other flags side by side)
LIB_PATH = os.environ.get("WORLD_HIP_LIB") or os.path.join(HERE, "libworld_hip.so")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


class DioOption(C.Structure):        # include/world_hip.h (reference dio.h:16-23)
    _fields_ = [("f0_floor", C.c_double), ("f0_ceil", C.c_double), ("channels_in_octave", C.c_double),
                ("frame_period", C.c_double), ("speed", C.c_int), ("allowed_range", C.c_double)]


class HarvestOption(C.Structure):    # reference harvest.h:16-20
    _fields_ = [("f0_floor", C.c_double), ("f0_ceil", C.c_double), ("frame_period", C.c_double)]


class CheapTrickOption(C.Structure):  # reference cheaptrick.h:16-20
    _fields_ = [("q1", C.c_double), ("f0_floor", C.c_double), ("fft_size", C.c_int)]


class D4COption(C.Structure):        # reference d4c.h:16-18
    _fields_ = [("threshold", C.c_double)]


def _hip_runtime_first():
    """torch ships its own libamdhip64; load it BEFORE libworld_hip.so so the process
    ends up with exactly one HIP runtime (the one torch allocates device memory with)."""
    try:
        import torch  # noqa: F401
    except ImportError:
        pass


def load_library(path=LIB_PATH):
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing: build it with `python -m world_amd.build` "
                          "(hipcc, gfx950). There is no non-GPU fallback.")
    _hip_runtime_first()
    lib = C.CDLL(path)
    # the batched ABI this binding was written against (include/world_hip.h: WORLD_HIP_ABI_VERSION).  Libraries of round 5
    # (tools/ab.py loads those) have the same prototypes and no version symbol; anything else is refused rather than called
    # with shifted arguments
    abi = lib.world_hip_abi_version() if hasattr(lib, "world_hip_abi_version") else 5
    if abi not in (5, 6):
        raise ImportError(f"{path}: batched ABI version {abi}, this binding speaks 5-6")
    lib.abi_version = abi
    vp = C.c_void_p
    lib.world_hip_create.restype = vp
    lib.world_hip_create.argtypes = [C.c_int, vp]
    lib.world_hip_destroy.argtypes = [vp]
    lib.world_hip_last_error.restype = C.c_char_p
    lib.world_hip_sync.argtypes = [vp]
    lib.world_hip_workspace_bytes.restype = C.c_ulonglong
    lib.world_hip_workspace_bytes.argtypes = [vp]
    lib.world_hip_noise_table_bytes.restype = C.c_ulonglong
    lib.world_hip_noise_table_bytes.argtypes = [vp]
    lib.world_hip_verify_tables.argtypes = [vp]
    lib.world_hip_harvest_batch.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, _ip, C.POINTER(HarvestOption),
                                            C.c_int, vp, vp]
    lib.world_hip_dio_batch.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, _ip, C.POINTER(DioOption),
                                        C.c_int, vp, vp]
    lib.world_hip_stonemask_batch.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, _ip, _ip, C.c_int, vp, vp, vp]
    lib.world_hip_cheaptrick_batch.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, _ip, _ip, C.c_int, vp, vp,
                                               C.POINTER(CheapTrickOption), vp]
    lib.world_hip_d4c_batch.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, _ip, _ip, C.c_int, vp, vp, C.c_int,
                                        C.POINTER(D4COption), vp]
    lib.world_hip_synthesis_batch.argtypes = [vp, C.c_int, C.c_int, C.c_double, C.c_int, _ip, C.c_int, vp, vp, vp,
                                              _ip, C.c_int, vp]
    lib.world_hip_pcm16_to_double.argtypes = [vp, C.c_longlong, vp, vp]
    lib.world_hip_set_synthesis_pulse_capacity.argtypes = [vp, C.c_int]
    lib.world_hip_synthesis_pulses_dropped.argtypes = [vp, _ip]
    lib.world_hip_probe_rfft.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_longlong, vp, vp]
    lib.world_hip_probe_irfft.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_longlong, vp, vp]
    lib.world_hip_analyze_packed.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, _ip, C.POINTER(HarvestOption),
                                             C.POINTER(CheapTrickOption), C.POINTER(D4COption), C.c_longlong, vp, C.c_int]
    lib.world_hip_analyze_sharded.argtypes = [C.c_int, C.POINTER(vp), C.c_int, C.c_int, C.POINTER(vp), _ip,
                                              C.POINTER(HarvestOption), C.POINTER(CheapTrickOption), C.POINTER(D4COption),
                                              C.c_int, C.POINTER(vp), C.c_longlong, C.c_int, C.POINTER(C.c_longlong)]
    lib.world_hip_check_shape.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_int]
    if hasattr(lib, "world_hip_set_hint"):
        lib.world_hip_set_hint.argtypes = [vp, C.c_int]
    if hasattr(lib, "world_hip_probe_machine"):                      # (absent from libraries of earlier rounds: tools/ab.py loads those)
        lib.world_hip_probe_machine.argtypes = [vp, C.POINTER(C.c_double), C.c_int]
    lib.world_hip_record_columns.argtypes = [C.c_int, C.c_int]
    if hasattr(lib, "world_hip_analyze_coded"):
        lib.world_hip_coded_columns.argtypes = [C.c_int, C.c_int]
        lib.world_hip_analyze_coded.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, _ip, C.POINTER(HarvestOption),
                                                C.POINTER(CheapTrickOption), C.POINTER(D4COption), C.c_int, C.c_longlong, vp, C.c_int]
    lib.world_hip_spectral_packed_range.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, _ip, _ip, C.c_int, vp, vp,
                                                    C.POINTER(CheapTrickOption), C.POINTER(D4COption), C.c_int, C.c_int,
                                                    C.c_int, C.c_longlong, vp, C.c_int]
    lib.world_hip_cheaptrick_batch_range.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, _ip, _ip, C.c_int, vp, vp,
                                                     C.POINTER(CheapTrickOption), C.c_int, C.c_int, C.c_int, vp]
    lib.world_hip_d4c_batch_range.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, _ip, _ip, C.c_int, vp, vp, C.c_int,
                                              C.POINTER(D4COption), C.c_int, C.c_int, C.c_int, vp]
    lib.world_hip_analyze_batch.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, _ip, C.POINTER(HarvestOption),
                                            C.POINTER(CheapTrickOption), C.POINTER(D4COption), C.c_int, vp, vp, vp, vp]
    lib.world_hip_graph_begin.argtypes = [vp]
    lib.world_hip_graph_end.argtypes = [vp, C.POINTER(vp)]
    lib.world_hip_graph_launch.argtypes = [vp, vp]
    lib.world_hip_graph_destroy.argtypes = [vp]
    lib.world_hip_pack_results.argtypes = [vp, C.c_int, _ip, C.c_int, C.c_int, vp, vp, vp, vp, C.c_longlong, vp]
    lib.world_hip_unpack_results.argtypes = [vp, C.c_int, _ip, C.c_int, C.c_int, vp, C.c_longlong, vp, vp, vp, vp]
    lib.world_hip_allgather_blocks.argtypes = [C.c_int, C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_longlong), C.c_int,
                                               C.POINTER(vp)]
    lib.world_hip_pcm_to_double.argtypes = [vp, C.c_longlong, C.c_int, vp, vp]
    lib.world_hip_double_to_pcm16.argtypes = [vp, C.c_longlong, vp, vp]
    lib.world_hip_wav_layout.argtypes = [C.c_char_p, _ip, _ip, _ip, C.POINTER(C.c_longlong)]
    lib.world_hip_wav_write_pcm16.argtypes = [C.c_char_p, C.c_int, C.c_longlong, vp]
    lib.world_hip_code_spectral_envelope.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]
    lib.world_hip_decode_spectral_envelope.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]
    lib.world_hip_code_aperiodicity.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp]
    lib.world_hip_decode_aperiodicity.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp]
    lib.GetNumberOfAperiodicities.argtypes = [C.c_int]
    lib.GetFFTSizeForCheapTrick.argtypes = [C.c_int, C.POINTER(CheapTrickOption)]
    lib.world_hip_profile_enable.argtypes = [C.c_int]
    lib.world_hip_profile_collect.argtypes = [C.c_char_p, C.c_int]
    return lib


def frame_count(fs, x_length, frame_period):
    """GetSamplesForHarvest / GetSamplesForDIO."""
    return int(1000.0 * x_length / fs / frame_period) + 1


def cheaptrick_fft_size(fs, f0_floor=71.0):
    import math
    return int(2.0 ** (1.0 + int(math.log(3.0 * fs / f0_floor + 1) / 0.69314718055994529)))


def _ints(v):
    a = np.ascontiguousarray(v, dtype=np.int32)
    return a, a.ctypes.data_as(_ip)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(_dp)


def _rows(a):
    """double** view over a dense 2-D array (the reference's row-pointer ABI).  The row addresses are computed by numpy
    and handed over as one pointer array (a Python loop over 2001 rows cost the caller milliseconds per call); the array
    object keeps itself alive on the ctypes pointer."""
    addr = (a.ctypes.data + a.strides[0] * np.arange(a.shape[0], dtype=np.uintp)).astype(np.uintp)
    ptrs = addr.ctypes.data_as(C.POINTER(_dp))
    ptrs._keep = addr
    return ptrs


class HostAPI:
    """numpy binding of the reference's 13-symbol C ABI.  It is the
    ctypes stub a user of the reference library would write; it works unchanged on
    libworld_hip.so (default) and on any build of the reference itself."""
    kind = "cabi"

    def __init__(self, path=LIB_PATH, hip_runtime=True):
        if not os.path.exists(path):
            raise ImportError(f"{path} is missing (python -m world_amd.build); there is no fallback.")
        self.path = path
        if hip_runtime:                      # False only for CPU-only libraries (the reference build)
            _hip_runtime_first()
        self.lib = L = C.CDLL(path)
        L.Dio.argtypes = [_dp, C.c_int, C.c_int, C.POINTER(DioOption), _dp, _dp]
        L.Harvest.argtypes = [_dp, C.c_int, C.c_int, C.POINTER(HarvestOption), _dp, _dp]
        L.StoneMask.argtypes = [_dp, C.c_int, C.c_int, _dp, _dp, C.c_int, _dp]
        L.CheapTrick.argtypes = [_dp, C.c_int, C.c_int, _dp, _dp, C.c_int, C.POINTER(CheapTrickOption),
                                 C.POINTER(_dp)]
        L.D4C.argtypes = [_dp, C.c_int, C.c_int, _dp, _dp, C.c_int, C.c_int, C.POINTER(D4COption), C.POINTER(_dp)]
        L.GetSamplesForDIO.argtypes = [C.c_int, C.c_int, C.c_double]
        L.GetSamplesForHarvest.argtypes = [C.c_int, C.c_int, C.c_double]
        L.GetFFTSizeForCheapTrick.argtypes = [C.c_int, C.POINTER(CheapTrickOption)]
        L.GetF0FloorForCheapTrick.argtypes = [C.c_int, C.c_int]
        L.GetF0FloorForCheapTrick.restype = C.c_double
        L.InitializeCheapTrickOption.argtypes = [C.c_int, C.POINTER(CheapTrickOption)]
        rows = C.POINTER(_dp)
        L.Synthesis.argtypes = [_dp, C.c_int, rows, rows, C.c_int, C.c_double, C.c_int, C.c_int, _dp]   # synthesis.h:30
        # codec.h:33-88
        L.GetNumberOfAperiodicities.argtypes = [C.c_int]
        L.CodeAperiodicity.argtypes = [rows, C.c_int, C.c_int, C.c_int, rows]
        L.DecodeAperiodicity.argtypes = [rows, C.c_int, C.c_int, C.c_int, rows]
        L.CodeSpectralEnvelope.argtypes = [rows, C.c_int, C.c_int, C.c_int, C.c_int, rows]
        L.DecodeSpectralEnvelope.argtypes = [rows, C.c_int, C.c_int, C.c_int, C.c_int, rows]

    def frame_count(self, fs, n, frame_period):
        return frame_count(fs, n, frame_period)

    def cheaptrick_fft_size(self, fs, f0_floor=71.0):
        return cheaptrick_fft_size(fs, f0_floor)

    def harvest(self, x, fs, f0_floor=71.0, f0_ceil=800.0, frame_period=5.0):
        x = _f64(x)
        opt = HarvestOption(); self.lib.InitializeHarvestOption(C.byref(opt))
        opt.f0_floor, opt.f0_ceil, opt.frame_period = f0_floor, f0_ceil, frame_period
        nf = self.lib.GetSamplesForHarvest(fs, len(x), frame_period)
        tp = np.zeros(nf); f0 = np.zeros(nf)
        self.lib.Harvest(_p(x), len(x), fs, C.byref(opt), _p(tp), _p(f0))
        return tp, f0

    def dio(self, x, fs, f0_floor=71.0, f0_ceil=800.0, channels_in_octave=2.0, frame_period=5.0,
            speed=1, allowed_range=0.1):
        x = _f64(x)
        opt = DioOption(); self.lib.InitializeDioOption(C.byref(opt))
        opt.f0_floor, opt.f0_ceil, opt.channels_in_octave = f0_floor, f0_ceil, channels_in_octave
        opt.frame_period, opt.speed, opt.allowed_range = frame_period, speed, allowed_range
        nf = self.lib.GetSamplesForDIO(fs, len(x), frame_period)
        tp = np.zeros(nf); f0 = np.zeros(nf)
        self.lib.Dio(_p(x), len(x), fs, C.byref(opt), _p(tp), _p(f0))
        return tp, f0

    def stonemask(self, x, fs, tp, f0):
        x, tp, f0 = _f64(x), _f64(tp), _f64(f0)
        out = np.zeros(len(f0))
        self.lib.StoneMask(_p(x), len(x), fs, _p(tp), _p(f0), len(f0), _p(out))
        return out

    def cheaptrick(self, x, fs, tp, f0, q1=-0.15, f0_floor=71.0, fft_size=None):
        x, tp, f0 = _f64(x), _f64(tp), _f64(f0)
        opt = CheapTrickOption(); self.lib.InitializeCheapTrickOption(fs, C.byref(opt))
        opt.q1, opt.f0_floor = q1, f0_floor
        opt.fft_size = fft_size or self.lib.GetFFTSizeForCheapTrick(fs, C.byref(opt))
        sp = np.zeros((len(f0), opt.fft_size // 2 + 1))
        self.lib.CheapTrick(_p(x), len(x), fs, _p(tp), _p(f0), len(f0), C.byref(opt), _rows(sp))
        return sp

    def d4c(self, x, fs, tp, f0, fft_size, threshold=0.85):
        x, tp, f0 = _f64(x), _f64(tp), _f64(f0)
        opt = D4COption(); self.lib.InitializeD4COption(C.byref(opt))
        opt.threshold = threshold
        ap = np.zeros((len(f0), fft_size // 2 + 1))
        self.lib.D4C(_p(x), len(x), fs, _p(tp), _p(f0), len(f0), fft_size, C.byref(opt), _rows(ap))
        return ap

    # -- synthesis (reference synthesis.h:30) --
    def synthesis(self, f0, sp, ap, fft_size, frame_period, fs, y_length):
        f0, sp, ap = _f64(f0), _f64(sp), _f64(ap)
        y = np.zeros(y_length)
        self.lib.Synthesis(_p(f0), len(f0), _rows(sp), _rows(ap), fft_size, frame_period, fs, y_length, _p(y))
        return y

    # -- codec (reference codec.h) --
    def number_of_aperiodicities(self, fs):
        return self.lib.GetNumberOfAperiodicities(fs)

    def code_aperiodicity(self, ap, fs, fft_size):
        ap = _f64(ap)
        out = np.zeros((ap.shape[0], self.number_of_aperiodicities(fs)))
        self.lib.CodeAperiodicity(_rows(ap), ap.shape[0], fs, fft_size, _rows(out))
        return out

    def decode_aperiodicity(self, coded, fs, fft_size):
        coded = _f64(coded)
        out = np.zeros((coded.shape[0], fft_size // 2 + 1))
        self.lib.DecodeAperiodicity(_rows(coded), coded.shape[0], fs, fft_size, _rows(out))
        return out

    def code_spectral_envelope(self, sp, fs, fft_size, ndim):
        sp = _f64(sp)
        out = np.zeros((sp.shape[0], ndim))
        self.lib.CodeSpectralEnvelope(_rows(sp), sp.shape[0], fs, fft_size, ndim, _rows(out))
        return out

    def decode_spectral_envelope(self, coded, fs, fft_size):
        coded = _f64(coded)
        out = np.zeros((coded.shape[0], fft_size // 2 + 1))
        self.lib.DecodeSpectralEnvelope(_rows(coded), coded.shape[0], fs, fft_size, coded.shape[1], _rows(out))
        return out


class FileAPI:
    """numpy binding of the reference's tools/ library (tools/audioio.h, tools/parameterio.h;
    SURVEY.md 8f.2): WAV and F0 / SPEC / AP files.  Like HostAPI it works unchanged on
    libworld_hip.so (default) and on a build of the reference's own tools."""

    def __init__(self, path=LIB_PATH, hip_runtime=True):
        if not os.path.exists(path):
            raise ImportError(f"{path} is missing (python -m world_amd.build); there is no fallback.")
        if hip_runtime:
            _hip_runtime_first()
        self.lib = L = C.CDLL(path)
        rows, s = C.POINTER(_dp), C.c_char_p
        L.wavwrite.argtypes = [_dp, C.c_int, C.c_int, C.c_int, s]
        L.GetAudioLength.argtypes = [s]
        L.wavread.argtypes = [s, _ip, _ip, _dp]
        L.WriteF0.argtypes = [s, C.c_int, C.c_double, _dp, _dp, C.c_int]
        L.ReadF0.argtypes = [s, _dp, _dp]
        L.GetHeaderInformation.argtypes = [s, s]
        L.GetHeaderInformation.restype = C.c_double
        for name in ("WriteSpectralEnvelope", "WriteAperiodicity"):
            getattr(L, name).argtypes = [s, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, rows]
        for name in ("ReadSpectralEnvelope", "ReadAperiodicity"):
            getattr(L, name).argtypes = [s, rows]

    @staticmethod
    def _s(path):
        return os.fsencode(path)

    def audio_length(self, path):
        return self.lib.GetAudioLength(self._s(path))

    def wavread(self, path):
        """-> (x, fs, nbit), or None where the reference prints an error and leaves x untouched."""
        n = self.audio_length(path)
        if n <= 0:
            return None
        x = np.zeros(n)
        fs, nbit = C.c_int(0), C.c_int(0)
        self.lib.wavread(self._s(path), C.byref(fs), C.byref(nbit), _p(x))
        return x, fs.value, nbit.value

    def wavwrite(self, path, x, fs, nbit=16):
        x = _f64(x)
        self.lib.wavwrite(_p(x), len(x), fs, nbit, self._s(path))

    def write_f0(self, path, frame_period, tpos, f0, text=False):
        tpos, f0 = _f64(tpos), _f64(f0)
        self.lib.WriteF0(self._s(path), len(f0), frame_period, _p(tpos), _p(f0), 1 if text else 0)

    def header(self, path, parameter):
        return self.lib.GetHeaderInformation(self._s(path), parameter.encode())

    def read_f0(self, path):
        n = int(self.header(path, "NOF "))
        tpos, f0 = np.zeros(n), np.zeros(n)
        return (tpos, f0) if self.lib.ReadF0(self._s(path), _p(tpos), _p(f0)) == 1 else None

    def _write_matrix(self, fn, path, m, fs, frame_period, fft_size, number_of_dimensions):
        m = _f64(m)
        fn(self._s(path), fs, m.shape[0], frame_period, fft_size, number_of_dimensions, _rows(m))

    def _read_matrix(self, fn, path):
        rows, fft, nod = (int(self.header(path, k)) for k in ("NOF ", "FFT ", "NOD "))
        m = np.zeros((rows, nod if nod else fft // 2 + 1))
        return m if fn(self._s(path), _rows(m)) == 1 else None

    def write_spectral_envelope(self, path, sp, fs, frame_period, fft_size, number_of_dimensions=0):
        self._write_matrix(self.lib.WriteSpectralEnvelope, path, sp, fs, frame_period, fft_size, number_of_dimensions)

    def write_aperiodicity(self, path, ap, fs, frame_period, fft_size, number_of_dimensions=0):
        self._write_matrix(self.lib.WriteAperiodicity, path, ap, fs, frame_period, fft_size, number_of_dimensions)

    def read_spectral_envelope(self, path):
        return self._read_matrix(self.lib.ReadSpectralEnvelope, path)

    def read_aperiodicity(self, path):
        return self._read_matrix(self.lib.ReadAperiodicity, path)


class Graph:
    """a captured sequence of batched calls (WorldHip.capture): launch() replays it on the stream it was captured on"""

    def __init__(self, wh, ctx, handle):
        self.wh, self.ctx, self.handle = wh, ctx, handle

    def launch(self):
        self.wh._check(self.wh.lib.world_hip_graph_launch(self.ctx, self.handle), "graph_launch")

    def close(self):
        if self.handle:
            self.wh.lib.world_hip_graph_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class WorldHip:
    """Batched analysis on one GPU.  Tensors: x [B, L] float64 on the GPU."""

    def __init__(self, device=None, lib_path=LIB_PATH, shared_device=False):
        """shared_device: other jobs run on this GPU at the same time (world_hip.h: WORLD_HIP_HINT_SHARED_DEVICE) --
        single-utterance calls keep the narrow launch shapes.  Results never depend on it."""
        import torch
        self.torch = torch
        self.shared_device = bool(shared_device)
        if not torch.cuda.is_available():
            raise RuntimeError("WorldHip needs a GPU (torch.cuda.is_available() is False)")
        self.lib = load_library(lib_path)
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self._ctxs = {}                  # stream handle -> library context

    def _context(self):
        """One library context (workspace + stream binding) per torch stream in use, created lazily and kept:
        code that alternates streams on one WorldHip keeps one context per stream."""
        s = self.torch.cuda.current_stream(self.device).cuda_stream
        ctx = self._ctxs.get(s)
        if ctx is None:
            ctx = self.lib.world_hip_create(self.device.index, C.c_void_p(s))
            if not ctx:
                raise RuntimeError("world_hip_create: " + self.lib.world_hip_last_error().decode())
            if self.shared_device and hasattr(self.lib, "world_hip_set_hint"):
                self.lib.world_hip_set_hint(C.c_void_p(ctx), 1)
            self._ctxs[s] = ctx
        return ctx

    @property
    def ctx(self):
        """the context of torch's current stream (None before the first call on it)"""
        return self._ctxs.get(self.torch.cuda.current_stream(self.device).cuda_stream)

    def close(self):
        for ctx in self._ctxs.values():
            self.lib.world_hip_destroy(ctx)
        self._ctxs = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what}: {self.lib.world_hip_last_error().decode()}")

    def _prep(self, x, x_len):
        t = self.torch
        assert x.dtype == t.float64 and x.is_cuda and x.dim() == 2 and x.is_contiguous()
        assert x.device == self.device, f"x lives on {x.device}, this WorldHip on {self.device}"
        B, L = x.shape
        if x_len is None:
            x_len = [L] * B
        return B, L, np.ascontiguousarray(x_len, dtype=np.int32)

    def profile(self, fn):
        """Run fn() with per-kernel HIP-event timing; returns {kernel: [ms per launch, ...]}."""
        self.lib.world_hip_profile_enable(1)
        try:
            fn()
        finally:
            self.lib.world_hip_profile_enable(0)
        buf = C.create_string_buffer(1 << 22)
        if self.lib.world_hip_profile_collect(buf, len(buf)) < 0:
            raise RuntimeError("profile: " + self.lib.world_hip_last_error().decode())
        out = {}
        for line in buf.value.decode().splitlines():
            name, ms = line.split()
            out.setdefault(name, []).append(float(ms))
        return out

    def workspace_bytes(self):
        """bytes of device workspace held by this object's contexts (their arenas)"""
        return sum(int(self.lib.world_hip_workspace_bytes(c)) for c in self._ctxs.values())

    def noise_table_bytes(self):
        """bytes of the device's shared randn table (one per device and process)"""
        return int(self.lib.world_hip_noise_table_bytes(self._context()))

    def verify_tables(self):
        """re-reduce the shared randn table on the device and compare with the host's sums; True = intact"""
        return self.lib.world_hip_verify_tables(self._context()) == 0

    # ---- F0 ----
    def harvest(self, x, fs, x_len=None, f0_floor=71.0, f0_ceil=800.0, frame_period=5.0):
        t = self.torch
        B, L, xl = self._prep(x, x_len)
        nf = np.array([frame_count(fs, int(n), frame_period) for n in xl], dtype=np.int32)
        F = int(nf.max())
        tpos = t.zeros((B, F), dtype=t.float64, device=x.device)
        f0 = t.zeros((B, F), dtype=t.float64, device=x.device)
        opt = HarvestOption(f0_floor, f0_ceil, frame_period)
        self._check(self.lib.world_hip_harvest_batch(self._context(), B, fs, x.data_ptr(), L,
                                                     xl.ctypes.data_as(_ip), C.byref(opt), F,
                                                     tpos.data_ptr(), f0.data_ptr()), "harvest")
        return tpos, f0, nf

    def dio(self, x, fs, x_len=None, f0_floor=71.0, f0_ceil=800.0, channels_in_octave=2.0, frame_period=5.0,
            speed=1, allowed_range=0.1):
        t = self.torch
        B, L, xl = self._prep(x, x_len)
        nf = np.array([frame_count(fs, int(n), frame_period) for n in xl], dtype=np.int32)
        F = int(nf.max())
        tpos = t.zeros((B, F), dtype=t.float64, device=x.device)
        f0 = t.zeros((B, F), dtype=t.float64, device=x.device)
        opt = DioOption(f0_floor, f0_ceil, channels_in_octave, frame_period, speed, allowed_range)
        self._check(self.lib.world_hip_dio_batch(self._context(), B, fs, x.data_ptr(), L, xl.ctypes.data_as(_ip),
                                                 C.byref(opt), F, tpos.data_ptr(), f0.data_ptr()), "dio")
        return tpos, f0, nf

    def stonemask(self, x, fs, tpos, f0, n_frames, x_len=None):
        t = self.torch
        B, L, xl = self._prep(x, x_len)
        nf = np.ascontiguousarray(n_frames, dtype=np.int32)
        out = t.zeros_like(f0)
        self._check(self.lib.world_hip_stonemask_batch(self._context(), B, fs, x.data_ptr(), L,
                                                       xl.ctypes.data_as(_ip), nf.ctypes.data_as(_ip), f0.shape[1],
                                                       tpos.data_ptr(), f0.data_ptr(), out.data_ptr()), "stonemask")
        return out

    # ---- spectral stages ----
    def cheaptrick(self, x, fs, tpos, f0, n_frames, x_len=None, q1=-0.15, f0_floor=71.0, fft_size=None, out=None):
        t = self.torch
        B, L, xl = self._prep(x, x_len)
        nf = np.ascontiguousarray(n_frames, dtype=np.int32)
        fft_size = fft_size or cheaptrick_fft_size(fs, f0_floor)
        F = f0.shape[1]
        sp = out if out is not None else t.zeros((B, F, fft_size // 2 + 1), dtype=t.float64, device=x.device)
        opt = CheapTrickOption(q1, f0_floor, fft_size)
        self._check(self.lib.world_hip_cheaptrick_batch(self._context(), B, fs, x.data_ptr(), L,
                                                        xl.ctypes.data_as(_ip), nf.ctypes.data_as(_ip), F,
                                                        tpos.data_ptr(), f0.data_ptr(), C.byref(opt),
                                                        sp.data_ptr()), "cheaptrick")
        return sp

    def d4c(self, x, fs, tpos, f0, n_frames, fft_size, x_len=None, threshold=0.85, out=None):
        t = self.torch
        B, L, xl = self._prep(x, x_len)
        nf = np.ascontiguousarray(n_frames, dtype=np.int32)
        F = f0.shape[1]
        ap = out if out is not None else t.zeros((B, F, fft_size // 2 + 1), dtype=t.float64, device=x.device)
        opt = D4COption(threshold)
        self._check(self.lib.world_hip_d4c_batch(self._context(), B, fs, x.data_ptr(), L, xl.ctypes.data_as(_ip),
                                                 nf.ctypes.data_as(_ip), F, tpos.data_ptr(), f0.data_ptr(),
                                                 fft_size, C.byref(opt), ap.data_ptr()), "d4c")
        return ap

    def synthesis(self, f0, sp, ap, n_frames, fft_size, frame_period, fs, y_length, check_pulses=True):
        """reference Synthesis() on a batch: f0 [B, F], sp / ap [B, F, fft/2+1] (float64, device);
        n_frames and y_length are per-utterance ints; returns y [B, max(y_length)].  check_pulses=False skips the (synchronising)
        check that every pitch pulse found room in the workspace -- the caller then asks synthesis_pulses_dropped() itself"""
        t = self.torch
        f0, sp, ap = f0.contiguous(), sp.contiguous(), ap.contiguous()
        B, F = f0.shape
        nf = np.ascontiguousarray(np.broadcast_to(n_frames, (B,)), dtype=np.int32)
        yl = np.ascontiguousarray(np.broadcast_to(y_length, (B,)), dtype=np.int32)
        Y = int(yl.max())
        y = t.zeros((B, Y), dtype=t.float64, device=f0.device)
        def run():
            self._check(self.lib.world_hip_synthesis_batch(self._context(), B, fs, float(frame_period), fft_size,
                                                           nf.ctypes.data_as(_ip), F, f0.data_ptr(), sp.data_ptr(),
                                                           ap.data_ptr(), yl.ctypes.data_as(_ip), Y, y.data_ptr()),
                        "synthesis")
        run()
        if check_pulses:
            # the pulse count is data dependent and only known on the device: a call that had no room for some says so
            # (this synchronises); repeat it once with exactly the capacity it asked for -- never a silently truncated waveform
            need = self.synthesis_pulses_dropped()
            if need:
                if need > Y:
                    raise RuntimeError(f"synthesis: {need} pitch pulses for {Y} output samples")
                self.set_synthesis_pulse_capacity(need + 16)
                try:
                    run()
                    if self.synthesis_pulses_dropped():
                        raise RuntimeError("synthesis: pulses dropped even at the requested capacity")
                finally:
                    self.set_synthesis_pulse_capacity(0)
        return y

    # ---- the per-frame FFT in isolation (include/world_hip.h: world_hip_probe_rfft) ----
    def probe_rfft(self, x, max_lr=3, threads=0, out=None, static_plan=False):
        """x [batch, N] float64 -> [batch, N/2+1, 2] (re, im) by csrc/fft.h's block_rfft, one workgroup per row"""
        t = self.torch
        assert x.dtype == t.float64 and x.dim() == 2 and x.is_contiguous()
        batch, N = x.shape
        lg = N.bit_length() - 1
        assert 1 << lg == N
        out = out if out is not None else t.empty((batch, N // 2 + 1, 2), dtype=t.float64, device=x.device)
        self._check(self.lib.world_hip_probe_rfft(self._context(), lg, max_lr, threads, int(static_plan), batch, x.data_ptr(),
                                                  out.data_ptr()),
                    "probe_rfft")
        return out

    def probe_irfft(self, spec, max_lr=3, threads=0, out=None, static_plan=False):
        """spec [batch, N/2+1, 2] -> [batch, N] = N * irfft (the reference's unscaled c2r) by block_irfft"""
        t = self.torch
        assert spec.dtype == t.float64 and spec.dim() == 3 and spec.is_contiguous()
        batch, N = spec.shape[0], 2 * (spec.shape[1] - 1)
        lg = N.bit_length() - 1
        assert 1 << lg == N
        out = out if out is not None else t.empty((batch, N), dtype=t.float64, device=spec.device)
        self._check(self.lib.world_hip_probe_irfft(self._context(), lg, max_lr, threads, int(static_plan), batch, spec.data_ptr(),
                                                   out.data_ptr()),
                    "probe_irfft")
        return out

    # ---- multi-GPU exchange records (include/world_hip.h: world_hip_pack_results) ----
    def capture(self, fn):
        """Run fn() -- batched calls of this object on torch's current stream, on preallocated tensors, shapes that ran before --
        inside a HIP graph capture; returns a Graph whose launch() replays all of it with one host call."""
        ctx = self._context()
        self._check(self.lib.world_hip_graph_begin(ctx), "graph_begin")
        g = C.c_void_p()
        try:
            fn()
        except BaseException:
            # the capture must be ended whatever fn() did; a graph it still produced is destroyed, not leaked
            if self.lib.world_hip_graph_end(ctx, C.byref(g)) == 0 and g:
                self.lib.world_hip_graph_destroy(g)
            raise
        self._check(self.lib.world_hip_graph_end(ctx, C.byref(g)), "graph_end")
        return Graph(self, ctx, g)

    def analyze_packed(self, x, fs, block, first_row=0, x_len=None, frame_period=5.0, f0_floor=71.0, f0_ceil=800.0,
                       q1=-0.15, threshold=0.85):
        """Harvest -> CheapTrick -> D4C of one batch written straight into packed records (no dense sp / ap, no pack
        pass): utterance u's n_frames[u] records start at block[first_row + sum(n_frames[:u])].  Returns n_frames."""
        t = self.torch
        B, L, xl = self._prep(x, x_len)
        fft_size = cheaptrick_fft_size(fs, 71.0)
        nb = fft_size // 2 + 1
        nf = [frame_count(fs, int(n), frame_period) for n in xl]
        cols = block.shape[-1]                       # 2 + 2 nb: f64 records; 2 + nb: the spectra as float32 (narrow wire)
        assert block.dtype == t.float64 and block.is_contiguous() and cols in (2 + 2 * nb, 2 + nb) and block.device == x.device
        assert first_row >= 0 and first_row + sum(nf) <= block.shape[0]
        hopt, copt, dopt = HarvestOption(f0_floor, f0_ceil, frame_period), CheapTrickOption(q1, 71.0, fft_size), D4COption(threshold)
        self._check(self.lib.world_hip_analyze_packed(self._context(), B, fs, x.data_ptr(), L, xl.ctypes.data_as(_ip),
                                                      C.byref(hopt), C.byref(copt), C.byref(dopt), first_row,
                                                      block.data_ptr(), cols), "analyze_packed")
        return nf

    def analyze_coded(self, x, fs, block, first_row=0, x_len=None, frame_period=5.0, f0_floor=71.0, f0_ceil=800.0,
                      q1=-0.15, threshold=0.85, number_of_dimensions=60):
        """Harvest -> CheapTrick -> D4C of one batch written as CODED records [tpos, f0, mel-cepstrum[D], band
        aperiodicity[nap]] (include/world_hip.h: world_hip_analyze_coded -- the reference's CodeSpectralEnvelope /
        CodeAperiodicity of the analysis, 31 x fewer bytes per frame at 48 kHz).  Returns n_frames."""
        t = self.torch
        B, L, xl = self._prep(x, x_len)
        fft_size = cheaptrick_fft_size(fs, 71.0)
        nf = [frame_count(fs, int(n), frame_period) for n in xl]
        cols = block.shape[-1]
        assert block.dtype == t.float64 and block.is_contiguous() and block.device == x.device
        assert cols == self.lib.world_hip_coded_columns(fs, number_of_dimensions), (cols, fs, number_of_dimensions)
        assert first_row >= 0 and first_row + sum(nf) <= block.shape[0]
        hopt, copt, dopt = HarvestOption(f0_floor, f0_ceil, frame_period), CheapTrickOption(q1, 71.0, fft_size), D4COption(threshold)
        self._check(self.lib.world_hip_analyze_coded(self._context(), B, fs, x.data_ptr(), L, xl.ctypes.data_as(_ip),
                                                     C.byref(hopt), C.byref(copt), C.byref(dopt), number_of_dimensions,
                                                     first_row, block.data_ptr(), cols), "analyze_coded")
        return nf

    def spectral_packed_range(self, x, fs, tpos, f0, n_frames, block, frame_lo, frame_hi, first_row=0, x_len=None, q1=-0.15,
                              threshold=0.85, reuse_offsets=False):
        """CheapTrick + D4C of frames [frame_lo, frame_hi) of every utterance, given F0, straight into packed records
        (include/world_hip.h: world_hip_spectral_packed_range): bit-identical to the same rows of a whole-utterance call.
        reuse_offsets: an earlier call on this context had the same inputs and only another range -- its offset scans and
        LoveTrain pass are reused (the library checks that they are still there and were made for these buffers).
        Returns the number of records written."""
        t = self.torch
        B, L, xl = self._prep(x, x_len)
        fft_size = cheaptrick_fft_size(fs, 71.0)
        nb = fft_size // 2 + 1
        nf = np.ascontiguousarray(n_frames, dtype=np.int32)
        F = tpos.shape[1]
        cols = block.shape[-1]
        rows = int(sum(min(frame_hi, int(n)) - min(frame_lo, int(n)) for n in nf))
        assert block.dtype == t.float64 and block.is_contiguous() and cols in (2 + 2 * nb, 2 + nb) and block.device == x.device
        assert tpos.is_contiguous() and f0.is_contiguous() and tpos.shape == f0.shape == (B, F)
        assert 0 <= frame_lo <= frame_hi and first_row >= 0 and first_row + rows <= block.shape[0]
        copt, dopt = CheapTrickOption(q1, 71.0, fft_size), D4COption(threshold)
        self._check(self.lib.world_hip_spectral_packed_range(self._context(), B, fs, x.data_ptr(), L, xl.ctypes.data_as(_ip),
                                                             nf.ctypes.data_as(_ip), F, tpos.data_ptr(), f0.data_ptr(),
                                                             C.byref(copt), C.byref(dopt), frame_lo, frame_hi,
                                                             1 if reuse_offsets else 0, first_row, block.data_ptr(), cols),
                    "spectral_packed_range")
        return rows

    def pack_results(self, tpos, f0, sp, ap, n_frames, block, first_row=0):
        """valid frames of a batched analysis -> records [tpos, f0, sp row, ap row] in block[first_row:] (device)"""
        t = self.torch
        B, F = f0.shape
        nb = sp.shape[-1]
        nf = np.ascontiguousarray(n_frames, dtype=np.int32)
        assert block.dtype == t.float64 and block.is_contiguous() and block.shape[-1] == 2 + 2 * nb
        assert first_row + int(nf.sum()) <= block.shape[0]
        assert all(a.is_contiguous() and a.dtype == t.float64 for a in (tpos, f0, sp, ap))
        self._check(self.lib.world_hip_pack_results(self._context(), B, nf.ctypes.data_as(_ip), F, nb, tpos.data_ptr(),
                                                    f0.data_ptr(), sp.data_ptr(), ap.data_ptr(), first_row,
                                                    block.data_ptr()), "pack_results")
        return block

    def unpack_results(self, block, n_frames, first_row=0):
        """the inverse: (tpos [B, F], f0 [B, F], sp [B, F, nb], ap [B, F, nb]) padded to the longest utterance"""
        t = self.torch
        nf = np.ascontiguousarray(n_frames, dtype=np.int32)
        B, F, nb = len(nf), int(nf.max()), (block.shape[-1] - 2) // 2
        tpos = t.zeros((B, F), dtype=t.float64, device=block.device)
        f0 = t.zeros_like(tpos)
        sp = t.zeros((B, F, nb), dtype=t.float64, device=block.device)
        ap = t.zeros_like(sp)
        self._check(self.lib.world_hip_unpack_results(self._context(), B, nf.ctypes.data_as(_ip), F, nb, block.data_ptr(),
                                                      first_row, tpos.data_ptr(), f0.data_ptr(), sp.data_ptr(),
                                                      ap.data_ptr()), "unpack_results")
        return tpos, f0, sp, ap

    def synthesis_pulses_dropped(self):
        """pulses per utterance the last synthesis calls had no room for (0 = all rendered); synchronises"""
        need = C.c_int(0)
        self._check(self.lib.world_hip_synthesis_pulses_dropped(self._context(), C.byref(need)), "synthesis_pulses_dropped")
        return need.value

    def set_synthesis_pulse_capacity(self, pulses_per_utterance):
        self._check(self.lib.world_hip_set_synthesis_pulse_capacity(self._context(), int(pulses_per_utterance)),
                    "set_synthesis_pulse_capacity")

    def pcm16_to_double(self, pcm):
        """int16 samples (any shape) -> float64 / 32768, wavread()'s convention, on the device"""
        t = self.torch
        assert pcm.dtype == t.int16
        pcm = pcm.contiguous()
        x = t.empty(pcm.shape, dtype=t.float64, device=pcm.device)
        self._check(self.lib.world_hip_pcm16_to_double(self._context(), pcm.numel(), pcm.data_ptr(), x.data_ptr()),
                    "pcm16_to_double")
        return x

    def pcm_to_double(self, pcm_bytes, nbit):
        """uint8 tensor holding a WAV file's sample bytes (nbit/8 per sample, little endian) ->
        float64 samples exactly as wavread() decodes them, on the device"""
        t = self.torch
        assert pcm_bytes.dtype == t.uint8 and pcm_bytes.dim() == 1 and nbit // 8 >= 1
        pcm_bytes = pcm_bytes.contiguous()
        n = pcm_bytes.numel() // (nbit // 8)
        x = t.empty(n, dtype=t.float64, device=pcm_bytes.device)
        self._check(self.lib.world_hip_pcm_to_double(self._context(), n, nbit, pcm_bytes.data_ptr(), x.data_ptr()),
                    "pcm_to_double")
        return x

    def double_to_pcm16(self, x):
        """float64 samples -> the int16 values wavwrite() stores, on the device"""
        t = self.torch
        assert x.dtype == t.float64
        x = x.contiguous()
        q = t.empty(x.shape, dtype=t.int16, device=x.device)
        self._check(self.lib.world_hip_double_to_pcm16(self._context(), x.numel(), x.data_ptr(), q.data_ptr()),
                    "double_to_pcm16")
        return q

    def wav_layout(self, path):
        """Host-side header parse: (fs, nbit, samples, byte offset of the samples); raises where wavread() refuses."""
        fs, nbit, n, off = C.c_int(0), C.c_int(0), C.c_int(0), C.c_longlong(0)
        rc = self.lib.world_hip_wav_layout(os.fsencode(path), C.byref(fs), C.byref(nbit), C.byref(n), C.byref(off))
        if rc != 1:
            raise OSError(f"{path}: " + ("cannot be opened" if rc == 0 else "not a mono PCM WAV file"))
        return fs.value, nbit.value, n.value, off.value

    def wavread(self, path):
        """A WAV file -> (x on the device, fs): only the file's own bytes cross PCIe."""
        fs, nbit, n, off = self.wav_layout(path)
        raw = np.fromfile(path, dtype=np.uint8, offset=off, count=n * (nbit // 8))
        if raw.size != n * (nbit // 8):
            raise OSError(f"{path}: shorter than its header says")
        return self.pcm_to_double(self.torch.from_numpy(raw).to(self.device), nbit), fs

    def wavwrite(self, path, x, fs):
        """float64 samples on the device -> a 16-bit WAV file as wavwrite() would store them
        (quantised on the device; int16 crosses PCIe)."""
        q = self.double_to_pcm16(x.reshape(-1)).cpu().numpy()
        if self.lib.world_hip_wav_write_pcm16(os.fsencode(path), fs, q.size, q.ctypes.data) != 1:
            raise OSError(f"{path}: cannot be written")

    # ---- coders (reference codec.h): dense [..., cols] tensors, leading dims are rows ----
    def _codec(self, fn, what, src, fs, fft_size, out_cols, *dims):
        t = self.torch
        src = src.contiguous()
        rows = int(src.numel() // src.shape[-1])
        out = t.empty(tuple(src.shape[:-1]) + (out_cols,), dtype=t.float64, device=src.device)
        self._check(fn(self._context(), rows, fs, fft_size, *dims, src.data_ptr(), out.data_ptr()), what)
        return out

    def code_spectral_envelope(self, sp, fs, fft_size, number_of_dimensions):
        return self._codec(self.lib.world_hip_code_spectral_envelope, "code_spectral_envelope", sp, fs, fft_size,
                           number_of_dimensions, number_of_dimensions)

    def decode_spectral_envelope(self, coded, fs, fft_size):
        return self._codec(self.lib.world_hip_decode_spectral_envelope, "decode_spectral_envelope", coded, fs,
                           fft_size, fft_size // 2 + 1, int(coded.shape[-1]))

    def code_aperiodicity(self, ap, fs, fft_size):
        return self._codec(self.lib.world_hip_code_aperiodicity, "code_aperiodicity", ap, fs, fft_size,
                           self.lib.GetNumberOfAperiodicities(fs))

    def decode_aperiodicity(self, coded, fs, fft_size):
        return self._codec(self.lib.world_hip_decode_aperiodicity, "decode_aperiodicity", coded, fs, fft_size,
                           fft_size // 2 + 1)

    def probe_machine(self):
        """world_hip_probe_machine (include/world_hip.h): ~50 ms of microbenchmarks that characterise the box"""
        if not hasattr(self.lib, "world_hip_probe_machine"):
            return None
        v = (C.c_double * 8)()
        self._check(self.lib.world_hip_probe_machine(self._context(), v, 8), "probe_machine")
        keys = ("sclk_mhz_under_fp64_load", "fp64_fma_tflops", "chase_ns_2gb", "chase_ns_64mb_warm", "chase_ns_1mb_warm",
                "lds_trip_cycles_idle_cu", "lds_trip_cycles_loaded_cu", "compute_units")
        return {k: round(float(x), 2) for k, x in zip(keys, v)}

    def analyze(self, x, fs, x_len=None, f0_method="harvest", frame_period=5.0, f0_floor=71.0, f0_ceil=800.0,
                q1=-0.15, threshold=0.85, sp_out=None, ap_out=None, tpos_out=None, f0_out=None):
        """The north-star pipeline: F0 (Harvest, or DIO+StoneMask) -> CheapTrick -> D4C.
        sp_out / ap_out / tpos_out / f0_out (Harvest route): caller-owned result buffers of the right shape, reused from
        call to call -- the library writes every frame below an utterance's count and nothing else, so what lies beyond
        (the padding of a ragged batch) is the caller's."""
        if f0_method == "harvest":
            # one library call: Harvest, then CheapTrick beside D4C on two streams of the context
            t = self.torch
            B, L, xl = self._prep(x, x_len)
            fft_size = cheaptrick_fft_size(fs, 71.0)
            nb = fft_size // 2 + 1
            nf = np.array([frame_count(fs, int(n), frame_period) for n in xl], dtype=np.int32)
            F = int(nf.max())
            # Fresh buffers are zero-filled only where a ragged batch leaves padding behind the shorter utterances'
            # frames: every frame below nf[u] is written by the stages (hc_output; ct_frame / d4c_finish write whole rows),
            # and a fill is a launch of its own per array and job (round 4's kernel trace: two per job for tpos / f0).
            new = t.zeros if int(nf.min()) < F else t.empty
            tpos = tpos_out if tpos_out is not None else new((B, F), dtype=t.float64, device=x.device)
            f0 = f0_out if f0_out is not None else new((B, F), dtype=t.float64, device=x.device)
            assert tpos.shape == (B, F) and f0.shape == (B, F) and tpos.is_contiguous() and f0.is_contiguous()
            sp = sp_out if sp_out is not None else new((B, F, nb), dtype=t.float64, device=x.device)
            ap = ap_out if ap_out is not None else new((B, F, nb), dtype=t.float64, device=x.device)
            assert sp.shape == (B, F, nb) and ap.shape == (B, F, nb) and sp.is_contiguous() and ap.is_contiguous()
            hopt, copt, dopt = HarvestOption(f0_floor, f0_ceil, frame_period), CheapTrickOption(q1, 71.0, fft_size), D4COption(threshold)
            self._check(self.lib.world_hip_analyze_batch(self._context(), B, fs, x.data_ptr(), L, xl.ctypes.data_as(_ip),
                                                         C.byref(hopt), C.byref(copt), C.byref(dopt), F, tpos.data_ptr(),
                                                         f0.data_ptr(), sp.data_ptr(), ap.data_ptr()), "analyze")
            return tpos, f0, sp, ap, nf
        elif f0_method == "dio":
            tpos, f0_raw, nf = self.dio(x, fs, x_len, f0_floor, f0_ceil, frame_period=frame_period)
            f0 = self.stonemask(x, fs, tpos, f0_raw, nf, x_len)
        else:
            raise ValueError(f0_method)
        fft_size = cheaptrick_fft_size(fs, 71.0)
        sp = self.cheaptrick(x, fs, tpos, f0, nf, x_len, q1=q1, fft_size=fft_size, out=sp_out)
        ap = self.d4c(x, fs, tpos, f0, nf, fft_size, x_len, threshold=threshold, out=ap_out)
        return tpos, f0, sp, ap, nf


def analyze_sharded_c(lib, ctxs, xs, fs, block_ptrs, rows_capacity, sub_batch=32, frame_period=5.0, f0_floor=71.0,
                      f0_ceil=800.0, q1=-0.15, threshold=0.85, wire=0):
    """world_hip_analyze_sharded (include/world_hip.h): ONE process, several contexts (normally one per GPU), the job's
    utterances as host arrays.  ctxs: library contexts; xs: list of 1-D float64 numpy arrays; block_ptrs: one device
    pointer per context to rows_capacity x world_hip_record_columns(fft, wire) doubles (wire 0: f64 records, 1: the
    spectra as float32).  Returns where [n_utt, 3] = (context index, first row, n_frames): every block then holds ALL
    records at those rows."""
    n = len(xs)
    fft_size = cheaptrick_fft_size(fs, 71.0)
    cols = lib.world_hip_record_columns(fft_size, wire)
    xs = [np.ascontiguousarray(x, dtype=np.float64) for x in xs]
    vp = C.c_void_p
    xp = (vp * max(1, n))(*[x.ctypes.data for x in xs])
    xl = np.ascontiguousarray([len(x) for x in xs], dtype=np.int32)
    cp = (vp * len(ctxs))(*ctxs)
    bp = (vp * len(ctxs))(*block_ptrs)
    where = np.zeros((max(1, n), 3), dtype=np.int64)
    hopt, copt, dopt = HarvestOption(f0_floor, f0_ceil, frame_period), CheapTrickOption(q1, 71.0, fft_size), D4COption(threshold)
    rc = lib.world_hip_analyze_sharded(len(ctxs), cp, n, fs, xp, xl.ctypes.data_as(_ip), C.byref(hopt), C.byref(copt),
                                       C.byref(dopt), sub_batch, bp, rows_capacity, cols,
                                       where.ctypes.data_as(C.POINTER(C.c_longlong)))
    if rc != 0:
        raise RuntimeError("analyze_sharded: " + lib.world_hip_last_error().decode())
    return where[:n]
