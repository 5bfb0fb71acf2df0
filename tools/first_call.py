"""Cold start of the drop-in path, measured in a FRESH process (bench.py runs this as a subprocess): the first
Harvest() + CheapTrick() + D4C() of a 10 s, 48 kHz utterance on host pointers -- HIP runtime start-up, code-object
load, context + tables, the randn table's first build and its verification against the host statement, workspace
allocation -- against the same calls repeated.  Prints one JSON object."""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np                                                    # noqa: E402

from world_amd import synth                                           # noqa: E402
from world_amd.api import HostAPI, load_library                       # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
fs = 48000
x = np.ascontiguousarray(synth.vowel(fs, seconds, seed=12345).numpy())
# The HIP runtime's own start-up, paid by ANY GPU program and timed apart: driver + device enumeration + primary context
# (hipInit / hipSetDevice / hipFree), and what the runtime initialises lazily on first use -- the device memory pool
# (first hipMalloc), a hardware queue (first stream), pinned host memory, its built-in kernels and the launch path (a
# memset on the stream), events.  Measured with a bare process: the first of each costs tens of milliseconds whoever
# asks for it, and it used to be booked on the library's first call.
t_rt = time.perf_counter()
try:
    _hip = C.CDLL("libamdhip64.so")
    _hip.hipInit(0)
    _hip.hipSetDevice(0)
    _hip.hipFree(None)
    _p, _h, _s, _e = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
    _hip.hipMalloc(C.byref(_p), C.c_size_t(64 << 20))
    _hip.hipStreamCreateWithFlags(C.byref(_s), 1)
    _hip.hipHostMalloc(C.byref(_h), C.c_size_t(4 << 20), 0)
    _hip.hipMemsetAsync(_p, 0, C.c_size_t(64 << 20), _s)
    _hip.hipMemcpyAsync(_h, _p, C.c_size_t(4 << 20), 2, _s)
    _hip.hipEventCreateWithFlags(C.byref(_e), 2)
    _hip.hipEventRecord(_e, _s)
    _hip.hipStreamSynchronize(_s)
    _hip.hipFree(_p); _hip.hipHostFree(_h)
except OSError:
    _hip = None
hip_runtime_ms = (time.perf_counter() - t_rt) * 1e3
t_load = time.perf_counter()
H = HostAPI()
L = load_library()
L.world_hip_noise_table_build_ms.restype = C.c_double
L.world_hip_noise_table_build_ms.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
t_lib = time.perf_counter()


def job():
    t0 = time.perf_counter()
    tp, f0 = H.harvest(x, fs)
    t1 = time.perf_counter()
    sp = H.cheaptrick(x, fs, tp, f0, fft_size=2048)
    t2 = time.perf_counter()
    ap = H.d4c(x, fs, tp, f0, 2048)
    t3 = time.perf_counter()
    return [(t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3]


first = job()
ctx = L.world_hip_create(0, None)
builds = C.c_int()
table_ms = L.world_hip_noise_table_build_ms(ctx, C.byref(builds))
later = [job() for _ in range(3)]
L.world_hip_destroy(ctx)
print(json.dumps({"seconds": seconds, "hip_runtime_start_ms": hip_runtime_ms, "library_load_ms": (t_lib - t_load) * 1e3,
                  "first_call_note": "first_call_ms excludes hip_runtime_start_ms (timed before it: driver, primary context and the runtime's "
                                     "lazily initialised memory pool / queue / pinned memory / launch path); it holds the library's "
                                     "code-object load, slot + context + tables, workspace allocation, filter-bank set-up and the "
                                     "randn table's first build",
                  "first_call_ms": sum(first), "first_call_stages_ms": first,
                  "randn_table_build_ms": table_ms, "randn_table_builds": builds.value,
                  "steady_call_ms": min(sum(j) for j in later),
                  "table_threads": os.environ.get("WORLD_HIP_TABLE_THREADS", "default: min(16, cores)")}))
