"""Helper of tests/test_dropin.py (a subprocess, so that the drop-in layer's environment applies and a fork() is safe to try):
   dropin_lifecycle.py <library> <out.npz> [fork]
Harvest + CheapTrick + D4C of one utterance through the drop-in symbols; then world_hip_shutdown() and the same again (a cold
restart must give the same arrays and the slot count must have gone to zero in between); with `fork`, a child forked after
the parent's calls runs the analysis itself (the parent's helper threads and slots do not exist there) and reports through
its exit status."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from world_amd import synth                                          # noqa: E402
from world_amd.api import HostAPI                                    # noqa: E402

lib_path, out = sys.argv[1], sys.argv[2]
do_fork = len(sys.argv) > 3 and sys.argv[3] == "fork"
H = HostAPI(lib_path)
fs = 16000
x = np.ascontiguousarray(synth.utterance(6, fs, 0.6).numpy())


def analyse():
    tp, f0 = H.harvest(x, fs)
    fft = H.cheaptrick_fft_size(fs)
    return tp, f0, H.cheaptrick(x, fs, tp, f0, fft_size=fft), H.d4c(x, fs, tp, f0, fft)


def slots():
    H.lib.world_hip_dropin_stats.argtypes = [C.POINTER(C.c_ulonglong)] * 3
    a, b, c = C.c_ulonglong(), C.c_ulonglong(), C.c_ulonglong()
    H.lib.world_hip_dropin_stats(C.byref(a), C.byref(b), C.byref(c))
    return a.value


first = analyse()
n_before = slots()
H.lib.world_hip_shutdown.restype = C.c_int
rc = H.lib.world_hip_shutdown()
n_after = slots()
second = analyse()
child_ok = -1
if do_fork:
    pid = os.fork()
    if pid == 0:
        try:
            third = analyse()
            ok = all(np.array_equal(a, b) for a, b in zip(first, third))
            os._exit(0 if ok else 3)
        except BaseException:                                        # noqa: BLE001
            os._exit(4)
    _, status = os.waitpid(pid, 0)
    child_ok = os.WEXITSTATUS(status) if os.WIFEXITED(status) else 100 + os.WTERMSIG(status)
    third = analyse()                                                # the parent goes on as before
    assert all(np.array_equal(a, b) for a, b in zip(first, third))
np.savez(out, tp=first[0], f0=first[1], sp=first[2], ap=first[3], tp2=second[0], f02=second[1], sp2=second[2], ap2=second[3],
         slots_before=n_before, slots_after=n_after, shutdown_rc=rc, child=child_ok)
