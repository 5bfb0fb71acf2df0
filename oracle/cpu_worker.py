"""One CPU analysis (Harvest + CheapTrick + D4C) by the oracle, timed -- a worker of
oracle.loader.parallel_analyses (TEST / BENCH-BASELINE INFRASTRUCTURE, never the product).
usage: cpu_worker.py x.npy fs frame_period fft_size start_at [library in oracle/_ref]   -> prints "frames t_start t_end"
start_at = "-": print "ready" once everything is loaded, then read the common start time from stdin (imports on a
cold box take seconds and must not count as analysis time)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402


def main():
    path, fs, frame_period, fft_size = sys.argv[1], int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])
    here = os.path.dirname(os.path.abspath(__file__))
    ref = os.path.join(here, "_ref", sys.argv[6] if len(sys.argv) > 6 else "libworld_ref.so")
    x = np.load(path)
    # bind through the generic C-ABI stub without importing torch-dependent code paths
    from world_amd.api import HostAPI
    if os.path.exists(ref):
        o = HostAPI(ref, hip_runtime=False)
    else:
        from oracle.loader import PortOracle
        o = PortOracle()
    if sys.argv[5] == "-":
        print("ready", flush=True)
        start_at = float(sys.stdin.readline())
    else:
        start_at = float(sys.argv[5])
    while time.time() < start_at:
        time.sleep(0.0005)
    t0 = time.time()
    tp, f0 = o.harvest(x, fs, frame_period=frame_period)
    o.cheaptrick(x, fs, tp, f0, fft_size=fft_size)
    o.d4c(x, fs, tp, f0, fft_size)
    print(len(f0), repr(t0), repr(time.time()))


if __name__ == "__main__":
    main()
