"""Build libworld_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m world_amd.build          # incremental
    python -m world_amd.build --force

The shared library is written next to this file (world_amd/libworld_hip.so) so it
travels to the GPU box with the source tree.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "_obj")
LIB = os.path.join(HERE, "libworld_hip.so")
UNITS = ["api.hip", "cheaptrick.hip", "d4c.hip", "codec.hip", "pcm.hip", "synthesis.hip", "harvest.hip", "harvest_contour.hip", "rng_fill.hip", "dio.hip",
         "stonemask.hip", "exchange.hip", "fft_probe.hip", "machine_probe.hip", "tables.cpp", "devrt.cpp"]
# -ffp-contract=off: the analysis is order/rounding sensitive in places;
# fused multiply-adds are requested explicitly (fma()) in the FP64 inner loops instead.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden",
         "-fno-gpu-rdc", "-Wall", "-Wno-unused-function", "-I", CSRC]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _newest_header():
    return max(os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC)
               if f.endswith((".h", ".inc")))


def build(force=False, verbose=False):
    extra = os.environ.get("WORLD_HIP_EXTRA_FLAGS", "").split()
    if extra:
        force = True
    os.makedirs(OBJ, exist_ok=True)
    units = [u for u in UNITS if os.path.exists(os.path.join(CSRC, u))]
    hdr = max(_newest_header(), os.path.getmtime(os.path.join(HERE, "..", "include", "world_hip.h")))
    jobs = []
    for u in units:
        src = os.path.join(CSRC, u)
        obj = os.path.join(OBJ, u + ".o")
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr):
            jobs.append([hipcc(), *FLAGS, *extra, "-x", "hip", "-c", src, "-o", obj])

    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if verbose and r.stderr:
            print(r.stderr)

    with ThreadPoolExecutor(max_workers=8) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(OBJ, u + ".o") for u in units]
    if jobs or not os.path.exists(LIB):
        run([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    return LIB


VAR_DIR = os.path.join(HERE, "variants")
# the race-shaking builds of devrt.h (test infrastructure: the GPU suite requires their results to be bit-identical to the
# product library's).  Not the product: nothing loads them unless a test names them.
CHECKED = {"jitter": "-DWH_JITTER", "poison": "-DWH_LDS_POISON"}


def build_variant(name, flags):
    """world_amd/variants/libworld_hip_<name>.so = the library's sources with extra compiler flags (tools/ab.py's A/B builds,
    the checked builds above).  Travels to the GPU box with the tree; git ignores the directory."""
    obj_dir = os.path.join(VAR_DIR, "_obj_" + name)
    os.makedirs(obj_dir, exist_ok=True)
    out = os.path.join(VAR_DIR, f"libworld_hip_{name}.so")
    units = [u for u in UNITS if os.path.exists(os.path.join(CSRC, u))]
    stamp = os.path.join(obj_dir, "flags.txt")
    same_flags = os.path.exists(stamp) and open(stamp).read() == flags
    hdr = _newest_header()
    jobs = []
    for u in units:
        src, obj = os.path.join(CSRC, u), os.path.join(obj_dir, u + ".o")
        if not same_flags or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr):
            jobs.append([hipcc(), *FLAGS, *flags.split(), "-x", "hip", "-c", src, "-o", obj])

    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
    with ThreadPoolExecutor(max_workers=8) as ex:
        list(ex.map(run, jobs))
    if jobs or not os.path.exists(out):
        run([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *[os.path.join(obj_dir, u + ".o") for u in units]])
    open(stamp, "w").write(flags)
    return out


def build_checked():
    return {name: build_variant(name, flags) for name, flags in CHECKED.items()}


def build_examples():
    """examples/batch_analysis.cpp: the device-resident C API from plain C++ (hipcc, host code only);
    examples/dropin_bench.cpp: the drop-in symbols timed from a plain C++ caller (g++: it knows nothing of HIP)."""
    root = os.path.dirname(HERE)
    inc = os.path.join(root, "include")
    out = None
    for name, cmd in (("batch_analysis", [hipcc(), "-O1"]), ("dropin_bench", ["g++", "-O2", "-std=c++17", "-pthread"])):
        src, exe = os.path.join(root, "examples", name + ".cpp"), os.path.join(root, "examples", name)
        if not os.path.exists(src):
            continue
        newest = max(os.path.getmtime(src), os.path.getmtime(LIB), os.path.getmtime(os.path.join(inc, "world_hip.h")))
        if not os.path.exists(exe) or os.path.getmtime(exe) < newest:
            r = subprocess.run([*cmd, "-I", inc, src, "-L", HERE, "-lworld_hip", "-Wl,-rpath,$ORIGIN/../world_amd",
                                "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath-link,/opt/rocm/lib", "-o", exe], capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"building examples/{name} failed:\n" + r.stdout + r.stderr)
        out = out or exe
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    if "--examples" in sys.argv:
        print(build_examples())
    if "--checked" in sys.argv:
        print(build_checked())
