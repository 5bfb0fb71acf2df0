"""A/B timing of library variants (development aid).  Build here (no GPU needed):

    python tools/ab.py build name1="-DFLAG1 -DFLAG2" name2="" ...

writes world_amd/variants/libworld_hip_<name>.so (they travel to the GPU box with the tree; git ignores them).  On the
GPU box, inside ONE gpurun call so that the variants share a box:

    python tools/ab.py run [name ...]

runs, per variant, the headline leg of bench.py (12 jobs in flight, per-kernel HIP-event profile) and the configs[3]
share (128 x 5 s batch), checks D4C / Harvest against the reference on the way (bench.py's own parity_in_run), and prints
one line per variant.  Variants are interleaved twice (A B A B) so that drift shows.
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VAR_DIR = os.path.join(ROOT, "world_amd", "variants")


def build(name, flags):
    from world_amd import build as B
    print("built", B.build_variant(name, flags), "(" + (flags or "no extra flags") + ")")


def bench(name, extra):
    lib = os.path.join(ROOT, "world_amd", "libworld_hip.so") if name == "tree" else os.path.join(VAR_DIR, f"libworld_hip_{name}.so")
    env = dict(os.environ, WORLD_HIP_LIB=lib)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-extras", *extra], capture_output=True, text=True,
                       env=env, timeout=900)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"error": (r.stderr or r.stdout)[-600:]}
    return json.loads(lines[-1])


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        for spec in sys.argv[2:]:
            name, _, flags = spec.partition("=")
            build(name, flags)
        return
    names = sys.argv[2:] or ["tree"] + sorted(f[len("libworld_hip_"):-3] for f in os.listdir(VAR_DIR) if f.endswith(".so"))
    show = ("d4c_frame", "hv_refine", "hv_band_events_fft", "ct_frame", "hv_raw_candidates", "hv_detect", "d4c_lovetrain",
            "ct_envelope", "ct_spectrum", "ct_scan")        # (the last three: variants built from trees before round 4's second half)
    for rnd in range(2):
        for name in names:
            d = bench(name, ["--no-configs", "--no-cpu-baseline", "--min-wall=1.5"])
            if "error" in d:
                print(name, "ERROR", d["error"], flush=True)
                continue
            k = d.get("kernels_ms_per_step", {})
            par = d.get("parity_in_run", {})
            print(f"[{rnd}] {name:14s} value {d['value'] / 1e6:.3f} M  single {d.get('value_single_job', 0) / 1e6:.3f} M  "
                  + " ".join(f"{n}={k.get(n, 0):.4f}" for n in show)
                  + f"  parity ok={par.get('ok')} ap={par.get('ap')} f0={par.get('f0')} sp={par.get('sp')}", flush=True)
    for name in names:                              # parity of every variant against the reference (0.5 s: __graft_entry__.smoke)
        lib = os.path.join(ROOT, "world_amd", "libworld_hip.so") if name == "tree" else os.path.join(VAR_DIR, f"libworld_hip_{name}.so")
        r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"], cwd=ROOT, capture_output=True, text=True,
                           env=dict(os.environ, WORLD_HIP_LIB=lib), timeout=600)
        print(f"[smoke] {name:14s}", (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1], flush=True)
    for name in names:
        d = bench(name, ["--only-config", "3", "--min-wall=1.5", "--no-cpu-baseline"])
        if "error" in d:
            print(name, "configs[3] ERROR", d["error"], flush=True)
            continue
        k = d.get("kernels_ms_per_step", {})
        print(f"[c3] {name:14s} value {d.get('value', 0) / 1e6:.3f} M  ms/step {d.get('ms_per_step', 0):.2f}  "
              + " ".join(f"{n}={k.get(n, 0):.3f}" for n in show), flush=True)


if __name__ == "__main__":
    main()
