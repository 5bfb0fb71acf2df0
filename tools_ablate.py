"""Ablation timing on the GPU box (development aid): patch a kernel source, rebuild, time, restore."""
import json, subprocess, sys
VARIANTS = {
  'base': ('world_amd/csrc/cheaptrick.hip', []),
  'ct_lb3': ('world_amd/csrc/cheaptrick.hip', [("__global__ void __launch_bounds__(256, 4) ct_frame(CtParams p) {", "__global__ void __launch_bounds__(256) ct_frame(CtParams p) {")]),
  'base2': ('world_amd/csrc/cheaptrick.hip', []),
  'ct_noprio': ('world_amd/csrc/cheaptrick.hip', [("    __builtin_amdgcn_s_setprio(3);\n", "")]),
}
KERNELS = ('ct_frame', 'd4c_groupdelay', 'd4c_band', 'hv_refine')
def run(name):
    out = subprocess.run([sys.executable, 'bench.py', '--steps', '30', '--warmup', '3', '--streams', '1', '--no-cpu-baseline'],
                         capture_output=True, text=True, timeout=300).stdout.strip().splitlines()[-1]
    d = json.loads(out)
    k = d['kernels_ms_per_step']
    print(name, 'ms/step %.3f' % d['ms_per_step'], {n: k[n] for n in KERNELS}, flush=True)
for name in (sys.argv[1:] or VARIANTS):
    src, edits = VARIANTS[name]
    orig = open(src).read()
    try:
        s = orig
        for a, b in edits:
            assert a in s, (name, a)
            s = s.replace(a, b, 1)
        open(src, 'w').write(s)
        r = subprocess.run([sys.executable, '-m', 'world_amd.build'], capture_output=True, text=True)
        if r.returncode: print(name, 'BUILD FAILED', r.stderr[-800:]); continue
        run(name)
    finally:
        open(src, 'w').write(orig)
subprocess.run([sys.executable, '-m', 'world_amd.build'], capture_output=True, text=True)
