"""Ablation timing on the GPU box (development aid): swap in alternative source files, rebuild, time, restore."""
import json, subprocess, sys, shutil
VARIANTS = {
  'base': {},
  'old_iir': {'world_amd/csrc/decimate.h': 'tools_old_dec.txt', 'world_amd/csrc/harvest_contour.hip': 'tools_old_hc.txt'},
  'base2': {},
}
KERNELS = ('hv_decimate_fwd', 'hv_decimate_bwd', 'hc_smooth', 'hc_extend', 'hc_merge')
def run(name):
    out = subprocess.run([sys.executable, 'bench.py', '--steps', '30', '--warmup', '3', '--streams', '1', '--no-cpu-baseline'],
                         capture_output=True, text=True, timeout=300).stdout.strip().splitlines()[-1]
    d = json.loads(out)
    k = d['kernels_ms_per_step']
    print(name, 'ms/step %.3f' % d['ms_per_step'], {n: k[n] for n in KERNELS}, flush=True)
for name in (sys.argv[1:] or VARIANTS):
    saved = {dst: open(dst).read() for dst in VARIANTS[name]}
    try:
        for dst, src in VARIANTS[name].items():
            shutil.copy(src, dst)
        r = subprocess.run([sys.executable, '-m', 'world_amd.build', '--force'], capture_output=True, text=True)
        if r.returncode: print(name, 'BUILD FAILED', r.stderr[-800:]); continue
        run(name)
    finally:
        for dst, text in saved.items():
            open(dst, 'w').write(text)
subprocess.run([sys.executable, '-m', 'world_amd.build', '--force'], capture_output=True, text=True)
