"""Ablation timing on the GPU box (development aid): swap in alternative source files, rebuild, time, restore."""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import json, subprocess, sys, shutil
VARIANTS = {
  'base': {},
  # 'name': {'world_amd/csrc/<unit>': 'path/to/alternative/source'},
  'base2': {},
}
KERNELS = ('d4c_groupdelay', 'd4c_band', 'd4c_lovetrain', 'ct_frame')
def run(name):
    out = subprocess.run([sys.executable, 'bench.py', '--steps', '30', '--warmup', '3', '--streams', '1', '--no-cpu-baseline', '--no-extras', '--no-configs', '--min-wall', '0'],,
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
