"""Development aid: one short bench run; prints throughput, latency and the per-kernel times whose names start with
one of the given prefixes.
    python tools/quick_bench.py <label> [prefix,prefix,...] [config]
config: 1 (default: the headline leg, configs[1], lone job's kernels) or 2 / 3 / 4 (that leg alone, one batched step)"""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pre = tuple((sys.argv[2] if len(sys.argv) > 2 else 'd4c,ct_').split(','))
cfg = int(sys.argv[3]) if len(sys.argv) > 3 else 1
cmd = [sys.executable, os.path.join(root, 'bench.py'), '--no-cpu-baseline', '--min-wall', '0.5']
cmd += ['--only-config', str(cfg)] if cfg != 1 else ['--no-configs', '--no-extras']
out = subprocess.run(cmd, capture_output=True, text=True).stdout.strip().splitlines()[-1]
d = json.loads(out); k = d['kernels_ms_per_step']
lat = d['single_job_latency_ms'] if cfg == 1 else d['single_step_latency_ms']
print(sys.argv[1], 'value %.0f lat %.3f' % (d['value'], lat), {n: round(k[n], 4) for n in k if n.startswith(pre)}, flush=True)
