"""Development aid: one short headline run (configs[1], no extra legs); prints throughput, lone-job latency and the
lone job's per-kernel times whose names start with one of the given prefixes.
    python tools/quick_bench.py <label> [prefix,prefix,...]        (default prefixes: d4c,ct_)"""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--no-configs', '--no-extras', '--no-cpu-baseline', '--min-wall', '0.5'],
                     capture_output=True, text=True).stdout.strip().splitlines()[-1]
d = json.loads(out); k = d['kernels_ms_per_step']
pre = tuple((sys.argv[2] if len(sys.argv) > 2 else 'd4c,ct_').split(','))
print(sys.argv[1], 'value %.0f lat %.3f' % (d['value'], d['single_job_latency_ms']), {n: k[n] for n in k if n.startswith(pre)},
      d['parity_in_run']['slots_bit_identical_to_serial_run'], flush=True)
