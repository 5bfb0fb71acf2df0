import json,subprocess,sys
out=subprocess.run([sys.executable,'bench.py','--no-configs','--no-extras','--no-cpu-baseline','--min-wall','0.5'],capture_output=True,text=True).stdout.strip().splitlines()[-1]
d=json.loads(out); k=d['kernels_ms_per_step']
print(sys.argv[1], 'value %.0f lat %.3f'%(d['value'],d['single_job_latency_ms']), {n:k[n] for n in k if n.startswith(('d4c','ct_'))}, d['parity_in_run']['slots_bit_identical_to_serial_run'], flush=True)
