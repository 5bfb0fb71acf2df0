for g in 128 64 32; do
  WORLD_HIP_EVENT_GROUP=$g python bench.py --only-config 3 --min-wall 1.5 --no-cpu-baseline > gpurun_out/r5i/c3_g$g.json 2>/dev/null
  python -c "
import json; d=json.loads(open('gpurun_out/r5i/c3_g$g.json').read().strip().splitlines()[-1]); k=d.get('kernels_ms_per_step',{}); print('group $g', round(d['value']/1e6,3), 'M  ms/step', round(d['ms_per_step'],2), 'workspace GB', round(d.get('workspace_bytes',0)/1e9,2), {n:k[n] for n in ('hv_band_events_fft','hv_raw_candidates','d4c_frame') if n in k})"
done
for g in 256 64; do
  WORLD_HIP_EVENT_GROUP=$g python bench.py --only-config 2 --min-wall 1.5 --no-cpu-baseline > gpurun_out/r5i/c2_g$g.json 2>/dev/null
  python -c "
import json; d=json.loads(open('gpurun_out/r5i/c2_g$g.json').read().strip().splitlines()[-1]); print('config2 group $g', round(d['value']/1e6,3), 'M  ms/step', round(d['ms_per_step'],2), 'workspace GB', round(d.get('workspace_bytes',0)/1e9,2))"
done
