#!/bin/bash
# Development aid: the headline leg at several numbers of jobs in flight / hardware queues, one gpurun call.
mkdir -p gpurun_out/streams
for s in 8 10 12 14 16 20; do
  python bench.py --streams $s --no-configs --no-extras --no-cpu-baseline --min-wall 1.5 > gpurun_out/streams/s$s.json 2>/dev/null
  python -c "
import json; d=json.loads(open('gpurun_out/streams/s$s.json').read().strip().splitlines()[-1]); print('streams $s', round(d['value']/1e6,3), 'M frames/s')"
done
for q in 8 24; do
  GPU_MAX_HW_QUEUES=$q python bench.py --streams 12 --no-configs --no-extras --no-cpu-baseline --min-wall 1.5 > gpurun_out/streams/q$q.json 2>/dev/null
  python -c "
import json; d=json.loads(open('gpurun_out/streams/q$q.json').read().strip().splitlines()[-1]); print('12 streams, $q hardware queues', round(d['value']/1e6,3), 'M frames/s')"
done
