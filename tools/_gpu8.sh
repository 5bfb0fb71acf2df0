mkdir -p gpurun_out
for s in 8 10 12 14 16 20; do
  python bench.py --no-configs --no-extras --no-cpu-baseline --min-wall 1.5 --streams $s 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('streams', $s, 'value %.3f M' % (d['value'] / 1e6), 'ms/step %.4f' % d['ms_per_step'])"
done
for q in 4 8 24; do
  GPU_MAX_HW_QUEUES=$q python bench.py --no-configs --no-extras --no-cpu-baseline --min-wall 1.5 --streams 12 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('queues', $q, 'streams 12 value %.3f M' % (d['value'] / 1e6))"
done
