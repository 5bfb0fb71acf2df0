#!/bin/bash
# Development aid: examples/dropin_bench (the drop-in path from a plain C++ caller) under the drop-in layer's knobs.
python - <<'PY'
import numpy as np, sys
sys.path.insert(0, '.')
from world_amd import synth
synth.vowel(48000, 10.0, seed=12345, base_f0=140.0).numpy().astype(np.float64).tofile('/tmp/x.f64')
PY
for rep in 1 2; do
for cfg in "" "WORLD_HIP_DROPIN_SPIN_US=0" "WORLD_HIP_DROPIN_SPIN_US=1000" "WORLD_HIP_DROPIN_WIRE=f32" "WORLD_HIP_DROPIN_RANGES=0" "WORLD_HIP_DROPIN_COPY_THREADS=0"; do
  echo "== [$cfg]"; env $cfg examples/dropin_bench /tmp/x.f64 48000 10 4 | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['separate_rows_ms'], d['separate_rows_stages_ms'], 'dense', d['dense_rows_ms'], 'threads', d['threads_ms_per_utterance'])"
done; done
