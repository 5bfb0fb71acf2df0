mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x 2>&1 | tail -6 > gpurun_out/t5.log
python tools/ab.py run tree base > gpurun_out/ab5.log 2>&1
python - <<'PY' > gpurun_out/dropin5.log 2>&1
import numpy as np, subprocess, os, sys
sys.path.insert(0, os.getcwd())
from world_amd import synth
x = synth.vowel(48000, 10.0, seed=12345).numpy().astype(np.float64)
x.tofile("/tmp/x.f64")
for env in ({}, {}):
    r = subprocess.run(["examples/dropin_bench", "/tmp/x.f64", "48000", "20", "4"], capture_output=True, text=True, env=dict(os.environ, **env))
    print(env, r.stdout.strip(), r.stderr[-300:])
PY
cat gpurun_out/t5.log gpurun_out/ab5.log gpurun_out/dropin5.log
