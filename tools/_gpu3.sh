mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_dropin.py -m gpu -q -x 2>&1 | tail -5 > gpurun_out/t3.log
python tools/ab.py run tree base > gpurun_out/ab3.log 2>&1
python - <<'PY' > gpurun_out/dropin3.log 2>&1
import numpy as np, subprocess, os, sys
sys.path.insert(0, os.getcwd())
from world_amd import synth
x = synth.vowel(48000, 10.0, seed=12345).numpy().astype(np.float64)
x.tofile("/tmp/x.f64")
for env in ({}, {"WORLD_HIP_HOST_TRACE": "1"}):
    r = subprocess.run(["examples/dropin_bench", "/tmp/x.f64", "48000", "10", "4"], capture_output=True, text=True, env=dict(os.environ, **env))
    print(env, r.stdout.strip(), "\n".join(l for l in r.stderr.splitlines() if "dropin" in l or "api:" in l or l.startswith("  Harvest") or l.startswith("  CheapTrick") or l.startswith("  D4C") or "total" in l))
PY
cat gpurun_out/t3.log gpurun_out/ab3.log gpurun_out/dropin3.log
