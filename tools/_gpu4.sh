mkdir -p gpurun_out
python tools/ab.py run tree base > gpurun_out/ab4.log 2>&1
python - <<'PY' > gpurun_out/dropin4.log 2>&1
import numpy as np, subprocess, os, sys
sys.path.insert(0, os.getcwd())
from world_amd import synth
x = synth.vowel(48000, 10.0, seed=12345).numpy().astype(np.float64)
x.tofile("/tmp/x.f64")
for env in ({}, {}, {"WORLD_HIP_DROPIN_COPY_THREADS": "1"}, {"WORLD_HIP_DROPIN_COPY_THREADS": "5"}):
    r = subprocess.run(["examples/dropin_bench", "/tmp/x.f64", "48000", "20", "4"], capture_output=True, text=True, env=dict(os.environ, **env))
    print(env, r.stdout.strip(), r.stderr[-300:])
PY
cp world_amd/libworld_hip.so /tmp/keep.so
python tools/trace.py 10 > gpurun_out/trace4.log 2>&1
cp /tmp/keep.so world_amd/libworld_hip.so
cat gpurun_out/ab4.log gpurun_out/dropin4.log gpurun_out/trace4.log
