for a in 0 1 2 3; do
  WORLD_HIP_EXTRA_FLAGS="-DD4C_ABLATE=$a" python -m world_amd.build > /dev/null 2>&1
  echo "ABLATE=$a"; timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; print({x:k[x] for x in ['d4c_body','d4c_lovetrain','ct_frame']})"
done
