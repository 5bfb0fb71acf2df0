for a in 4 2; do
  WORLD_HIP_EXTRA_FLAGS="-DD4C_MIN_WAVES=$a" python -m world_amd.build > /dev/null 2>&1
  echo "D4C_MIN_WAVES=$a"; timeout 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; print(d['ms_per_step'], {x:k[x] for x in ['d4c_body','d4c_lovetrain','ct_frame','hv_band_events','hv_refine']})"
done
