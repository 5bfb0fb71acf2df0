# timing-only ablations of d4c_body (results are wrong on purpose); restores the file afterwards
F=world_amd/csrc/d4c.hip
cp $F /tmp/d4c.orig
run() { python -m world_amd.build > /tmp/build.log 2>&1 || tail -5 /tmp/build.log; timeout 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; print('$1', k['d4c_body'])"; }
run base
sed -i 's/    block_smallest_sum(B, H + 1, H - bnd, hist, scratch, &part, &tot);/    part = 1.0; tot = 2.0;/' $F; run no_select; cp /tmp/d4c.orig $F
sed -i 's/  for (int band = 0; band < p.nap; ++band) {/  for (int band = 0; band < 0; ++band) {/' $F; run no_bands; cp /tmp/d4c.orig $F
sed -i 's/  for (int c = 0; c < 2; ++c) {/  for (int c = 0; c < 0; ++c) {/' $F; run no_centroid; cp /tmp/d4c.orig $F
sed -i 's/  for (int c = 0; c < 2; ++c) {/  for (int c = 0; c < 0; ++c) {/; s/  for (int band = 0; band < p.nap; ++band) {/  for (int band = 0; band < 0; ++band) {/' $F; run no_centroid_no_bands; cp /tmp/d4c.orig $F
python -m world_amd.build > /dev/null 2>&1
