F=world_amd/csrc/bandfilter.h
cp $F /tmp/bf.orig
run() { python -m world_amd.build > /tmp/build.log 2>&1 || tail -5 /tmp/build.log; timeout 200 python bench.py --steps 10 --warmup 2 --streams 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; print('$1', k['hv_band_events'])"; }
run base
# no events
sed -i 's/    for (int sub = 0; sub < kTile; sub += nt \* kOutPer) {/    for (int sub = 0; sub < 0; sub += nt * kOutPer) {/' $F; run no_events; cp /tmp/bf.orig $F
# no FIR main loop
sed -i 's/    for (int j0 = 0; j0 < ntap_main; j0 += kOutPer, blk -= kOutPer + 1) {/    for (int j0 = 0; j0 < 0; j0 += kOutPer, blk -= kOutPer + 1) {/' $F; run no_fir; cp /tmp/bf.orig $F
# neither
sed -i 's/    for (int sub = 0; sub < kTile; sub += nt \* kOutPer) {/    for (int sub = 0; sub < 0; sub += nt * kOutPer) {/; s/    for (int j0 = 0; j0 < ntap_main; j0 += kOutPer, blk -= kOutPer + 1) {/    for (int j0 = 0; j0 < 0; j0 += kOutPer, blk -= kOutPer + 1) {/' $F; run neither; cp /tmp/bf.orig $F
python -m world_amd.build > /dev/null 2>&1
