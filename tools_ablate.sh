# timing-only ablations (results are wrong on purpose); restores sources afterwards
D=world_amd/csrc
mkdir -p /tmp/orig; cp $D/*.hip $D/*.h /tmp/orig/
restore() { cp /tmp/orig/* $D/; }
run() { python -m world_amd.build > /tmp/build.log 2>&1 || tail -5 /tmp/build.log; timeout 200 python bench.py --steps 10 --warmup 2 --streams 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; print('$1', ' '.join('%s=%.3f'%(x,k[x]) for x in ['hv_refine','d4c_groupdelay','d4c_band','ct_frame']))"; restore; }
run base
# refine: no DFT loop
sed -i 's/          for (int i = g; i < blen; i += G) {/          for (int i = g; i < 0; i += G) {/' $D/harvest.hip; run refine_no_dft
# refine: no window / products (treat every window as cached)
sed -i 's/      const bool same_window = hw == c_hw \&\& first == c_first;/      const bool same_window = true;/' $D/harvest.hip; run refine_no_window
# d4c_band: no select
sed -i 's/  block_smallest_sum(key, filled, H + 1, H - bnd, hist, scratch, \&part, \&tot);/  part = 1.0; tot = 2.0;/' $D/d4c.hip; run band_no_select
# all FFT butterflies off (dif/dit stages do nothing): FFT share of every kernel
sed -i 's/  for (int b = threadIdx.x; b < nbf; b += blockDim.x) {/  for (int b = threadIdx.x; b < 0; b += blockDim.x) {/' $D/fft.h; run no_fft_stages
# ct_frame: no serial scan
sed -i 's/    for (int i = 1; i < seg_len; ++i) { acc = seg\[i\] + acc; seg\[i\] = acc; }/    (void)acc;/' $D/cheaptrick.hip; run ct_no_serial_scan
# groupdelay: no smoothing scans
sed -i 's/  block_scan_incl_double(seg, seg_len, scratch);/  __syncthreads();/' $D/d4c.hip; run gd_no_scan
# groupdelay: only one centroid
sed -i 's/  for (int c = 0; c < 2; ++c) {/  for (int c = 0; c < 1; ++c) {/' $D/d4c.hip; run gd_one_centroid
python -m world_amd.build > /dev/null 2>&1
