OUT=gpurun_out/fuzz_sweeps_r06_final.txt
{
echo "The same sweeps on the round's last tree (csrc hash $(python -c 'import bench; print(bench.csrc_hash())'); seeds 301 / 302):"
for cmd in "tests/fuzz_parity.py 301 400" "tests/fuzz_parity.py 302 400" "tests/fuzz_given_f0.py 301 500" "tests/fuzz_batched.py 301 100"; do
  echo "== python $cmd"; python $cmd 2>&1 | grep -v amdgpu.ids | tail -2
done
} > $OUT 2>&1
cat $OUT
