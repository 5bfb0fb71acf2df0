OUT=gpurun_out/fuzz_sweeps_r06_more.txt
{
echo "More randomised sweeps against the reference on the final kernels (csrc hash $(python -c 'import bench; print(bench.csrc_hash())'); other seeds, twice the cases):"
for cmd in "tests/fuzz_parity.py 172 600" "tests/fuzz_parity.py 173 600" "tests/fuzz_given_f0.py 172 800" "tests/fuzz_batched.py 172 160"; do
  echo "== python $cmd"; python $cmd 2>&1 | grep -v amdgpu.ids | tail -2
done
} > $OUT 2>&1
cat $OUT
