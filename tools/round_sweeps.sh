#!/bin/bash
# The randomised sweeps against the reference and the long utterance, on the tree as it stands (one gpurun call):
#   tools/round_sweeps.sh <round> <seed>   -> profiles/<round>/fuzz_sweeps.txt (copied to gpurun_out/ for the merge back)
RND=${1:-r06}; SEED=${2:-52}
OUT=profiles/$RND/fuzz_sweeps.txt; mkdir -p profiles/$RND
{
echo "Randomised sweeps against the reference on the round's final kernels (world_amd/csrc hash $(python -c 'import bench; print(bench.csrc_hash())'); one gpurun call, MI355X):"
for cmd in "tests/fuzz_parity.py $SEED 300" "tests/fuzz_parity.py $((SEED+1)) 300" "tests/fuzz_given_f0.py $SEED 400" "tests/fuzz_batched.py $SEED 80"; do
  echo "== python $cmd"; python $cmd 2>&1 | grep -v amdgpu.ids | tail -2
done
echo; echo "One long utterance (python tests/long_utterance_check.py 240):"
python tests/long_utterance_check.py 240 2>&1 | grep -v amdgpu.ids | tail -3
} > $OUT 2>&1
mkdir -p gpurun_out; cp $OUT gpurun_out/fuzz_sweeps_$RND.txt; cat $OUT
