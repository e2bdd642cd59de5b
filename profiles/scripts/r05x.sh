#!/bin/bash
# round 5, call x: the shape fuzzes on the round's new kernels (K1x / K2x forward, K1x as the saving forward, one autograd node per module)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
: > $O/r05x_fuzz_summary.txt
for seed in 11 12 13; do
  echo "== fuzz_forward seed $seed x 300" >> $O/r05x_fuzz_summary.txt;  python profiles/scripts/fuzz_forward.py $seed 300 2>&1 | grep -v amdgpu | tail -4 >> $O/r05x_fuzz_summary.txt
  echo "== fuzz_backward seed $seed x 300" >> $O/r05x_fuzz_summary.txt; python profiles/scripts/fuzz_backward.py $seed 300 2>&1 | grep -v amdgpu | tail -4 >> $O/r05x_fuzz_summary.txt
  echo "== fuzz_models seed $seed x 150" >> $O/r05x_fuzz_summary.txt;   python profiles/scripts/fuzz_models.py $seed 150 2>&1 | grep -v amdgpu | tail -4 >> $O/r05x_fuzz_summary.txt
  echo "== fuzz_dae_encoded seed $seed x 60" >> $O/r05x_fuzz_summary.txt; python profiles/scripts/fuzz_dae_encoded.py $seed 60 2>&1 | grep -v amdgpu | tail -3 >> $O/r05x_fuzz_summary.txt
done
