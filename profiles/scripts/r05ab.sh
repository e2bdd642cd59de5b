#!/bin/bash
# round 5, call ab: the whole GPU suite + the fuzzes after the K3f two-role form and K8f's MFMA rank-1 updates
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
python -m pytest tests/ -m gpu -q --tb=short 2>&1 | tail -15 > $O/r05ab_pytest.txt
for seed in 31 32; do
  python profiles/scripts/fuzz_models.py $seed 200 2>&1 | grep -v amdgpu | tail -2 >> $O/r05ab_pytest.txt
  python profiles/scripts/fuzz_forward.py $seed 200 2>&1 | grep -v amdgpu | tail -2 >> $O/r05ab_pytest.txt
done
