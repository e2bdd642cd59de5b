#!/bin/bash
# round 5, call b: the log2e-scaled ELU domain of the inference forwards K1 / K2 (tree) against the clamp-only build (var_clamp)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w $R/profiles/scripts/ubench_4x4.hip -o /tmp/ub4 && /tmp/ub4 > $O/r05_ubench_4x4.txt 2>&1
{
bash $R/profiles/scripts/ab_libs.sh 3 1 "clamp tree" --workload ode01 --method rk4
bash $R/profiles/scripts/ab_libs.sh 2 0 "clamp tree" --workload ode01 --method euler
bash $R/profiles/scripts/ab_libs.sh 2 0 "clamp tree" --workload dae01 --method rk4
bash $R/profiles/scripts/ab_libs.sh 2 0 "clamp tree" --workload dae01 --method euler
} > $O/r05b_elu_scaled_ab.txt 2>&1
cd $R && python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_determinism.py -m gpu -x -q 2>&1 | tail -5 > $O/r05b_pytest.txt
