#!/bin/bash
# round 5, call a: the 4x4x1 exchange-free tile micro-benchmark + the clamp-ELU A/B against round 4's library
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w $R/profiles/scripts/ubench_4x4.hip -o /tmp/ub4 && /tmp/ub4 > $O/r05_ubench_4x4.txt 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w $R/profiles/scripts/ubench_exchange.hip -o /tmp/ubx && /tmp/ubx > $O/r05_ubench_exchange.txt 2>&1
{
bash $R/profiles/scripts/ab_libs.sh 3 1 "r4 tree" --workload ode01 --method rk4
bash $R/profiles/scripts/ab_libs.sh 2 0 "r4 tree" --workload ode01 --method euler
bash $R/profiles/scripts/ab_libs.sh 2 0 "r4 tree" --workload dae01 --method rk4
bash $R/profiles/scripts/ab_libs.sh 2 0 "r4 tree" --workload dae01 --method euler
bash $R/profiles/scripts/ab_libs.sh 2 0 "r4 tree" --workload ode01 --method rk4 --hidden 128
bash $R/profiles/scripts/ab_libs.sh 2 0 "r4 tree" --workload ode01 --method rk4 --train
bash $R/profiles/scripts/ab_libs.sh 2 0 "r4 tree" --workload dae01 --method rk4 --train
} > $O/r05a_elu_clamp_ab.txt 2>&1
cd $R && python -m pytest tests/test_gpu_parity.py tests/test_grad_goldens.py tests/test_tf_goldens.py -m gpu -x -q 2>&1 | tail -5 > $O/r05a_pytest.txt
