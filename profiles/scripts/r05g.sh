#!/bin/bash
# round 5, call g: SQ counters of K1x (Euler / RK4) and K1 Euler; the rest of the GPU suite
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
export GRAFT_REPO_ROOT=$R
bash profiles/scripts/pmc_sq.sh r05g_k1x_euler integrate_x --workload ode01 --method euler --kernel wave > /dev/null 2>&1
bash profiles/scripts/pmc_sq.sh r05g_k1x_rk4 integrate_x --workload ode01 --method rk4 --kernel wave > /dev/null 2>&1
bash profiles/scripts/pmc_sq.sh r05g_k1_euler integrate_mfma --workload ode01 --method euler --kernel tile > /dev/null 2>&1
cd $R
python -m pytest tests/ -m gpu -q --tb=line -k "not parity" 2>&1 | tail -15 > $O/r05g_pytest_rest.txt
python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=line -k "order_of_accuracy" 2>&1 | tail -3 > $O/r05g_pytest_order.txt
