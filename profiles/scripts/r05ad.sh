#!/bin/bash
# round 5, call ad: SQ counters of the two-role K3f (config 3 forward) at RK4 and Euler, and of the two-role K8f
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $GRAFT_REPO_ROOT
bash profiles/scripts/pmc_sq.sh r05ad_ode02_rk4 latent_dpp --workload ode02 --warmup 10 > /dev/null 2>&1
bash profiles/scripts/pmc_sq.sh r05ad_ode02_euler latent_dpp --workload ode02 --method euler --warmup 10 > /dev/null 2>&1
rm -f gpurun_out/pmc_r05ad_*.log
