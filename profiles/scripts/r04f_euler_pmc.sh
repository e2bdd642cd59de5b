#!/bin/bash
# round 4: SQ breakdown of the Euler training step's backward kernels at hidden 64 (none existed: VERDICT round 3 item 3)
R=$GRAFT_REPO_ROOT; cd $R
bash profiles/scripts/pmc_sq.sh r04_k4f_saved_euler ode_backward_fused --train --method euler --steps 2 --warmup 1 > /dev/null
PSNODE_SAVE_ACTIVATIONS=0 bash profiles/scripts/pmc_sq.sh r04_k4_euler ode_backward_kernel --train --method euler --steps 2 --warmup 1 > /dev/null
bash profiles/scripts/pmc_sq.sh r04_k7_euler dae_backward_kernel --train --workload dae01 --method euler --steps 2 --warmup 1 > /dev/null
PSNODE_SAVE_ACTIVATIONS=1 bash profiles/scripts/pmc_sq.sh r04_k7f_saved_euler dae_backward_fused --train --workload dae01 --method euler --steps 2 --warmup 1 > /dev/null
bash profiles/scripts/pmc_sq.sh r04_k1_saving_euler integrate_mfma --train --method euler --steps 2 --warmup 1 > /dev/null
rm -f gpurun_out/pmc_r04_*.log
for f in gpurun_out/r04_*_pmc_sq.txt; do echo "== $f"; cat $f; done
# ... and the RK4 instances of the same kernels at hidden 64 (the 4-wave laggards of VERDICT round 3 item 5)
bash profiles/scripts/pmc_sq.sh r04_k4f_saved_rk4 ode_backward_fused --train --steps 2 --warmup 1 > /dev/null
bash profiles/scripts/pmc_sq.sh r04_k7f_saved_rk4 dae_backward_fused --train --workload dae01 --steps 2 --warmup 1 > /dev/null
bash profiles/scripts/pmc_sq.sh r04_k7h_rk4 head_grads --train --workload dae01 --steps 2 --warmup 1 > /dev/null
rm -f gpurun_out/pmc_r04_*.log
