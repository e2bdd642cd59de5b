#!/bin/bash
# round 4: ODE_02 (hidden 16) training step, kernel breakdown
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rocprofv3 --kernel-trace --stats -d $O/r04ah_d -o t -- python $R/profiles/scripts/train_step_models.py ode02 > $O/r04ah_train.log 2>&1
python $R/profiles/summarize_rocprof.py $O/r04ah_d/t_results.db > $O/r04ah_train_ode02_kernel_stats.txt; rm -rf $O/r04ah_d
grep -v amdgpu $O/r04ah_train.log | tail -2
head -26 $O/r04ah_train_ode02_kernel_stats.txt | cut -c1-170
