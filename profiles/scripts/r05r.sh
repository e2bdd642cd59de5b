#!/bin/bash
# round 5, call r: kernel times of the ODE_01 training step with K1x as the saving forward
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=r05r
B="python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extras --train"
kt() { rocprofv3 --kernel-trace --stats -d $O/${TAG}_$1 -o t -- "${@:2}" > $O/${TAG}_$1.log 2>&1; python $R/profiles/summarize_rocprof.py $O/${TAG}_$1/t_results.db | head -8 > $O/${TAG}_$1_kernel_stats.txt; rm -rf $O/${TAG}_$1 $O/${TAG}_$1.log; }
kt train_ode01 $B
kt train_ode01_euler $B --method euler
