#!/bin/bash
# kernel-trace of the hidden-64 training steps after the round-4 changes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
B="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --train"
kt() { rocprofv3 --kernel-trace --stats -d $O/r04_$1 -o t -- "${@:2}" > $O/r04_$1.log 2>&1; python $R/profiles/summarize_rocprof.py $O/r04_$1/t_results.db > $O/r04_$1_kernel_stats.txt; rm -rf $O/r04_$1 $O/r04_$1.log; }
kt train_dae01 $B --workload dae01
kt train_dae01_euler $B --workload dae01 --method euler
kt train_ode01 $B
kt train_ode01_euler $B --method euler
head -25 $O/r04_train_dae01_kernel_stats.txt; head -25 $O/r04_train_dae01_euler_kernel_stats.txt
