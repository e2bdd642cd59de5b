cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
kt() { rocprofv3 --kernel-trace --stats -d $O/r03x_$1 -o t -- "${@:2}" > $O/r03x_$1.log 2>&1; python $R/profiles/summarize_rocprof.py $O/r03x_$1/t_results.db > $O/r03x_$1_kernel_stats.txt; rm -rf $O/r03x_$1 $O/r03x_$1.log; }
kt train_dae01_h128 $B --train --workload dae01 --hidden 128
head -16 $O/r03x_train_dae01_h128_kernel_stats.txt | cut -c1-150
