cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rocprofv3 --kernel-trace --stats -d $O/r02f_train_h128 -o t -- python $R/bench.py --train --workload dae01 --hidden 128 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python $R/profiles/summarize_rocprof.py $O/r02f_train_h128/t_results.db > $O/r02f_train_dae01_h128_kernel_stats.txt; rm -rf $O/r02f_train_h128
head -24 $O/r02f_train_dae01_h128_kernel_stats.txt | cut -c1-200
