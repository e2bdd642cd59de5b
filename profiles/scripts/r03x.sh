cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
kt() { rocprofv3 --kernel-trace --stats -d $O/r03x_$1 -o t -- "${@:2}" > $O/r03x_$1.log 2>&1; python $R/profiles/summarize_rocprof.py $O/r03x_$1/t_results.db > $O/r03x_$1_kernel_stats.txt; rm -rf $O/r03x_$1 $O/r03x_$1.log; }
kt train_dae02 python $R/profiles/scripts/train_step_models.py dae02
head -45 $O/r03x_train_dae02_kernel_stats.txt | cut -c1-200
