cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rocprofv3 --kernel-trace --stats -d $O/r02g_dae02 -o t -- python $R/profiles/scripts/train_step_models.py dae02 > /dev/null 2>&1
python $R/profiles/summarize_rocprof.py $O/r02g_dae02/t_results.db > $O/r02g_train_dae02_kernel_stats.txt; rm -rf $O/r02g_dae02
head -40 $O/r02g_train_dae02_kernel_stats.txt | cut -c1-170
