cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rocprofv3 --kernel-trace --stats -d $O/r02g_ode02 -o t -- python $R/profiles/scripts/train_step_models.py ode02 > /dev/null 2>&1
python $R/profiles/summarize_rocprof.py $O/r02g_ode02/t_results.db > $O/r02g_train_ode02_kernel_stats.txt; rm -rf $O/r02g_ode02
head -34 $O/r02g_train_ode02_kernel_stats.txt | cut -c1-150
