set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_encoded.py tests/test_gpu_bench_dist.py -x -q > $O/r03d_pytest.txt 2>&1; tail -5 $O/r03d_pytest.txt
bash profiles/scripts/pmc_sq.sh r03d_k4f_h128_rk4 ode_backward_fused --train --hidden 128 --steps 2 --warmup 1 > /dev/null
bash profiles/scripts/pmc_sq.sh r03d_k4f_h128_euler ode_backward_fused --train --hidden 128 --method euler --steps 2 --warmup 1 > /dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/r03d_kt -o t -- python $R/bench.py --steps 3 --warmup 1 --train --hidden 128 --no-cpu-baseline > /dev/null 2>&1; python $R/profiles/summarize_rocprof.py $O/r03d_kt/t_results.db > $O/r03d_train_ode01_h128_kernel_stats.txt; rm -rf $O/r03d_kt
cd $R; rm -f $O/pmc_r03d*.log
head -30 $O/r03d_train_ode01_h128_kernel_stats.txt
