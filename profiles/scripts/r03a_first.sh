# Round 3, first GPU pass: bench.py's own launcher, the default line with the extra workloads, SQ counters of the Euler kernels.
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
python bench.py --gpus 1 --force-dist --steps 3 --warmup 1 --no-cpu-baseline > $O/r03a_selflaunch.out 2> $O/r03a_selflaunch.err; echo "rc=$?" >> $O/r03a_selflaunch.out
python bench.py --gpus 1 --steps 20 --warmup 5 2> $O/r03a_bench_default.err | tail -1 > $O/r03a_bench_default.json
timeout 900 python -m pytest tests/test_gpu_bench_dist.py -x -q > $O/r03a_pytest_dist.txt 2>&1
bash profiles/scripts/pmc_sq.sh r03a_ode01_euler integrate_mfma --workload ode01 --method euler > /dev/null
bash profiles/scripts/pmc_sq.sh r03a_dae01_euler integrate_mfma --workload dae01 --method euler > /dev/null
bash profiles/scripts/pmc_sq.sh r03a_ode01_rk4 integrate_mfma --workload ode01 > /dev/null
rm -f $O/pmc_r03a_*.log
ls $O
