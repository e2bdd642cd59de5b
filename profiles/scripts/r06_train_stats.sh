# Round 6: rocprofv3 kernel-trace stats of the ODE_01 / DAE_01 training steps (K1x SAVE + K6 + K4x / K7f).
#   gpurun -- 'bash profiles/scripts/r06_train_stats.sh [tag] [workloads...]'   then copy gpurun_out/<tag>_* into profiles/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-r06}
B="python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras"
kt() { timeout 300 rocprofv3 --kernel-trace --stats -d $O/${TAG}_$1 -o t -- "${@:2}" > $O/${TAG}_$1.log 2>&1; timeout 60 python $R/profiles/summarize_rocprof.py $O/${TAG}_$1/t_results.db > $O/${TAG}_$1_kernel_stats.txt; rm -rf $O/${TAG}_$1 $O/${TAG}_$1.log; }
kt train_ode01 $B --train
kt train_ode01_euler $B --train --method euler
if [ "$2" = "dae" ]; then
kt train_dae01 $B --train --workload dae01
kt train_dae01_euler $B --train --workload dae01 --method euler
fi
head -12 $O/${TAG}_train_ode01_kernel_stats.txt $O/${TAG}_train_ode01_euler_kernel_stats.txt < /dev/null
