set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rocprofv3 --kernel-trace --stats -d $O/r01b_train_trace -o t -- python $R/bench.py --train --steps 5 --warmup 2 --no-cpu-baseline > $O/r01b_train_trace.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/r01b_dae01_trace -o t -- python $R/bench.py --workload dae01 --steps 5 --warmup 2 --no-cpu-baseline > $O/r01b_dae01_trace.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/r01b_ode02_trace -o t -- python $R/bench.py --workload ode02 --steps 5 --warmup 2 --no-cpu-baseline > $O/r01b_ode02_trace.log 2>&1
python $R/bench.py --train --steps 5 --warmup 2 --no-cpu-baseline --train-baseline-steps 50 | grep "^{" > $O/r01b_bench_ode01_train_n1.json
