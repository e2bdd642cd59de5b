#!/bin/bash
# round 5, call ac: K1x as the saving forward with two steps of look-ahead at Euler / Midpoint (two steps of saved-row stores in flight)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R; python -m pytest tests/test_gpu_backward.py -m gpu -q --tb=short -k "saves_the_same_rows or k4f or wide" 2>&1 | tail -4 > $O/r05ac_pytest.txt; cd /tmp
for m in euler midpoint; do
rocprofv3 --kernel-trace --stats -d $O/r05ac_$m -o t -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extras --train --method $m > /dev/null 2>&1
python $R/profiles/summarize_rocprof.py $O/r05ac_$m/t_results.db | head -5 > $O/r05ac_train_ode01_${m}_kernel_stats.txt; rm -rf $O/r05ac_$m
done
