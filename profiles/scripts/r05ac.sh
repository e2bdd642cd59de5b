#!/bin/bash
# round 5, call ac: K1x as the saving forward with a ring of R steps of look-ahead (R steps of saved-row stores in flight): tests, kernel times
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R; python -m pytest tests/test_gpu_backward.py tests/test_grad_goldens.py tests/test_gpu_determinism.py -m gpu -q --tb=short 2>&1 | tail -4 > $O/r05ac_pytest.txt; cd /tmp
for m in rk4 euler midpoint; do
rocprofv3 --kernel-trace --stats -d $O/r05ac_$m -o t -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extras --train --method $m > /dev/null 2>&1
python $R/profiles/summarize_rocprof.py $O/r05ac_$m/t_results.db | head -5 > $O/r05ac_train_ode01_${m}_kernel_stats.txt; rm -rf $O/r05ac_$m
done
