set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 1800 python -m pytest tests/test_gpu_backward.py tests/test_grad_goldens.py tests/test_gpu_determinism.py -x -q > $O/r03f_pytest.txt 2>&1; tail -5 $O/r03f_pytest.txt
for wl in ode01 dae01; do for m in rk4 euler; do python bench.py --steps 5 --warmup 2 --train --workload $wl --method $m --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train h64 $wl $m ms', d['ms_per_step'])"; done; done
cd /tmp && export TMPDIR=/tmp
for wl in ode01 dae01; do rocprofv3 --kernel-trace --stats -d $O/r03f_kt -o t -- python $R/bench.py --steps 5 --warmup 2 --train --workload $wl --no-cpu-baseline > /dev/null 2>&1; python $R/profiles/summarize_rocprof.py $O/r03f_kt/t_results.db | head -8 | cut -c1-150; rm -rf $O/r03f_kt; done
