set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 1500 python -m pytest tests/test_gpu_backward.py tests/test_grad_goldens.py tests/test_gpu_determinism.py -q -x > $O/r03p_pytest.txt 2>&1; tail -4 $O/r03p_pytest.txt
( for r in 1 2; do for sv in 0 auto; do for m in rk4 midpoint euler; do
  PSNODE_SAVE_ACTIVATIONS=$sv python bench.py --steps 4 --warmup 2 --train --hidden 128 --method $m --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('round $r save=$sv h128 $m train ms', round(d['ms_per_step'],3))"
done; done; done
for sv in 0 1; do for h in 64 32; do
  PSNODE_SAVE_ACTIVATIONS=$sv python bench.py --steps 4 --warmup 2 --train --hidden $h --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('save=$sv h$h rk4 train ms', round(d['ms_per_step'],3))"
done; done ) 2>/dev/null | grep "train ms" > $O/r03p_saved_ab.txt
cat $O/r03p_saved_ab.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/r03p_kt -o t -- python $R/bench.py --steps 3 --warmup 1 --train --hidden 128 --no-cpu-baseline > /dev/null 2>&1; python $R/profiles/summarize_rocprof.py $O/r03p_kt/t_results.db | head -8 | cut -c1-150; rm -rf $O/r03p_kt
