#!/bin/bash
# round 5, call q: K1x as the training forward (SAVE: the rows K4f reads): parity against K1's rows, the training tests, step times tile vs auto
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
python -m pytest tests/test_gpu_backward.py tests/test_grad_goldens.py tests/test_gpu_determinism.py tests/test_gpu_fuzz.py tests/test_tf_goldens.py -m gpu -q --tb=short 2>&1 | tail -15 > $O/r05q_pytest.txt
{
for r in 1 2; do for k in tile auto; do for m in rk4 euler midpoint; do
  python bench.py --workload ode01 --method $m --kernel $k --train --steps 8 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('round $r $k $m TRAIN ms_per_step %.4f frac %.4f saved %s' % (d['ms_per_step'], d['roofline']['frac'], d.get('saved_bytes')))"
done; done; done
} > $O/r05q_train_tile_vs_wave.txt 2>&1
