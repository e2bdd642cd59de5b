#!/bin/bash
# round 5, call c: K1x (one wave per 4 trajectories) first light: parity, then tile (K1) vs wave (K1x) on the headline
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -15 > $O/r05c_pytest_parity.txt
{
for r in 1 2; do for k in tile wave; do for m in rk4 euler midpoint; do
  python bench.py --workload ode01 --method $m --kernel $k --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('round $r $k $m kernel_ms %.4f frac %.4f err %s' % (d['roofline']['kernel_ms'], d['roofline']['frac'], d.get('traj_rel_err', d.get('parity', ''))))"
done; done; done
for B in 8192 16384; do for k in tile wave; do
  python bench.py --workload ode01 --method rk4 --kernel $k --batch $B --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=$B $k rk4 kernel_ms %.4f frac %.4f' % (d['roofline']['kernel_ms'], d['roofline']['frac']))"
done; done
} > $O/r05c_tile_vs_wave.txt 2>&1
python profiles/scripts/accuracy_report.py 2>&1 | grep -v amdgpu.ids > $O/r05c_accuracy.txt
python -m pytest tests/test_gpu_backward.py tests/test_grad_goldens.py tests/test_tf_goldens.py -m gpu -x -q 2>&1 | tail -15 > $O/r05c_pytest_bwd.txt
