#!/bin/bash
# round 5, call d: K1x v2 (scalar row bases, 8-byte stores, asm row sums): parity + tile vs wave; SALU-fill micro-benchmark; backward suites on the pruned library
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w $R/profiles/scripts/ubench_4x4.hip -o /tmp/ub4 && /tmp/ub4 > $O/r05_ubench_4x4.txt 2>&1
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -6 > $O/r05d_pytest_parity.txt
{
for r in 1 2; do for k in tile wave; do for m in rk4 euler midpoint; do
  python bench.py --workload ode01 --method $m --kernel $k --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('round $r $k $m kernel_ms %.4f frac %.4f' % (d['roofline']['kernel_ms'], d['roofline']['frac']))"
done; done; done
for B in 2048 6144 8192; do for k in tile wave; do
  python bench.py --workload ode01 --method rk4 --kernel $k --batch $B --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=$B $k rk4 kernel_ms %.4f frac %.4f' % (d['roofline']['kernel_ms'], d['roofline']['frac']))"
done; done
} > $O/r05d_tile_vs_wave.txt 2>&1
python -m pytest tests/test_gpu_backward.py tests/test_grad_goldens.py tests/test_tf_goldens.py tests/test_gpu_rows_backward.py tests/test_gpu_dae_encoded.py -m gpu -q 2>&1 | tail -25 > $O/r05d_pytest_bwd.txt
