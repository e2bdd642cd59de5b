#!/bin/bash
# round 5, call f: K1x with the FAST loop (no event / teacher-forcing / clamp code, peeled last step): parity, tile vs wave, the full suites
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short 2>&1 | tail -25 > $O/r05f_pytest_parity.txt
{
for r in 1 2; do for k in tile wave; do for m in rk4 euler midpoint; do
  python bench.py --workload ode01 --method $m --kernel $k --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('round $r $k $m kernel_ms %.4f frac %.4f' % (d['roofline']['kernel_ms'], d['roofline']['frac']))"
done; done; done
for B in 6144 8192 12288; do for k in tile wave; do for m in rk4 euler; do
  python bench.py --workload ode01 --method $m --kernel $k --batch $B --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=$B $k $m kernel_ms %.4f frac %.4f' % (d['roofline']['kernel_ms'], d['roofline']['frac']))"
done; done; done
} > $O/r05f_tile_vs_wave.txt 2>&1
python profiles/scripts/grad_accuracy_report.py > $O/r05f_grad_accuracy.txt 2>&1
python -m pytest tests/ -m gpu -q --tb=line -x -k "not parity" 2>&1 | tail -15 > $O/r05f_pytest_rest.txt
