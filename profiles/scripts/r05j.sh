#!/bin/bash
# round 5, call j: runtime teacher-forcing flag in K1 / K2 (A/B against call i's K1 numbers on the same command lines), K1x two-ahead tail fix, full GPU suite
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
python profiles/scripts/fuzz_forward.py 5 150 2>&1 | grep -v amdgpu | tail -6 > $O/r05j_fuzz_forward.txt
{
for r in 1 2; do
for args in "--workload ode01 --method rk4 --kernel tile" "--workload ode01 --method euler --kernel tile" "--workload ode01 --method rk4 --kernel wave" "--workload ode01 --method euler --kernel wave" "--workload ode01 --method midpoint --kernel wave" "--workload dae01 --method rk4" "--workload dae01 --method euler" "--workload ode01 --method rk4 --hidden 128" "--workload dae01 --method rk4 --hidden 128" "--workload ode01 --method rk4 --batch 16384" "--workload ode01 --method rk4 --train" "--workload dae01 --method rk4 --train"; do
  python bench.py $args --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('round $r [$args] kernel_ms %.4f frac %.4f  %s' % (d['roofline']['kernel_ms'], d['roofline']['frac'], d['config']['kernel']))"
done; done
} > $O/r05j_bench_matrix.txt 2>&1
python -m pytest tests/ -m gpu -q --tb=line 2>&1 | tail -12 > $O/r05j_pytest_all.txt
