#!/bin/bash
# round 5, call h: K1x with two steps of look-ahead at Euler / Midpoint; the GPU suite on the pruned tree; ATen glue of the ODE_02 training step
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
for r in 1 2; do for k in tile wave; do for m in rk4 euler midpoint; do
  python bench.py --workload ode01 --method $m --kernel $k --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('round $r $k $m kernel_ms %.4f frac %.4f' % (d['roofline']['kernel_ms'], d['roofline']['frac']))"
done; done; done
} > $O/r05h_tile_vs_wave.txt 2>&1
python profiles/scripts/glue_trace_model.py ode02 rk4 > $O/r05h_glue_ode02.txt 2>&1
python -m pytest tests/ -m gpu -q --tb=line 2>&1 | tail -15 > $O/r05h_pytest_all.txt
