#!/bin/bash
# round 4: hidden 129..256 on the MFMA integrators (streamed H->H weights): parity tests and timings vs K0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "streamed or hidden_widths or kernel_for or dispatch or shapes" > $O/r04j_pytest.txt 2>&1; tail -4 $O/r04j_pytest.txt | cut -c1-300
B="python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3"
for h in 128 192 256; do for k in mfma generic; do
  timeout 600 $B --hidden $h --kernel $k 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ode01 rk4 h$h $k ms %.3f frac %.3f' % (d['ms_per_step'], d['roofline']['frac']))"
done; done
for h in 160 192; do for k in mfma generic; do
  timeout 600 $B --workload dae01 --hidden $h --kernel $k 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dae01 rk4 h$h $k ms %.3f frac %.3f' % (d['ms_per_step'], d['roofline']['frac']))"
done; done
timeout 600 $B --hidden 256 --method euler 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ode01 euler h256 ms %.3f frac %.3f' % (d['ms_per_step'], d['roofline']['frac']))"
timeout 600 $B --hidden 256 --batch 8192 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ode01 rk4 h256 B8192 ms %.3f frac %.3f' % (d['ms_per_step'], d['roofline']['frac']))"
