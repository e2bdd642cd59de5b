#!/bin/bash
# round 4: K7f with the per-step inputs requested a step ahead; K4f / K7f prefetches kept raw
mkdir -p gpurun_out/r04g; O=gpurun_out/r04g
timeout 1500 python -m pytest tests/test_gpu_backward.py tests/test_grad_goldens.py tests/test_tf_goldens.py tests/test_gpu_fuzz.py tests/test_gpu_determinism.py -m gpu -q -x > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
B="python bench.py --no-cpu-baseline --no-extras --train --steps 10 --warmup 3"
for w in ode01 dae01; do for m in rk4 euler; do
  $B --workload $w --method $m 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w $m auto     ms %.3f' % d['ms_per_step'])"
done; done
PSNODE_SAVE_ACTIVATIONS=1 $B --workload dae01 --method euler 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dae01 euler save=1 ms %.3f' % d['ms_per_step'])"
PSNODE_SAVE_ACTIVATIONS=0 $B --workload dae01 --method euler 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dae01 euler save=0 ms %.3f' % d['ms_per_step'])"
PSNODE_SAVE_ACTIVATIONS=0 $B --workload dae01 --method rk4 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dae01 rk4 save=0 ms %.3f' % d['ms_per_step'])"
for h in 32 128; do $B --workload dae01 --hidden $h 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dae01 rk4 h$h ms %.3f' % d['ms_per_step'])"; done
for h in 32 128; do $B --workload ode01 --hidden $h 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ode01 rk4 h$h ms %.3f' % d['ms_per_step'])"; done
