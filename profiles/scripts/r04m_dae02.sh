#!/bin/bash
# round 4: K3g two-role form (8 waves, lock-step barriers): parity, timing vs the one-role form and the row-kernel route
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 600 python -m pytest tests/test_gpu_dae_encoded.py -q -x -m gpu > $O/r04m_pytest.txt 2>&1; tail -3 $O/r04m_pytest.txt | cut -c1-300
B="timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 --workload dae02"
for m in rk4 euler midpoint; do
  $B --method $m 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dae02 $m two-role ms %.3f' % (d['ms_per_step']))"
  PSNODE_K3G_ONE_ROLE=1 $B --method $m 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dae02 $m one-role ms %.3f' % (d['ms_per_step']))"
  PSNODE_DAE02_ONE_LAUNCH=0 $B --method $m 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dae02 $m row kernels + K3c ms %.3f' % (d['ms_per_step']))"
done
