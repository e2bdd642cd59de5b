#!/bin/bash
# round 4: two-role vs one-role backward kernels at larger batches (several tiles per CU)
R=$GRAFT_REPO_ROOT; cd $R
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], "ms %.3f  state-steps/s %.3e" % (d["ms_per_step"], d["value"]))'
for Bt in 8192 16384 32768; do
  for w in ode01 dae01; do
    B="python bench.py --no-cpu-baseline --no-extras --train --steps 4 --warmup 2 --workload $w --batch $Bt --grid 501"
    $B 2>/dev/null | tail -1 | python -c "$P" "$w rk4 B=$Bt T=501 two-role"
    PSNODE_K4F_NO_ROLES=1 PSNODE_K7F_NO_ROLES=1 $B 2>/dev/null | tail -1 | python -c "$P" "$w rk4 B=$Bt T=501 one-role"
  done
done
