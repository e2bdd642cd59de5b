#!/bin/bash
# round 4: one-role K4f instances with the two-stages-ahead request (variant sah3all) vs the shipped forms
B="python bench.py --no-cpu-baseline --no-extras --train --steps 10 --warmup 3"
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], "ms %.3f" % d["ms_per_step"])'
for m in rk4 euler; do
  $B --workload ode01 --method $m 2>/dev/null | tail -1 | python -c "$P" "ode01 $m h64 tree(roles)"
  PSNODE_K4F_NO_ROLES=1 $B --workload ode01 --method $m 2>/dev/null | tail -1 | python -c "$P" "ode01 $m h64 one-role"
  PSNODE_K4F_NO_ROLES=1 PSNODE_LIB_PATH=build/var_sah3all/lib.so $B --workload ode01 --method $m 2>/dev/null | tail -1 | python -c "$P" "ode01 $m h64 one-role sah3"
  $B --workload ode01 --method $m --hidden 128 2>/dev/null | tail -1 | python -c "$P" "ode01 $m h128 tree"
  PSNODE_LIB_PATH=build/var_sah3all/lib.so $B --workload ode01 --method $m --hidden 128 2>/dev/null | tail -1 | python -c "$P" "ode01 $m h128 sah3"
done
