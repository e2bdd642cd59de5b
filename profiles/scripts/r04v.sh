#!/bin/bash
# round 4: two-role K4f -- chain-only timing (gradient waves ablated) and the SQ breakdown of the shipped form
B="python bench.py --no-cpu-baseline --no-extras --train --steps 10 --warmup 3"
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], "ms %.3f" % d["ms_per_step"])'
for m in rk4 euler; do
PSNODE_LIB_PATH=build/var_gabl/lib.so $B --workload ode01 --method $m 2>/dev/null | tail -1 | python -c "$P" "ode01 $m h64 chain-only(gradient waves idle)"
$B --workload ode01 --method $m 2>/dev/null | tail -1 | python -c "$P" "ode01 $m h64 tree"
done
bash profiles/scripts/pmc_sq.sh r04v_k4f_roles_rk4 ode_backward_fused --train --steps 2 --warmup 1 > /dev/null
cat gpurun_out/r04v_k4f_roles_rk4_pmc_sq.txt
