#!/bin/bash
# round 4: K7f two-role form (chain + gradient waves) at <= 4 waves, saved activations; PSNODE_K7F_NO_ROLES=1 = the one-role instances
# usage: r04x_k7f_roles.sh "variant names"
mkdir -p gpurun_out/r04x; O=gpurun_out/r04x
timeout 1500 python -m pytest tests/test_gpu_backward.py tests/test_grad_goldens.py tests/test_gpu_determinism.py -m gpu -q -x -k "dae" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
B="python bench.py --no-cpu-baseline --no-extras --train --steps 10 --warmup 3"
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], "ms %.3f" % d["ms_per_step"])'
lib() { if [ "$1" = tree ]; then echo py_psnode_amd/libpsnode_hip.so; else echo build/var_$1/lib.so; fi; }
for m in rk4 euler; do
  for v in ${1:-tree}; do
    PSNODE_LIB_PATH=$(lib $v) $B --workload dae01 --method $m 2>/dev/null | tail -1 | python -c "$P" "dae01 $m h64 $v"
  done
  PSNODE_K7F_NO_ROLES=1 $B --workload dae01 --method $m 2>/dev/null | tail -1 | python -c "$P" "dae01 $m h64 one-role"
done
for v in ${1:-tree}; do
  PSNODE_LIB_PATH=$(lib $v) $B --workload dae01 --hidden 32 2>/dev/null | tail -1 | python -c "$P" "dae01 rk4 h32 $v"
done
PSNODE_K7F_NO_ROLES=1 $B --workload dae01 --hidden 32 2>/dev/null | tail -1 | python -c "$P" "dae01 rk4 h32 one-role"
