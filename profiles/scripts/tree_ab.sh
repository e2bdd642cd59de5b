#!/bin/bash
# A/B of the working tree against a git revision on ONE box (box-to-box clock spread is 2-3 %: only same-call comparisons count).
# usage (on the GPU box): bash profiles/scripts/tree_ab.sh <path to an exported csrc/ + include/ of the other revision>
R=$GRAFT_REPO_ROOT; OLD=$1
make -s -C $R/py_psnode_amd/csrc -j16 BUILD=/tmp/ab_new OUT=/tmp/ab_new/lib.so > /tmp/ab_new.log 2>&1 || tail -3 /tmp/ab_new.log
make -s -C $OLD/py_psnode_amd/csrc -j16 BUILD=/tmp/ab_old OUT=/tmp/ab_old/lib.so > /tmp/ab_old.log 2>&1 || tail -3 /tmp/ab_old.log
for r in 1 2 3; do for v in old new; do for wm in "ode01 rk4" "ode01 euler" "dae01 rk4"; do set -- $wm
  PSNODE_LIB_PATH=/tmp/ab_$v/lib.so python $R/bench.py --workload $1 --method $2 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v $1 $2 %.3f' % d['roofline']['kernel_ms'])"
done; done; done
