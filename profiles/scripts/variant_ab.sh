#!/bin/bash
# A/B of whole-library build variants on one box, interleaved so that clock drift hits every arm alike.
# usage: bash profiles/scripts/variant_ab.sh [-w "ode01 dae01"] [-r ROUNDS] "name:EXTRA flags" ...
# Each variant is `make BUILD=/tmp/var_<name> OUT=/tmp/var_<name>/lib.so EXTRA="<flags>"`; the bench loads it through
# PSNODE_LIB_PATH.  Accuracy of every arm is printed too (profiles/scripts/accuracy_report.py).
R=${GRAFT_REPO_ROOT:-/root/repo}
WL="ode01"; ROUNDS=3
while getopts "w:r:" o; do case $o in w) WL="$OPTARG";; r) ROUNDS=$OPTARG;; esac; done; shift $((OPTIND-1))
names=()
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}; names+=($name)
  make -s -C $R/py_psnode_amd/csrc -j16 BUILD=/tmp/var_$name OUT=/tmp/var_$name/lib.so EXTRA="$flags" > /tmp/var_$name.log 2>&1 || { echo "BUILD FAILED $name"; tail -5 /tmp/var_$name.log; }
  echo "== built $name ($flags)"
done
for wl in $WL; do
  for r in $(seq $ROUNDS); do
    for name in "${names[@]}"; do
      PSNODE_LIB_PATH=/tmp/var_$name/lib.so python $R/bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl round $r %-14s kernel_ms %.3f  value %.4g  frac %.4f' % ('$name', d['roofline']['kernel_ms'], d['value'], d['roofline']['frac']))"
    done
  done
done
for name in "${names[@]}"; do
  echo "-- accuracy $name"; PSNODE_LIB_PATH=/tmp/var_$name/lib.so python $R/profiles/scripts/accuracy_report.py 2>&1 | sed 's/^/   /'
done
