#!/bin/bash
# Same-box A/B of prebuilt library variants (built in the build container into build/var_<name>/lib.so, shipped with the snapshot;
# `tree` = py_psnode_amd/libpsnode_hip.so), interleaved so that clock drift hits every arm alike.
#   usage: ab_libs.sh ROUNDS ACC(0|1) "name1 name2 ..." [bench args...]        (ACC=1: accuracy report per arm)
R=${GRAFT_REPO_ROOT:-/root/repo}; ROUNDS=$1; ACC=$2; NAMES=$3; shift 3
lib() { if [ "$1" = tree ]; then echo $R/py_psnode_amd/libpsnode_hip.so; else echo $R/build/var_$1/lib.so; fi; }
for r in $(seq $ROUNDS); do
  for name in $NAMES; do
    PSNODE_LIB_PATH=$(lib $name) python $R/bench.py "$@" --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$*] round $r %-10s kernel_ms %.4f (median %.4f)  frac %.4f' % ('$name', d['roofline']['kernel_ms'], d['roofline']['kernel_ms_median'], d['roofline']['frac']))"
  done
done
if [ "$ACC" = 1 ]; then for name in $NAMES; do echo "-- accuracy $name"; PSNODE_LIB_PATH=$(lib $name) python $R/profiles/scripts/accuracy_report.py 2>&1 | grep -v amdgpu.ids | sed 's/^/   /'; done; fi
