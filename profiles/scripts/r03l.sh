set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 900 python -m pytest tests/test_gpu_backward.py -q -x -k "wide or hidden or edge" > $O/r03l_pytest.txt 2>&1; tail -3 $O/r03l_pytest.txt
for r in 1 2; do for v in nopipe tree; do for m in rk4 midpoint euler; do
  L=$R/build/var_$v/lib.so; [ $v = tree ] && L=$R/py_psnode_amd/libpsnode_hip.so
  PSNODE_LIB_PATH=$L python bench.py --steps 4 --warmup 2 --train --hidden 128 --method $m --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('round $r $v h128 $m train ms', round(d['ms_per_step'],3))"
done; done; done
