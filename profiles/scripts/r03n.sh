set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
lib() { if [ "$1" = tree ]; then echo $R/py_psnode_amd/libpsnode_hip.so; else echo $R/build/var_$1/lib.so; fi; }
( for r in 1 2 3; do for v in tree k7pre2 k7pre3; do
  PSNODE_LIB_PATH=$(lib $v) python bench.py --steps 5 --warmup 2 --train --workload dae01 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('round $r $v dae01 h64 rk4 train ms', round(d['ms_per_step'],3))"
done; done
for r in 1 2; do for v in tree every1 every4; do for m in rk4 midpoint; do
  PSNODE_LIB_PATH=$(lib $v) python bench.py --steps 4 --warmup 2 --train --hidden 128 --method $m --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('round $r $v h128 $m train ms', round(d['ms_per_step'],3))"
done; done; done ) 2>/dev/null | grep "^round" > $O/r03n_bwd_ab.txt
cat $O/r03n_bwd_ab.txt
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_determinism.py -q -x > $O/r03n_pytest.txt 2>&1; tail -3 $O/r03n_pytest.txt
