R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
lib() { if [ "$1" = tree ]; then echo $R/py_psnode_amd/libpsnode_hip.so; else echo $R/build/var_$1/lib.so; fi; }
( for r in 1 2; do for v in tree rl1; do for m in rk4 midpoint; do
  PSNODE_SAVE_ACTIVATIONS=0 PSNODE_LIB_PATH=$(lib $v) python bench.py --steps 4 --warmup 2 --train --hidden 128 --method $m --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('round $r $v recompute h128 $m train ms', round(d['ms_per_step'],3))"
done; done; done ) 2>/dev/null | grep "train ms" > $O/r03v_ring_late.txt
cat $O/r03v_ring_late.txt
timeout 1200 python -m pytest tests/test_gpu_backward.py tests/test_grad_goldens.py tests/test_gpu_determinism.py -q -x > $O/r03v_pytest.txt 2>&1; tail -3 $O/r03v_pytest.txt
