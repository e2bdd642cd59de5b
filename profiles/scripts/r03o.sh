set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
PSNODE_LIB_PATH=$R/build/var_k9touch/lib.so timeout 900 python -m pytest tests/test_gpu_determinism.py tests/test_gpu_rows_backward.py tests/test_grad_goldens.py -q -x > $O/r03o_pytest.txt 2>&1; tail -3 $O/r03o_pytest.txt
( for r in 1 2; do for v in tree k9touch; do
  L=$R/build/var_$v/lib.so; [ $v = tree ] && L=$R/py_psnode_amd/libpsnode_hip.so
  PSNODE_LIB_PATH=$L python profiles/scripts/train_step_models.py dae02 2>/dev/null | grep "^dae02" | sed "s/^/round $r $v /" | cut -c1-90
done; done ) > $O/r03o_k9_touch_ab.txt 2>&1
cat $O/r03o_k9_touch_ab.txt
