set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 1200 python -m pytest tests/test_gpu_determinism.py tests/test_gpu_parity.py -q -x > $O/r03i_pytest.txt 2>&1; tail -4 $O/r03i_pytest.txt
( bash profiles/scripts/ab_libs.sh 3 0 "pre0 pre1 pre2 pre4" --workload ode01
  bash profiles/scripts/ab_libs.sh 2 0 "pre0 pre1 pre2 pre4" --workload ode01 --method euler
  bash profiles/scripts/ab_libs.sh 2 0 "pre0 pre1 pre2 pre4" --workload dae01
  bash profiles/scripts/ab_libs.sh 2 0 "pre0 pre1 pre2 pre4" --workload ode01 --batch 32768 ) > $O/r03i_mid_pre_ab.txt 2>&1
cat $O/r03i_mid_pre_ab.txt
