set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_determinism.py -q -x > $O/r03k_pytest.txt 2>&1; tail -3 $O/r03k_pytest.txt
( bash profiles/scripts/ab_libs.sh 3 0 "elumax tree" --workload ode01
  bash profiles/scripts/ab_libs.sh 3 0 "elumax tree" --workload ode01 --method euler
  bash profiles/scripts/ab_libs.sh 2 0 "elumax tree" --workload dae01
  bash profiles/scripts/ab_libs.sh 2 0 "elumax tree" --workload dae01 --method euler
  bash profiles/scripts/ab_libs.sh 2 0 "elumax tree" --workload ode02
  bash profiles/scripts/ab_libs.sh 0 1 "tree" ) > $O/r03k_ab.txt 2>&1
cat $O/r03k_ab.txt
