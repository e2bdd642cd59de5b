set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q > $O/r03b_pytest_parity.txt 2>&1; tail -3 $O/r03b_pytest_parity.txt
( bash profiles/scripts/ab_libs.sh 3 0 "r02 tree exp2" --workload ode01
  bash profiles/scripts/ab_libs.sh 3 0 "r02 tree exp2" --workload ode01 --method euler
  bash profiles/scripts/ab_libs.sh 3 0 "r02 tree exp2" --workload dae01
  bash profiles/scripts/ab_libs.sh 3 0 "r02 tree exp2" --workload dae01 --method euler
  bash profiles/scripts/ab_libs.sh 0 1 "r02 tree exp2" ) > $O/r03b_elu_exp2_ab.txt 2>&1
cat $O/r03b_elu_exp2_ab.txt
