set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
( bash profiles/scripts/ab_libs.sh 3 0 "pre1 tree pre3" --workload ode01
  bash profiles/scripts/ab_libs.sh 2 0 "pre1 tree pre3" --workload dae01 ) > $O/r03j_mid_pre_ab.txt 2>&1
cat $O/r03j_mid_pre_ab.txt
