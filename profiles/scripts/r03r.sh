set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
lib() { if [ "$1" = tree ]; then echo $R/py_psnode_amd/libpsnode_hip.so; else echo $R/build/var_$1/lib.so; fi; }
( for r in 1 2; do for v in tree se1 se4 se8; do for m in rk4 euler; do
  PSNODE_LIB_PATH=$(lib $v) python bench.py --steps 4 --warmup 2 --train --hidden 128 --method $m --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('round $r $v saved h128 $m train ms', round(d['ms_per_step'],3))"
done; done; done ) 2>/dev/null | grep "train ms" > $O/r03r_saved_every_ab.txt
cat $O/r03r_saved_every_ab.txt
bash profiles/scripts/pmc_sq.sh r03r_k4f_saved_h128 ode_backward_fused --train --hidden 128 --steps 2 --warmup 1 > /dev/null; rm -f $O/pmc_r03r*.log
