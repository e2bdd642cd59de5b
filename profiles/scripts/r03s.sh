R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
lib() { if [ "$1" = tree ]; then echo $R/py_psnode_amd/libpsnode_hip.so; else echo $R/build/var_$1/lib.so; fi; }
( for v in tree ab1 ab2 ab3; do for sv in auto 0; do
  PSNODE_SAVE_ACTIVATIONS=$sv PSNODE_LIB_PATH=$(lib $v) python bench.py --steps 4 --warmup 2 --train --hidden 128 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v save=$sv h128 rk4 train ms', round(d['ms_per_step'],3))"
done; done ) 2>/dev/null | grep "train ms" > $O/r03s_ablate.txt
cat $O/r03s_ablate.txt
