R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
for sv in auto 1; do for m in rk4 midpoint euler; do
PSNODE_SAVE_ACTIVATIONS=$sv python bench.py --steps 5 --warmup 2 --train --workload dae01 --method $m --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('save=$sv dae01 h64 $m train ms', round(d['ms_per_step'],3))"
done; done 2>&1 | tee $O/r03z_h64.txt
