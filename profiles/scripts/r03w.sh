R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 1500 python -m pytest tests/test_gpu_backward.py -q -x -k "dae_wide or zero_padded" > $O/r03w_pytest.txt 2>&1; tail -30 $O/r03w_pytest.txt | cut -c1-300
for sv in 1 0; do for m in rk4 euler; do for h in 128 32; do
PSNODE_SAVE_ACTIVATIONS=$sv python bench.py --steps 4 --warmup 2 --train --workload dae01 --hidden $h --method $m --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('save=$sv dae01 h$h $m train ms', round(d['ms_per_step'],3))"
done; done; done 2>&1 | tee $O/r03w_bench.txt
