set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 1500 python -m pytest tests/test_gpu_backward.py tests/test_grad_goldens.py -x -q > $O/r03e_pytest.txt 2>&1; tail -5 $O/r03e_pytest.txt
for m in rk4 midpoint euler; do python bench.py --steps 5 --warmup 2 --train --hidden 128 --method $m --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train h128 $m ms', d['ms_per_step'])"; done
python bench.py --steps 5 --warmup 2 --train --hidden 32 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train h32 rk4 ms', d['ms_per_step'])"
python bench.py --steps 5 --warmup 2 --train --hidden 100 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train h100 rk4 ms', d['ms_per_step'])"
bash profiles/scripts/pmc_sq.sh r03e_k4f_h128_rk4 ode_backward_fused --train --hidden 128 --steps 2 --warmup 1 > /dev/null
rm -f $O/pmc_r03e*.log
