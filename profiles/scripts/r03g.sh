set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 2400 python -m pytest tests -m gpu -q -x > $O/r03g_pytest_all.txt 2>&1; tail -6 $O/r03g_pytest_all.txt
python profiles/scripts/train_step_models.py > $O/r03g_train_step_models.txt 2>&1; cp $O/train_step_models.json $O/r03g_train_step_models.json; tail -12 $O/r03g_train_step_models.txt
python bench.py --steps 10 --warmup 3 --workload dae01 --hidden 128 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dae01 h128 fwd ms', d['ms_per_step'], d['roofline']['frac'])"
python bench.py --steps 10 --warmup 3 --hidden 128 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ode01 h128 fwd ms', d['ms_per_step'], d['roofline']['frac'])"
python bench.py --steps 5 --warmup 2 --train --workload dae01 --hidden 128 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dae01 h128 train ms', d['ms_per_step'])"
