set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 2400 python -m pytest tests -m gpu -q -x > $O/r03h_pytest_all.txt 2>&1; tail -4 $O/r03h_pytest_all.txt
python profiles/scripts/train_step_models.py dae02 ode02 dae01 > $O/r03h_train_step_models.txt 2>&1; tail -6 $O/r03h_train_step_models.txt
python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r03h_bench_default.json
python -c "
import json; d=json.load(open('$O/r03h_bench_default.json'))
print(d['ms_per_step'], d['roofline']['frac'])
for e in d['extra']: print(e['workload'], e['roofline']['kernel_ms'], round(e['roofline']['frac'],4))"
python bench.py --steps 10 --warmup 3 --workload dae01 --hidden 128 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dae01 h128 fwd ms', d['ms_per_step'], d['roofline']['frac'])"
