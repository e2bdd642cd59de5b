set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 1500 python -m pytest tests/test_gpu_backward.py -x -q > $O/r03c_pytest_backward.txt 2>&1; tail -5 $O/r03c_pytest_backward.txt
for h in 128 32; do python bench.py --steps 5 --warmup 2 --train --hidden $h --no-cpu-baseline 2>/dev/null | tail -1 > $O/r03c_bench_ode01_h${h}_train.json; python -c "import json; d=json.load(open('$O/r03c_bench_ode01_h${h}_train.json')); print('train h$h ms', d['ms_per_step'])"; done
python bench.py --steps 5 --warmup 2 --train --hidden 128 --method euler --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train h128 euler ms', d['ms_per_step'])"
python bench.py --steps 5 --warmup 2 --train --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train h64 (K4) ms', d['ms_per_step'])"
python bench.py --steps 5 --warmup 2 --train --kernel wide --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train h64 (K4f) ms', d['ms_per_step'])"
timeout 2400 python -m pytest tests -m gpu -q --deselect tests/test_gpu_backward.py > $O/r03c_pytest_rest.txt 2>&1; tail -15 $O/r03c_pytest_rest.txt
python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r03c_bench_default.json
python -c "
import json; d=json.load(open('$O/r03c_bench_default.json'))
print(d['ms_per_step'], d['roofline']['frac'])
for e in d['extra']: print(e['workload'], e['roofline']['kernel_ms'], round(e['roofline']['frac'],4))"
