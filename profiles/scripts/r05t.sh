#!/bin/bash
# round 5, call t: eight-chain fixed-order reductions (row-MLP backward, reduce_partials in two launches, loss finish): the whole GPU suite, glue, steps
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
python -m pytest tests/ -m gpu -q --tb=short 2>&1 | tail -15 > $O/r05t_pytest.txt
python profiles/scripts/glue_trace_model.py ode02 rk4 2>&1 | grep -v "Warning\|warn\|amdgpu" > $O/r05t_glue_ode02.txt
python - > $O/r05t_model_train.txt 2>&1 <<'PY'
import json, torch, bench
dev = torch.device("cuda", 0)
for wl, m in (("ode02", "rk4"), ("ode02", "euler"), ("dae02", "rk4"), ("dae02", "euler")):
    r = bench.model_train_extra_line(wl, m, dev)
    print(json.dumps({k: v for k, v in r.items() if not isinstance(v, (dict, list))}))
PY
for m in rk4 euler; do python bench.py --workload ode01 --method $m --train --steps 8 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ode01 $m TRAIN ms_per_step %.4f frac %.4f' % (d['ms_per_step'], d['roofline']['frac']))"; done >> $O/r05t_model_train.txt
