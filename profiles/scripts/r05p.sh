#!/bin/bash
# round 5, call p: ODE_02 training after the in-place time-major row reads (ABI 9) and integrate_ODE's x_init: tests, ATen glue, step time
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
python -m pytest tests/test_gpu_rows_backward.py tests/test_grad_goldens.py tests/test_gpu_encoded.py tests/test_host_models.py tests/test_gpu_example.py tests/test_gpu_parity.py -m gpu -q --tb=short 2>&1 | tail -15 > $O/r05p_pytest.txt
python profiles/scripts/glue_trace_model.py ode02 rk4 2>&1 | grep -v Warning > $O/r05p_glue_ode02.txt
python profiles/scripts/glue_trace_model.py dae02 rk4 2>&1 | grep -v Warning > $O/r05p_glue_dae02.txt
python - > $O/r05p_model_train.txt 2>&1 <<'PY'
import json, torch, bench
dev = torch.device("cuda", 0)
for wl, m in (("ode02", "rk4"), ("ode02", "euler"), ("dae02", "rk4")):
    r = bench.model_train_extra_line(wl, m, dev)
    print(json.dumps({k: r[k] for k in r if k in ("workload", "ms_per_step", "value", "roofline_frac", "frac", "kernel_ms")} | {"all": {k: v for k, v in r.items() if not isinstance(v, (dict, list))}}))
PY
