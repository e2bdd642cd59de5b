#!/bin/bash
# round 5, call al: K3r, the reconstruction branch of ODE_02 in one kernel each way: tests, ODE_02 training step, glue
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
python -m pytest tests/test_gpu_rows_backward.py tests/test_grad_goldens.py tests/test_gpu_encoded.py tests/test_gpu_determinism.py tests/test_gpu_example.py tests/test_host_models.py tests/test_capi_exports.py -m gpu -q --tb=short 2>&1 | tail -12 > $O/r05al_pytest.txt
python profiles/scripts/fuzz_models.py 41 150 2>&1 | grep -v amdgpu | tail -2 >> $O/r05al_pytest.txt
python profiles/scripts/glue_trace_model.py ode02 rk4 2>&1 | grep -v "Warning\|warn\|amdgpu" > $O/r05al_glue_ode02.txt
python - > $O/r05al_model_train.txt 2>&1 <<'PY'
import json, torch, bench
dev = torch.device("cuda", 0)
for wl, m in (("ode02", "rk4"), ("ode02", "euler")):
    r = bench.model_train_extra_line(wl, m, dev)
    print(json.dumps({k: v for k, v in r.items() if not isinstance(v, (dict, list))}))
PY
