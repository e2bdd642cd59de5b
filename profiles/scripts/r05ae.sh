#!/bin/bash
# round 5, call ae: row kernels walk a time-major view in 4 x 4 (grid point, trajectory) tiles: tests, ODE_02 / DAE_02 step, glue
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
python -m pytest tests/test_gpu_rows_backward.py tests/test_grad_goldens.py tests/test_gpu_encoded.py tests/test_gpu_dae_encoded.py tests/test_gpu_determinism.py tests/test_gpu_example.py -m gpu -q --tb=short 2>&1 | tail -8 > $O/r05ae_pytest.txt
python profiles/scripts/glue_trace_model.py ode02 rk4 2>&1 | grep -v "Warning\|warn\|amdgpu" > $O/r05ae_glue_ode02.txt
python - > $O/r05ae_model_train.txt 2>&1 <<'PY'
import json, torch, bench
dev = torch.device("cuda", 0)
for wl, m in (("ode02", "rk4"), ("ode02", "euler"), ("dae02", "rk4"), ("dae02", "euler")):
    r = bench.model_train_extra_line(wl, m, dev)
    print(json.dumps({k: v for k, v in r.items() if not isinstance(v, (dict, list))}))
PY
