#!/bin/bash
# round 4: K9 two-role form (DAE_02 latent backward at hidden 64, saved activations); PSNODE_K9_NO_ROLES=1 = the one-role kernel
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out
timeout 1500 python -m pytest tests/test_grad_goldens.py tests/test_gpu_backward.py tests/test_gpu_rows_backward.py tests/test_gpu_determinism.py tests/test_gpu_dae_encoded.py -m gpu -q -x -k "dae02 or latent or encoded or direct or model" 2>&1 | tail -4 | cut -c1-300
python profiles/scripts/train_step_models.py dae02 2>&1 | grep -v amdgpu
PSNODE_K9_NO_ROLES=1 python profiles/scripts/train_step_models.py dae02 2>&1 | grep -v amdgpu | sed 's/^/one-role: /'
