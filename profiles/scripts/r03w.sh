R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 1500 python -m pytest tests/test_grad_goldens.py tests/test_gpu_determinism.py -q -x -m gpu > $O/r03w_pytest.txt 2>&1; tail -8 $O/r03w_pytest.txt | cut -c1-300
for sv in auto 0; do PSNODE_SAVE_ACTIVATIONS=$sv python profiles/scripts/train_step_models.py dae02 ode02 2>/dev/null | grep -v amdgpu | sed "s/^/save=$sv /"; done | tee $O/r03w_models.txt
HIDDEN=64 PSNODE_SAVE_ACTIVATIONS=auto python profiles/scripts/train_step_models.py ode02 2>/dev/null | grep -v amdgpu | sed "s/^/h64 save=auto /" | tee -a $O/r03w_models.txt
HIDDEN=64 PSNODE_SAVE_ACTIVATIONS=0 python profiles/scripts/train_step_models.py ode02 2>/dev/null | grep -v amdgpu | sed "s/^/h64 save=0 /" | tee -a $O/r03w_models.txt
