R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 1500 python -m pytest tests/test_grad_goldens.py tests/test_gpu_determinism.py -q -x -m gpu -k "h64 or dae02 or ode02" > $O/r03w_pytest.txt 2>&1; tail -4 $O/r03w_pytest.txt | cut -c1-300
python profiles/scripts/train_step_models.py dae02 2>/dev/null | grep -v amdgpu | tee $O/r03w_models.txt
