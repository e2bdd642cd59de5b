R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 1500 python -m pytest tests/test_gpu_backward.py -q -x -k "head_grads or dae_hidden128 or hidden128_training" > $O/r03w_pytest.txt 2>&1; tail -30 $O/r03w_pytest.txt | cut -c1-300
