R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 1800 python -m pytest tests/test_gpu_rows_backward.py tests/test_export.py tests/test_host_models.py tests/test_gpu_loss.py tests/test_datapath.py -q -x -m gpu > $O/r03w_pytest.txt 2>&1; tail -4 $O/r03w_pytest.txt | cut -c1-300
