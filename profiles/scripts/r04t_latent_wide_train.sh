#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 1200 python -m pytest tests/test_gpu_latent_wide.py -q -x -m gpu > $O/r04t_pytest.txt 2>&1; tail -5 $O/r04t_pytest.txt | cut -c1-300
timeout 900 python profiles/scripts/r04q_direct_encode_widths.py 2>&1 | grep -v amdgpu.ids | tail -10
