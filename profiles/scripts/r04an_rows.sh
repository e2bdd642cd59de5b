#!/bin/bash
# round 4: decoder backward (rows_bwd_kernel<16,4,1>) with the first layer's operand sets in LDS: 373 -> 252 registers, two waves per SIMD
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_rows_backward.py tests/test_gpu_determinism.py tests/test_grad_goldens.py -m gpu -q -x 2>&1 | tail -2
python profiles/scripts/train_step_models.py dae02 2>&1 | grep -v amdgpu
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r04an_d -o t -- python $R/profiles/scripts/train_step_models.py dae02 > /dev/null 2>&1
python $R/profiles/summarize_rocprof.py $R/gpurun_out/r04an_d/t_results.db > $R/gpurun_out/r04an_train_dae02_kernel_stats.txt; rm -rf $R/gpurun_out/r04an_d
head -12 $R/gpurun_out/r04an_train_dae02_kernel_stats.txt | cut -c1-130
