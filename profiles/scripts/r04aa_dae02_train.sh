#!/bin/bash
# round 4: DAE_02 training step at hidden 64 -- kernel breakdown (rocprofv3 --kernel-trace --stats) and the SQ view of K9
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-r04aa}
rocprofv3 --kernel-trace --stats -d $O/${TAG}_d -o t -- python $R/profiles/scripts/train_step_models.py dae02 > $O/${TAG}_train.log 2>&1
python $R/profiles/summarize_rocprof.py $O/${TAG}_d/t_results.db > $O/${TAG}_train_dae02_kernel_stats.txt; rm -rf $O/${TAG}_d
grep -v amdgpu $O/${TAG}_train.log | tail -2
head -24 $O/${TAG}_train_dae02_kernel_stats.txt | cut -c1-150
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES -d $O/${TAG}_p1 -o p -- python $R/profiles/scripts/train_step_models.py dae02 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $O/${TAG}_p2 -o p -- python $R/profiles/scripts/train_step_models.py dae02 > /dev/null 2>&1
cd $R
for p in 1 2; do echo "[pass $p]"; python profiles/summarize_pmc.py $O/${TAG}_p$p/p_results.db latent64_backward 2>&1 | cut -c1-60,92-160; rm -rf $O/${TAG}_p$p; done > $O/${TAG}_k9_pmc_sq.txt
cat $O/${TAG}_k9_pmc_sq.txt
