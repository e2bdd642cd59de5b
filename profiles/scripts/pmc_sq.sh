#!/bin/bash
# SQ issue/wait breakdown of one bench workload: two --pmc passes (8 SQ slots each) + GRBM.  usage: pmc_sq.sh <tag> <kernel substring> [bench args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; tag=$1; sub=$2; shift 2
B="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras $@"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES -d $O/pmc_${tag}_1 -o p -- $B > $O/pmc_${tag}_1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $O/pmc_${tag}_2 -o p -- $B > $O/pmc_${tag}_2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT -d $O/pmc_${tag}_3 -o p -- $B > $O/pmc_${tag}_3.log 2>&1
cd $R
: > $O/${tag}_pmc_sq.txt
for p in 1 2 3; do echo "[pass $p]" >> $O/${tag}_pmc_sq.txt; python profiles/summarize_pmc.py $O/pmc_${tag}_$p/p_results.db $sub 2>&1 | cut -c1-60,92-160 >> $O/${tag}_pmc_sq.txt; rm -rf $O/pmc_${tag}_$p; done
cat $O/${tag}_pmc_sq.txt
