set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
B="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $O/r01b_mfma_trace -o t -- $B > $O/r01b_mfma_trace.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $O/r01b_mfma_pmc1 -o p -- $B > $O/r01b_mfma_pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $O/r01b_mfma_pmc2 -o p -- $B > $O/r01b_mfma_pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/r01b_mfma_fetch -o p -- $B > $O/r01b_mfma_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/r01b_mfma_write -o p -- $B > $O/r01b_mfma_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TCC_HIT_sum TCC_MISS_sum -d $O/r01b_mfma_pmc3 -o p -- $B > $O/r01b_mfma_pmc3.log 2>&1
for b in 8192 16384 32768; do $B --batch $b 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B', d['config']['trajectories_total'], 'value', d['value'], 'kernel_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'])"; done
for m in euler midpoint; do $B --method $m 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['metric'], 'value', d['value'], 'kernel_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'])"; done
python $R/bench.py --steps 10 --warmup 2 2>&1 | tail -1 > $O/r01b_bench_mfma.json; cat $O/r01b_bench_mfma.json
