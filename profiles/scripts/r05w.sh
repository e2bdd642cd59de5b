#!/bin/bash
# round 5, call w: instruction-cache counters of K1x / K2x (is the unrolled RK4 step an I-fetch problem?)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
for w in ode01 dae01; do
  sub=integrate_x; [ $w = dae01 ] && sub=integrate_xd
  rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQC_TC_INST_REQ SQC_TC_STALL SQ_WAVE_CYCLES -d $O/pmc_ic_$w -o p -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --workload $w > $O/pmc_ic_$w.log 2>&1
  python $R/profiles/summarize_pmc.py $O/pmc_ic_$w/p_results.db $sub 2>&1 | cut -c1-60,92-160 > $O/r05w_${w}_icache_pmc.txt
  rm -rf $O/pmc_ic_$w $O/pmc_ic_$w.log
done
