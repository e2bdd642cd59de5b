#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
bash profiles/scripts/pmc_sq.sh r04n_k3g2_rk4 latent64_model2 --workload dae02 > /dev/null 2>&1
PSNODE_DAE02_ONE_LAUNCH=0 bash profiles/scripts/pmc_sq.sh r04n_k3c_rk4 "latent64_kernel" --workload dae02 > /dev/null 2>&1
grep -h "latent64" $O/r04n_k3g2_rk4_pmc_sq.txt $O/r04n_k3c_rk4_pmc_sq.txt | awk '{print $(NF-4), $(NF-2), $NF}' | cut -c1-120
cd /tmp && export TMPDIR=/tmp
PSNODE_DAE02_ONE_LAUNCH=0 rocprofv3 --kernel-trace --stats -d $O/r04n_kt -o t -- python $R/bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2 --workload dae02 > $O/r04n_kt.log 2>&1
python $R/profiles/summarize_rocprof.py $O/r04n_kt/t_results.db > $O/r04n_dae02_rowroute_kernel_stats.txt; rm -rf $O/r04n_kt
head -14 $O/r04n_dae02_rowroute_kernel_stats.txt | cut -c1-150
