# MFMA-pipe utilisation of the MFMA kernels (SQ_VALU_MFMA_BUSY_CYCLES vs GRBM_GUI_ACTIVE, one --pmc pass per workload).
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
B="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
run() { rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES -d $O/pmc_$1 -o p -- "${@:2}" > $O/pmc_$1.log 2>&1; }
run k1_h64 $B
run k1_h128 $B --hidden 128
run k2 $B --workload dae01
run train_ode01 $B --train
run train_dae01 $B --train --workload dae01
run k3c $B --workload ode02_latent64
run train_latent64 $B --train --workload ode02_latent64
cd $R
: > $O/r01d_mfma_util_pmc.txt
for d in k1_h64 k1_h128 k2 train_ode01 train_dae01 k3c train_latent64; do
  echo "## $d" >> $O/r01d_mfma_util_pmc.txt
  python profiles/summarize_pmc.py $O/pmc_$d/p_results.db psnode >> $O/r01d_mfma_util_pmc.txt
  rm -rf $O/pmc_$d
done
grep -v "pack\|event_table\|reduce\|masked" $O/r01d_mfma_util_pmc.txt | cut -c1-170
