# Round-1 final profiles: rocprofv3 kernel-trace stats for the bench lines quoted in DESIGN.md, PMC HBM traffic for K1 and K6.
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
B="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $O/r01d_ode01 -o t -- $B > $O/r01d_ode01.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/r01d_dae01 -o t -- $B --workload dae01 > $O/r01d_dae01.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/r01d_ode01_h128 -o t -- $B --hidden 128 > $O/r01d_ode01_h128.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/r01d_train_ode01 -o t -- $B --train --steps 5 > $O/r01d_train_ode01.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/r01d_train_dae01 -o t -- $B --train --workload dae01 --steps 5 > $O/r01d_train_dae01.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/r01d_ode01_fetch -o p -- $B > $O/r01d_ode01_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/r01d_ode01_write -o p -- $B > $O/r01d_ode01_write.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/r01d_loss_fetch -o p -- python $R/profiles/scripts/loss_bench.py > $O/r01d_loss_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/r01d_loss_write -o p -- python $R/profiles/scripts/loss_bench.py > $O/r01d_loss_write.log 2>&1
cd $R
python bench.py --steps 20 --warmup 3 2>/dev/null | tail -1 > $O/r01d_bench_ode01_n1.json
python bench.py --steps 10 --workload dae01 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r01d_bench_dae01_n1.json
python bench.py --steps 5 --train --no-cpu-baseline 2>/dev/null | tail -1 > $O/r01d_bench_ode01_train_n1.json
python bench.py --steps 5 --train --workload dae01 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r01d_bench_dae01_train_n1.json
python bench.py --steps 10 --workload ode02 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r01d_bench_ode02_n1.json
for d in ode01 dae01 ode01_h128 train_ode01 train_dae01; do python profiles/summarize_rocprof.py $O/r01d_$d/t_results.db > $O/r01d_${d}_kernel_stats.txt; done
for d in ode01_fetch ode01_write; do python profiles/summarize_pmc.py $O/r01d_$d/p_results.db integrate_mfma > $O/r01d_${d}_pmc.txt; done
for d in loss_fetch loss_write; do python profiles/summarize_pmc.py $O/r01d_$d/p_results.db masked_mse > $O/r01d_${d}_pmc.txt; done
rm -rf $O/r01d_ode01 $O/r01d_dae01 $O/r01d_ode01_h128 $O/r01d_train_ode01 $O/r01d_train_dae01 $O/r01d_ode01_fetch $O/r01d_ode01_write $O/r01d_loss_fetch $O/r01d_loss_write
ls $O | grep r01d
