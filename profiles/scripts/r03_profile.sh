# Round-3 profiles: rocprofv3 kernel-trace stats for the default bench line (headline + the four extra workloads), PMC HBM traffic
# (separate FETCH / WRITE passes) for all five, SQ breakdowns, accuracy report, batch sweep, training-step tables, hidden-128 training.
#   gpurun -- 'bash profiles/scripts/r03_profile.sh'   then copy gpurun_out/r03_* into profiles/
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
B="python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras"
kt() { rocprofv3 --kernel-trace --stats -d $O/r03_$1 -o t -- "${@:2}" > $O/r03_$1.log 2>&1; python $R/profiles/summarize_rocprof.py $O/r03_$1/t_results.db > $O/r03_$1_kernel_stats.txt; rm -rf $O/r03_$1 $O/r03_$1.log; }
kt default python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline
kt ode01 $B
kt dae01 $B --workload dae01
kt ode02 $B --workload ode02
kt ode01_euler $B --method euler
kt dae01_euler $B --workload dae01 --method euler
kt ode01_h128 $B --hidden 128
kt dae01_h128 $B --workload dae01 --hidden 128
kt train_ode01 $B --train --steps 5
kt train_dae01 $B --train --workload dae01 --steps 5
kt train_ode01_h128 $B --train --hidden 128 --steps 3 --warmup 1
kt train_ode01_h32 $B --train --hidden 32 --steps 3 --warmup 1
kt train_dae01_h128 $B --train --workload dae01 --hidden 128 --steps 3 --warmup 1
kt train_dae01_h32 $B --train --workload dae01 --hidden 32 --steps 3 --warmup 1
kt train_models python $R/profiles/scripts/train_step_models.py ode02 dae02
pmc() { rocprofv3 --kernel-trace --pmc $2 -d $O/r03_$1_$2 -o p -- "${@:4}" > /dev/null 2>&1; python $R/profiles/summarize_pmc.py $O/r03_$1_$2/p_results.db $3 > $O/r03_$1_$2_pmc.txt; rm -rf $O/r03_$1_$2; }
for c in FETCH_SIZE WRITE_SIZE; do
  pmc ode01 $c integrate_mfma $B
  pmc dae01 $c integrate_mfma $B --workload dae01
  pmc ode02 $c latent_dpp $B --workload ode02
  pmc ode01_euler $c integrate_mfma $B --method euler
  pmc dae01_euler $c integrate_mfma $B --workload dae01 --method euler
done
cd $R
bash profiles/scripts/pmc_sq.sh r03_ode01 integrate_mfma --workload ode01 > /dev/null
bash profiles/scripts/pmc_sq.sh r03_dae01 integrate_mfma --workload dae01 > /dev/null
bash profiles/scripts/pmc_sq.sh r03_ode02 latent_dpp --workload ode02 > /dev/null
bash profiles/scripts/pmc_sq.sh r03_ode01_euler integrate_mfma --workload ode01 --method euler > /dev/null
bash profiles/scripts/pmc_sq.sh r03_dae01_euler integrate_mfma --workload dae01 --method euler > /dev/null
bash profiles/scripts/pmc_sq.sh r03_k4 ode_backward_kernel --train --steps 2 --warmup 1 > /dev/null
bash profiles/scripts/pmc_sq.sh r03_k7 dae_backward_kernel --train --workload dae01 --steps 2 --warmup 1 > /dev/null
bash profiles/scripts/pmc_sq.sh r03_k4f_h128 ode_backward_fused --train --hidden 128 --steps 2 --warmup 1 > /dev/null
bash profiles/scripts/pmc_sq.sh r03_k7f_h128 dae_backward_fused --train --workload dae01 --hidden 128 --steps 2 --warmup 1 > /dev/null
rm -f $O/pmc_r03_*.log
python profiles/scripts/accuracy_report.py > $O/r03_accuracy_report.txt 2>&1
bash profiles/scripts/batch_sweep.sh > $O/r03_batch_sweep.txt 2>&1
python profiles/scripts/train_step_models.py > $O/r03_train_step_models.txt 2>&1; cp $O/train_step_models.json $O/r03_train_step_models.json
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/r03_bench_default_n1.json
python bench.py --gpus 1 --force-dist --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r03_bench_ode01_forcedist_n1.json
python bench.py --steps 5 --train --no-cpu-baseline 2>/dev/null | tail -1 > $O/r03_bench_ode01_train_n1.json
python bench.py --steps 5 --train --workload dae01 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r03_bench_dae01_train_n1.json
for w in ode01 dae01; do for h in 128 32; do python bench.py --steps 5 --warmup 2 --train --workload $w --hidden $h --no-cpu-baseline 2>/dev/null | tail -1 > $O/r03_bench_${w}_h${h}_train_n1.json; done; done
for w in ode01 dae01; do for m in midpoint euler; do python bench.py --steps 5 --warmup 2 --train --workload $w --hidden 128 --method $m --no-cpu-baseline 2>/dev/null | tail -1 > $O/r03_bench_${w}_h128_${m}_train_n1.json; done; done
for w in ode01 dae01; do PSNODE_SAVE_ACTIVATIONS=0 python bench.py --steps 5 --warmup 2 --train --workload $w --hidden 128 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r03_bench_${w}_h128_train_recompute_n1.json; done
python bench.py --steps 5 --hidden 128 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $O/r03_bench_ode01_h128_n1.json
python bench.py --steps 5 --hidden 128 --workload dae01 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r03_bench_dae01_h128_n1.json
ls $O | grep r03_
