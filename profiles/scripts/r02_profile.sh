# Round-2 profiles: rocprofv3 kernel-trace stats for every bench line quoted in DESIGN.md, PMC HBM traffic (separate FETCH / WRITE
# passes) for the three single-GPU configs, SQ breakdowns, accuracy report, batch sweep, training-step table.
#   gpurun -- 'bash profiles/scripts/r02_profile.sh'   then copy gpurun_out/r02_* into profiles/
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
B="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline"
kt() { rocprofv3 --kernel-trace --stats -d $O/r02_$1 -o t -- "${@:2}" > $O/r02_$1.log 2>&1; python $R/profiles/summarize_rocprof.py $O/r02_$1/t_results.db > $O/r02_$1_kernel_stats.txt; rm -rf $O/r02_$1 $O/r02_$1.log; }
kt ode01 $B
kt dae01 $B --workload dae01
kt ode02 $B --workload ode02
kt ode02_latent16 $B --workload ode02_latent16
kt ode01_h128 $B --hidden 128
kt dae01_h128 $B --workload dae01 --hidden 128
kt train_ode01 $B --train --steps 5
kt train_dae01 $B --train --workload dae01 --steps 5
kt train_models python $R/profiles/scripts/train_step_models.py ode02 dae02
kt train_ode01_h128 $B --train --hidden 128 --steps 3 --warmup 1
kt train_dae01_h128 $B --train --workload dae01 --hidden 128 --steps 3 --warmup 1
pmc() { rocprofv3 --kernel-trace --pmc $2 -d $O/r02_$1_$2 -o p -- "${@:4}" > /dev/null 2>&1; python $R/profiles/summarize_pmc.py $O/r02_$1_$2/p_results.db $3 > $O/r02_$1_$2_pmc.txt; rm -rf $O/r02_$1_$2; }
for c in FETCH_SIZE WRITE_SIZE; do
  pmc ode01 $c integrate_mfma $B
  pmc dae01 $c integrate_mfma $B --workload dae01
  pmc ode02 $c latent_dpp $B --workload ode02
done
cd $R
bash profiles/scripts/pmc_sq.sh r02_ode01 integrate_mfma --workload ode01 > /dev/null
bash profiles/scripts/pmc_sq.sh r02_dae01 integrate_mfma --workload dae01 > /dev/null
bash profiles/scripts/pmc_sq.sh r02_ode02 latent_dpp --workload ode02 > /dev/null
rm -f $O/pmc_r02_*.log
python profiles/scripts/accuracy_report.py > $O/r02_accuracy_report.txt 2>&1
bash profiles/scripts/batch_sweep.sh > $O/r02_batch_sweep.txt 2>&1
python profiles/scripts/train_step_models.py > $O/r02_train_step_models.txt 2>&1; cp $O/train_step_models.json $O/r02_train_step_models.json
python bench.py --steps 20 --warmup 3 2>/dev/null | tail -1 > $O/r02_bench_ode01_n1.json
python bench.py --steps 10 --workload dae01 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r02_bench_dae01_n1.json
python bench.py --steps 50 --warmup 10 --workload ode02 2>/dev/null | tail -1 > $O/r02_bench_ode02_n1.json     # (sub-ms calls: the first ~10 carry a one-time 1.5 ms transient)
python bench.py --steps 50 --warmup 10 --workload ode02_latent16 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r02_bench_ode02_latent16_n1.json
python bench.py --steps 5 --train --no-cpu-baseline 2>/dev/null | tail -1 > $O/r02_bench_ode01_train_n1.json
python bench.py --steps 5 --train --workload dae01 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r02_bench_dae01_train_n1.json
for w in ode01 dae01; do for h in 128 32; do python bench.py --steps 5 --warmup 2 --train --workload $w --hidden $h --no-cpu-baseline 2>/dev/null | tail -1 > $O/r02_bench_${w}_h${h}_train_n1.json; done; done
python bench.py --steps 5 --hidden 128 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r02_bench_ode01_h128_n1.json
python bench.py --steps 5 --hidden 128 --workload dae01 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r02_bench_dae01_h128_n1.json
ls $O | grep r02_
