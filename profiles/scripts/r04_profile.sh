# Round-4 profiles (session 3): kernel-trace stats of the default bench line, PMC HBM traffic of the new workloads (DAE_02 both routes,
# hidden 256), SQ breakdown of the streamed-weight kernel, accuracy report, training-step tables.
#   gpurun -- 'bash profiles/scripts/r04_profile.sh'   then copy gpurun_out/r04_* into profiles/
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
B="python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras"
kt() { rocprofv3 --kernel-trace --stats -d $O/r04_$1 -o t -- "${@:2}" > $O/r04_$1.log 2>&1; python $R/profiles/summarize_rocprof.py $O/r04_$1/t_results.db > $O/r04_$1_kernel_stats.txt; rm -rf $O/r04_$1 $O/r04_$1.log; }
kt default python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline
kt ode01 $B
kt ode01_h256 $B --hidden 256
kt ode01_h192 $B --hidden 192
kt dae01_h192 $B --workload dae01 --hidden 192
kt dae02_k3g $B --workload dae02
PSNODE_DAE02_ONE_LAUNCH=0 kt dae02_rows $B --workload dae02
kt train_ode01 $B --train --steps 5
kt train_dae01 $B --train --workload dae01 --steps 5
kt train_ode01_euler $B --train --steps 5 --method euler
kt train_dae01_euler $B --train --workload dae01 --steps 5 --method euler
kt train_ode01_h128 $B --train --hidden 128 --steps 3 --warmup 1
kt train_dae01_h128 $B --train --workload dae01 --hidden 128 --steps 3 --warmup 1
pmc() { rocprofv3 --kernel-trace --pmc $2 -d $O/r04_$1_$2 -o p -- "${@:4}" > /dev/null 2>&1; python $R/profiles/summarize_pmc.py $O/r04_$1_$2/p_results.db $3 > $O/r04_$1_$2_pmc.txt; rm -rf $O/r04_$1_$2; }
for c in FETCH_SIZE WRITE_SIZE; do
  pmc ode01 $c integrate_mfma $B
  pmc ode01_h256 $c integrate_mfma $B --hidden 256
  pmc dae02_k3g $c latent64_model2 $B --workload dae02
  PSNODE_DAE02_ONE_LAUNCH=0 pmc dae02_rows $c "" $B --workload dae02
done
cd $R
bash profiles/scripts/pmc_sq.sh r04_ode01 integrate_mfma --workload ode01 > /dev/null
bash profiles/scripts/pmc_sq.sh r04_ode01_h256 integrate_mfma --workload ode01 --hidden 256 > /dev/null
rm -f $O/pmc_r04_*.log
python profiles/scripts/accuracy_report.py > $O/r04_accuracy_report.txt 2>&1
python profiles/scripts/train_step_models.py > $O/r04_train_step_models.txt 2>&1; cp $O/train_step_models.json $O/r04_train_step_models.json
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/r04_bench_default_n1.json
ls $O | grep r04_
