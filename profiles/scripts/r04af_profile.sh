# Round-4 FINAL profiles, two-role backward kernels (K4f / K7f at <= 4 waves, K9): kernel-trace stats of the default line and of every training
# step, SQ breakdown of the three two-role kernels, HBM traffic (FETCH / WRITE) of the hidden-64 training backward in both forms.
#   gpurun -- 'bash profiles/scripts/r04af_profile.sh'   then copy gpurun_out/r04af_* into profiles/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
B="python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras"
kt() { rocprofv3 --kernel-trace --stats -d $O/r04af_$1 -o t -- "${@:2}" > $O/r04af_$1.log 2>&1; python $R/profiles/summarize_rocprof.py $O/r04af_$1/t_results.db > $O/r04af_$1_kernel_stats.txt; rm -rf $O/r04af_$1 $O/r04af_$1.log; }
kt default python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline
kt train_ode01 $B --train --steps 5
kt train_dae01 $B --train --workload dae01 --steps 5
kt train_ode01_euler $B --train --steps 5 --method euler
kt train_dae01_euler $B --train --workload dae01 --steps 5 --method euler
kt train_ode01_h32 $B --train --hidden 32 --steps 5
kt train_dae01_h32 $B --train --workload dae01 --hidden 32 --steps 5
kt train_dae02 python $R/profiles/scripts/train_step_models.py dae02
pmc() { rocprofv3 --kernel-trace --pmc $2 -d $O/r04af_$1_$2 -o p -- "${@:4}" > /dev/null 2>&1; python $R/profiles/summarize_pmc.py $O/r04af_$1_$2/p_results.db $3 > $O/r04af_$1_$2_pmc.txt; rm -rf $O/r04af_$1_$2; }
for c in FETCH_SIZE WRITE_SIZE; do
  pmc k4f_roles $c ode_backward_fused $B --train --steps 2 --warmup 1
  PSNODE_K4F_NO_ROLES=1 pmc k4f_onerole $c ode_backward_fused $B --train --steps 2 --warmup 1
  pmc k7f_roles $c dae_backward_fused $B --train --workload dae01 --steps 2 --warmup 1
  PSNODE_K7F_NO_ROLES=1 pmc k7f_onerole $c dae_backward_fused $B --train --workload dae01 --steps 2 --warmup 1
done
cd $R
bash profiles/scripts/pmc_sq.sh r04af_k4f_roles_rk4 ode_backward_fused --train --steps 2 --warmup 1 > /dev/null
bash profiles/scripts/pmc_sq.sh r04af_k4f_roles_euler ode_backward_fused --train --method euler --steps 2 --warmup 1 > /dev/null
bash profiles/scripts/pmc_sq.sh r04af_k7f_roles_rk4 dae_backward_fused --train --workload dae01 --steps 2 --warmup 1 > /dev/null
bash profiles/scripts/pmc_sq.sh r04af_k7f_roles_euler dae_backward_fused --train --workload dae01 --method euler --steps 2 --warmup 1 > /dev/null
bash profiles/scripts/r04aa_dae02_train.sh r04af_k9 > /dev/null 2>&1
rm -f $O/pmc_r04af_*.log $O/r04af_k9_train.log
python profiles/scripts/train_step_models.py > $O/r04af_train_step_models.txt 2>&1; cp $O/train_step_models.json $O/r04af_train_step_models.json
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/r04af_bench_default_n1.json
ls $O | grep r04af_
