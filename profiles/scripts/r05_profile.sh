# Round-5 profiles: kernel-trace stats of the default bench line and of the headline alone, PMC HBM traffic of K1x / K2x (RK4, Euler), SQ breakdown,
# training-step kernel stats, accuracy reports, the default bench line itself.
#   gpurun -- 'bash profiles/scripts/r05_profile.sh [tag]'   then copy gpurun_out/<tag>_* into profiles/
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-r05}
B="python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras"
kt() { rocprofv3 --kernel-trace --stats -d $O/${TAG}_$1 -o t -- "${@:2}" > $O/${TAG}_$1.log 2>&1; python $R/profiles/summarize_rocprof.py $O/${TAG}_$1/t_results.db > $O/${TAG}_$1_kernel_stats.txt; rm -rf $O/${TAG}_$1 $O/${TAG}_$1.log; }
kt default python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline
kt headline python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras
kt ode01_euler $B --method euler
kt dae01 $B --workload dae01
kt dae01_euler $B --workload dae01 --method euler
kt ode02 $B --workload ode02 --warmup 10
kt train_ode01 $B --train
kt train_ode01_euler $B --train --method euler
kt train_dae01 $B --train --workload dae01
pmc() { rocprofv3 --kernel-trace --pmc $2 -d $O/${TAG}_$1_$2 -o p -- "${@:4}" > /dev/null 2>&1; python $R/profiles/summarize_pmc.py $O/${TAG}_$1_$2/p_results.db $3 > $O/${TAG}_$1_$2_pmc.txt; rm -rf $O/${TAG}_$1_$2; }
for c in FETCH_SIZE WRITE_SIZE; do
  pmc ode01 $c integrate_x $B
  pmc ode01_euler $c integrate_x $B --method euler
  pmc dae01 $c integrate_xd $B --workload dae01
  pmc dae01_euler $c integrate_xd $B --workload dae01 --method euler
  pmc ode02 $c latent_dpp $B --workload ode02 --warmup 10
done
cd $R
export GRAFT_REPO_ROOT=$R
bash profiles/scripts/pmc_sq.sh ${TAG}_ode01 integrate_x --workload ode01 > /dev/null 2>&1
bash profiles/scripts/pmc_sq.sh ${TAG}_dae01 integrate_xd --workload dae01 > /dev/null 2>&1
rm -f $O/pmc_${TAG}_*.log
python profiles/scripts/accuracy_report.py 2>&1 | grep -v amdgpu > $O/${TAG}_accuracy_report.txt
python profiles/scripts/grad_accuracy_report.py 2>&1 | grep -v amdgpu > $O/${TAG}_grad_accuracy_report.txt
python profiles/scripts/glue_trace_model.py ode02 rk4 2>&1 | grep -v "Warning\|warn\|amdgpu" > $O/${TAG}_glue_ode02.txt
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/${TAG}_bench_default_n1.json
ls $O | grep ${TAG}_
