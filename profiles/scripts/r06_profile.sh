# Round-6 profiles: kernel-trace stats of the default bench line and of the ODE_01 training steps (K1x SAVE + K6 + K4x), PMC HBM traffic and the SQ
# issue / wait breakdown of K4x, the default bench line itself.
#   gpurun -- 'bash profiles/scripts/r06_profile.sh [tag]'   then copy gpurun_out/<tag>_* into profiles/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-r06}
B="python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras"
kt() { timeout 400 rocprofv3 --kernel-trace --stats -d $O/${TAG}_$1 -o t -- "${@:2}" > $O/${TAG}_$1.log 2>&1; timeout 60 python $R/profiles/summarize_rocprof.py $O/${TAG}_$1/t_results.db > $O/${TAG}_$1_kernel_stats.txt; rm -rf $O/${TAG}_$1 $O/${TAG}_$1.log; }
kt default python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline
kt headline python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras
kt train_ode01 $B --train
kt train_ode01_euler $B --train --method euler
kt train_ode01_tile $B --train --kernel tile
kt train_dae01 $B --train --workload dae01
pmc() { timeout 400 rocprofv3 --kernel-trace --pmc $2 -d $O/${TAG}_$1_$2 -o p -- "${@:4}" > /dev/null 2>&1; timeout 60 python $R/profiles/summarize_pmc.py $O/${TAG}_$1_$2/p_results.db $3 > $O/${TAG}_$1_$2_pmc.txt; rm -rf $O/${TAG}_$1_$2; }
for c in FETCH_SIZE WRITE_SIZE; do
  pmc ode01 $c integrate_x $B
  pmc train_ode01_bwd $c ode_backward_x $B --train
  pmc train_ode01_fwd $c integrate_x $B --train
  pmc train_ode01_euler_bwd $c ode_backward_x $B --train --method euler
done
cd $R
export GRAFT_REPO_ROOT=$R
timeout 900 bash profiles/scripts/pmc_sq.sh ${TAG}_k4x_rk4 ode_backward_x --train > /dev/null 2>&1
timeout 900 bash profiles/scripts/pmc_sq.sh ${TAG}_k4x_euler ode_backward_x --train --method euler > /dev/null 2>&1
rm -f $O/pmc_${TAG}_*.log
timeout 300 python profiles/scripts/r06_k4x_time.py rk4,euler,midpoint 2>&1 | grep -v amdgpu > $O/${TAG}_k4x_vs_k4f_backward_alone.txt
timeout 600 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/${TAG}_bench_default_n1.json
ls $O | grep ${TAG}_
