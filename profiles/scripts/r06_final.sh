# Round-6 final pass (one gpurun call, one box): the full GPU suite, kernel-trace stats of the default bench run and of the generic integrator K0
# (x_dim 20 ODE, z + v + i = 16 DAE), K0's HBM traffic and SQ breakdown, the default bench line itself.
#   gpurun -- 'bash profiles/scripts/r06_final.sh'   then copy gpurun_out/r06e_* into profiles/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=r06e
B="python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras"
kt() { timeout 600 rocprofv3 --kernel-trace --stats -d $O/${TAG}_$1 -o t -- "${@:2}" > $O/${TAG}_$1.log 2>&1 < /dev/null; timeout 60 python $R/profiles/summarize_rocprof.py $O/${TAG}_$1/t_results.db > $O/${TAG}_$1_kernel_stats.txt; rm -rf $O/${TAG}_$1 $O/${TAG}_$1.log; }
kt default python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline
kt k0_ode_x20 $B --workload ode01_x20
kt k0_dae_zvi16 $B --workload dae01_zvi16
pmc() { timeout 400 rocprofv3 --kernel-trace --pmc $2 -d $O/${TAG}_$1_$2 -o p -- "${@:4}" > /dev/null 2>&1 < /dev/null; timeout 60 python $R/profiles/summarize_pmc.py $O/${TAG}_$1_$2/p_results.db $3 > $O/${TAG}_$1_$2_pmc.txt; rm -rf $O/${TAG}_$1_$2; }
for c in FETCH_SIZE WRITE_SIZE; do
  pmc k0_ode_x20 $c generic_kernel $B --workload ode01_x20
done
cd $R
export GRAFT_REPO_ROOT=$R
timeout 900 bash profiles/scripts/pmc_sq.sh ${TAG}_k0_ode_x20 generic_kernel --workload ode01_x20 > /dev/null 2>&1 < /dev/null
rm -f $O/pmc_${TAG}_*.log
timeout 900 python bench.py --steps 20 --warmup 5 2>/dev/null < /dev/null | tail -1 > $O/${TAG}_bench_default_n1.json
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 < /dev/null | tail -4 > $O/${TAG}_gpu_tests_full.txt
ls $O | grep ${TAG}_
