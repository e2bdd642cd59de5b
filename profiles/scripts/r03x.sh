cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
kt() { rocprofv3 --kernel-trace --stats -d $O/r03x_$1 -o t -- "${@:2}" > $O/r03x_$1.log 2>&1; python $R/profiles/summarize_rocprof.py $O/r03x_$1/t_results.db > $O/r03x_$1_kernel_stats.txt; rm -rf $O/r03x_$1 $O/r03x_$1.log; }
kt e128 $B --train --workload dae01 --hidden 128 --method euler
kt e64 $B --train --workload dae01 --hidden 64 --method euler
kt o128 $B --train --workload ode01 --hidden 128 --method euler
head -7 $O/r03x_e128_kernel_stats.txt | cut -c1-140; head -7 $O/r03x_e64_kernel_stats.txt | cut -c1-140; head -5 $O/r03x_o128_kernel_stats.txt | cut -c1-140
