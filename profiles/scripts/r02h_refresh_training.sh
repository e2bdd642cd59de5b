cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
kt() { rocprofv3 --kernel-trace --stats -d $O/r02_$1 -o t -- "${@:2}" > $O/r02_$1.log 2>&1; python $R/profiles/summarize_rocprof.py $O/r02_$1/t_results.db > $O/r02_$1_kernel_stats.txt; rm -rf $O/r02_$1 $O/r02_$1.log; }
kt train_ode01 $B --train --steps 5 --warmup 2
kt train_dae01 $B --train --workload dae01 --steps 5 --warmup 2
kt train_ode01_h128 $B --train --hidden 128
kt train_dae01_h128 $B --train --workload dae01 --hidden 128
kt train_models python $R/profiles/scripts/train_step_models.py ode02 dae02
cd $R
for w in ode01 dae01; do for h in 128 32; do python bench.py --steps 5 --warmup 2 --train --workload $w --hidden $h --no-cpu-baseline 2>/dev/null | tail -1 > $O/r02_bench_${w}_h${h}_train_n1.json; done; done
python bench.py --steps 5 --train --no-cpu-baseline 2>/dev/null | tail -1 > $O/r02_bench_ode01_train_n1.json
python bench.py --steps 5 --train --workload dae01 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r02_bench_dae01_train_n1.json
python profiles/scripts/train_step_models.py > $O/r02_train_step_models.txt 2>&1; cp $O/train_step_models.json $O/r02_train_step_models.json
grep -A2 "dominant" $O/r02_train_dae01_h128_kernel_stats.txt | cut -c1-150; grep -v amdgpu $O/r02_train_step_models.txt
