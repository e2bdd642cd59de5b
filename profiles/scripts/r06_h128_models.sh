# Round 6: kernel-trace stats of the direct_encode models' training steps at the scripts' argparse default --hidden 128 (Euler).
#   gpurun -- 'bash profiles/scripts/r06_h128_models.sh [tag]'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-r06}
cat > /tmp/h128_one.py <<PY
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch, bench
dev = torch.device("cuda", 0)
print(bench.model_train_extra_line(sys.argv[1], "euler", dev, steps=5, warmup=2, hidden=128)["ms_per_step"])
PY
for m in ode02 dae02; do
  timeout 400 rocprofv3 --kernel-trace --stats -d $O/${TAG}_h128_$m -o t -- python /tmp/h128_one.py $m > $O/${TAG}_h128_$m.log 2>&1
  timeout 60 python $R/profiles/summarize_rocprof.py $O/${TAG}_h128_$m/t_results.db > $O/${TAG}_train_${m}_h128_euler_kernel_stats.txt
  rm -rf $O/${TAG}_h128_$m $O/${TAG}_h128_$m.log
  head -22 $O/${TAG}_train_${m}_h128_euler_kernel_stats.txt | cut -c1-200
done
