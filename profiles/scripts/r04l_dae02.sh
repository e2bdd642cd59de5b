#!/bin/bash
# round 4: K3g after deferring its output stores; SQ breakdown
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 900 python -m pytest tests/test_gpu_dae_encoded.py -q -x -m gpu > $O/r04l_pytest.txt 2>&1; tail -2 $O/r04l_pytest.txt | cut -c1-300
B="python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 --workload dae02"
for m in rk4 euler; do
  timeout 600 $B --method $m 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dae02 $m one-launch ms %.3f' % (d['ms_per_step']))"
  PSNODE_DAE02_ONE_LAUNCH=0 timeout 600 $B --method $m 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dae02 $m row kernels + K3c ms %.3f' % (d['ms_per_step']))"
done
bash profiles/scripts/pmc_sq.sh r04l_k3g_rk4 latent64_model --workload dae02 > /dev/null 2>&1
bash profiles/scripts/pmc_sq.sh r04l_k3g_euler latent64_model --workload dae02 --method euler > /dev/null 2>&1
cat $O/r04l_k3g_rk4_pmc_sq.txt | head -40
