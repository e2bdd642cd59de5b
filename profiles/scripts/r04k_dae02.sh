#!/bin/bash
# round 4: DAE_02 model forward in one launch (K3g) -- parity tests, timing against the row-kernel + K3c route
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 1500 python -m pytest tests/test_gpu_dae_encoded.py tests/test_gpu_parity.py -q -x -m gpu -k "encoded or streamed" > $O/r04k_pytest.txt 2>&1; tail -4 $O/r04k_pytest.txt | cut -c1-300
B="python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 --workload dae02"
for m in rk4 euler; do
  timeout 600 $B --method $m 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dae02 $m one-launch ms %.3f frac %.3f' % (d['ms_per_step'], d['roofline']['frac']))"
  PSNODE_DAE02_ONE_LAUNCH=0 timeout 600 $B --method $m 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dae02 $m row kernels + K3c ms %.3f frac %.3f' % (d['ms_per_step'], d['roofline']['frac']))"
done
