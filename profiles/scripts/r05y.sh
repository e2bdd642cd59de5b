#!/bin/bash
# round 5, call y: K3f two-role form (row-wise MLPs on the partner wave, MFMA tiles, LDS rings): the ODE_02 / latent tests, config 3 forward time
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
python -m pytest tests/test_gpu_encoded.py tests/test_gpu_parity.py tests/test_grad_goldens.py tests/test_gpu_example.py -m gpu -q --tb=short -k "ode02 or encoded or g4 or model or example or latent or event" 2>&1 | tail -12 > $O/r05y_pytest.txt
for r in 1 2; do for m in rk4 euler midpoint; do
  python bench.py --workload ode02 --method $m --steps 20 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('round $r ode02 $m ms %.4f kernel_ms %.4f frac %.4f %s err %s' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['config']['kernel'], d['config'].get('rel_err_vs_oracle')))"
done; done > $O/r05y_ode02.txt 2>&1
