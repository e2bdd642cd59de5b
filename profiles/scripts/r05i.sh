#!/bin/bash
# round 5, call i: K1x with the FAST loop chosen on the device (event table scanned): timings; the forward fuzz with both MFMA integrators
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
python profiles/scripts/fuzz_forward.py 5 150 2>&1 | grep -v amdgpu | tail -12 > $O/r05i_fuzz_forward.txt
{
for r in 1 2; do for k in tile wave; do for m in rk4 euler midpoint; do
  python bench.py --workload ode01 --method $m --kernel $k --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('round $r $k $m kernel_ms %.4f frac %.4f  %s' % (d['roofline']['kernel_ms'], d['roofline']['frac'], d['config']['kernel']))"
done; done; done
} > $O/r05i_tile_vs_wave.txt 2>&1
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_latent_wide.py tests/test_gpu_encoded.py -m gpu -q --tb=line 2>&1 | tail -8 > $O/r05i_pytest.txt
