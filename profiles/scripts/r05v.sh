#!/bin/bash
# round 5, call v: K2x with the AE inputs two steps ahead and ONE wait per step (vmcnt(2): loads only)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -k "dae" 2>&1 | tail -25 > $O/r05v_pytest_dae.txt
python profiles/scripts/fuzz_forward.py 5 150 2>&1 | grep -v amdgpu | tail -12 > $O/r05v_fuzz_forward.txt
{
for r in 1 2; do for k in tile wave; do for m in rk4 euler midpoint; do
  python bench.py --workload dae01 --method $m --kernel $k --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('round $r $k $m kernel_ms %.4f frac %.4f  %s' % (d['roofline']['kernel_ms'], d['roofline']['frac'], d['config']['kernel']))"
done; done; done
} > $O/r05v_dae_tile_vs_wave.txt 2>&1
