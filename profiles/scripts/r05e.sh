#!/bin/bash
# round 5, call e: tracebacks of the backward-suite failures after the prune; K1x with the store behind the prefetch (vmcnt(1))
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
python -m pytest "tests/test_gpu_backward.py::test_dae_wide_backward_edge_sizes_and_chunks" "tests/test_gpu_backward.py::test_dae_mfma_backward_edge_sizes" "tests/test_gpu_backward.py::test_wide_backwards_at_zero_padded_hidden_widths" "tests/test_gpu_parity.py::test_order_of_accuracy_full_batch" -m gpu -q -x --tb=short 2>&1 | tail -60 > $O/r05e_tb1.txt
python -m pytest tests/test_gpu_backward.py tests/test_grad_goldens.py tests/test_tf_goldens.py tests/test_gpu_rows_backward.py tests/test_gpu_dae_encoded.py -m gpu -q --tb=line 2>&1 | grep -v "^tests.*PASSED" | tail -60 > $O/r05e_pytest_bwd.txt
{
for r in 1 2; do for k in tile wave; do for m in rk4 euler; do
  python bench.py --workload ode01 --method $m --kernel $k --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('round $r $k $m kernel_ms %.4f frac %.4f' % (d['roofline']['kernel_ms'], d['roofline']['frac']))"
done; done; done
} > $O/r05e_tile_vs_wave.txt 2>&1
python -m pytest tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -4 > $O/r05e_pytest_parity.txt
