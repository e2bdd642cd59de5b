#!/bin/bash
# round 5, call o: the whole GPU suite on K2x (AE weights in AccVGPRs), default bench lines of dae01
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
python -m pytest tests/ -m gpu -q --tb=short 2>&1 | tail -25 > $O/r05o_pytest_gpu.txt
python bench.py --workload dae01 --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/r05o_bench_dae01.json
