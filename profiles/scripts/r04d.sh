#!/bin/bash
# round 4: after the fixes (branch-free loads, no uniform branch behind MFMAs, LDS-DMA wait states): defects, suite, bench
mkdir -p gpurun_out/r04d
O=gpurun_out/r04d
DEFECT_VERBOSE=1 timeout 300 python profiles/scripts/r04_defects.py a1 3 > $O/a1.txt 2>&1
timeout 300 python profiles/scripts/r04_defects.py a 3 > $O/a.txt 2>&1
timeout 600 python profiles/scripts/r04_defects.py b 2 > $O/b.txt 2>&1
grep "TOTAL\|BAD" $O/a1.txt $O/a.txt $O/b.txt | head -20
timeout 2400 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt
tail -4 $O/pytest_gpu.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04d/bench_default.json").read().strip().split("\n")[-1])
print("headline", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["config"].get("rel_err_vs_oracle"))
for e in d.get("extra", []):
    print(" ", e["workload"][:70], "ms", round(e["ms_per_step"], 3), "frac", round(e["roofline"]["frac"], 4), e["roofline"].get("kernel_ms_by_family"), e.get("saved_bytes"), e.get("host_enqueue_ms"))
print("cpu", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("gpu_vs_oracle"))
PY
