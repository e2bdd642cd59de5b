#!/bin/bash
# round 4: K9 two-role agreement test, default bench line with the model-level training steps
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out
timeout 1500 python -m pytest tests/test_gpu_rows_backward.py tests/test_gpu_bench_dist.py -m gpu -q -x -k "two_role or default_bench" 2>&1 | tail -4 | cut -c1-300
timeout 900 python bench.py 2>/dev/null | tail -1 > $O/r04ad_bench_default.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04ad_bench_default.json"))
print("headline ms %.3f frac %.3f" % (d["ms_per_step"], d["roofline"]["frac"]))
for e in d["extra"]:
    print("%-100s ms %7.3f frac %.3f" % (e["workload"][:100], e["ms_per_step"], e["roofline"]["frac"]))
PY
