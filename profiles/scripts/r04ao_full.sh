#!/bin/bash
# round 4: full GPU suite + smoke + default bench with the two-role backward kernels (K4f / K7f at <= 4 waves)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 2700 python -m pytest tests -q -x -m gpu > $O/r04ao_full_pytest.txt 2>&1; tail -4 $O/r04ao_full_pytest.txt | cut -c1-300
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py 2>/dev/null | tail -1 > $O/r04ao_bench_default.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04ao_bench_default.json"))
print("headline ms %.3f frac %.3f" % (d["ms_per_step"], d["roofline"]["frac"]))
for e in d["extra"]:
    print("%-95s ms %7.3f frac %.3f%s" % (e["workload"][:95], e["ms_per_step"], e["roofline"]["frac"], ("  each " + str(e["roofline"].get("kernel_ms_each"))) if "TRAIN" in e["workload"] else ""))
PY
