#!/bin/bash
# K1/K2 at the scripts' --hidden widths (32, 64, 128): python bench.py lines condensed to one row each.
mkdir -p gpurun_out
for wl in ode01 dae01; do
  for h in 32 64 128; do
    python bench.py --workload $wl --no-cpu-baseline --hidden $h --steps 10 "$@" 2>/dev/null | tail -1 > gpurun_out/_hw.json
    python - "$wl" "$h" <<'PY'
import json, sys
d = json.load(open("gpurun_out/_hw.json"))
print(f"{sys.argv[1]} H={sys.argv[2]:>3}: {d['ms_per_step']:.3f} ms  {d['value']:.4g} state-steps/s  "
      f"{d['roofline']['achieved']:.1f} TFLOP/s = {d['roofline']['frac']:.3f} of fp32 peak  kernel={d['config']['kernel']}")
PY
  done
done
