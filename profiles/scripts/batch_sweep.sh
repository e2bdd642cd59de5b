# K1 / K2 throughput vs trajectories per GPU and vs method (one MI355X): the one-tile-per-CU regime (B=4096) against 2+ tiles per CU.
for wl in ode01 dae01; do
  for b in 4096 8192 16384 32768; do
    python bench.py --workload $wl --batch $b --no-cpu-baseline --no-extras --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl rk4 B=%6d  %.3f ms  %.4g state-steps/s  %.3f of fp32 peak' % ($b, d['ms_per_step'], d['value'], d['roofline']['frac']))"
  done
  for m in euler midpoint; do
    python bench.py --workload $wl --method $m --no-cpu-baseline --no-extras --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl $m B=  4096  %.3f ms  %.4g state-steps/s  %.3f of fp32 peak' % (d['ms_per_step'], d['value'], d['roofline']['frac']))"
  done
done
