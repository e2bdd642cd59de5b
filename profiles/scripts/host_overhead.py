"""Host-side cost of one call of the hot path (enqueue only, no device sync inside the loop) + cProfile of the ODE_02 model route."""
import cProfile, pstats, sys, time, torch
sys.path.insert(0, ".")
import bench
from py_psnode_amd import fused

dev = torch.device("cuda")
for wl in sys.argv[1:] or ["ode02", "ode01", "dae01"]:
    w = dict(bench.WORKLOADS[wl])
    p = bench.to_dev(bench.make_problem(w, w["B"], w["T"]), dev)
    f = lambda: bench.run_fused(fused, w, p, "rk4", "auto")
    for _ in range(3): f()
    torch.cuda.synchronize()
    n = 30
    t0 = time.perf_counter()
    for _ in range(n): f()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{wl}: host enqueue {1e3 * (t1 - t0) / n:.3f} ms/call, total {1e3 * (t2 - t0) / n:.3f} ms/call")
    if wl == "ode02":
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(50): f()
        pr.disable()
        torch.cuda.synchronize()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
