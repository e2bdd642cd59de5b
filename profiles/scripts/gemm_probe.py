import torch, time
dev = torch.device("cuda")
N = 4096 * 125 * 4   # rows of one time chunk (125 steps x 4 stages x 4096 trajectories)
for H in (128, 32):
    d = torch.randn(N, H, device=dev); h = torch.randn(N, H, device=dev)
    for C in (1, 64, 512, 2048):
        def f():
            return torch.bmm(d.view(C, N // C, H).transpose(1, 2), h.view(C, N // C, H)).sum(0)
        f(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): r = f()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        print(f"H={H} C={C}: {dt*1e3:.2f} ms  {2*N*H*H/dt/1e12:.1f} TFLOP/s  {2*N*H*4/dt/1e9:.0f} GB/s")
