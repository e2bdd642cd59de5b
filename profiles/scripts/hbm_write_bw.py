import torch, time
dev = torch.device("cuda", 0)
for gb in (1, 4, 12):
    n = gb * (1 << 30) // 4
    a = torch.empty(n, dtype=torch.float32, device=dev)
    for _ in range(3): a.fill_(1.0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): a.fill_(2.0)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"fill {gb} GB: {ms:.3f} ms = {gb * 1.073741824 / ms:.2f} TB/s")
    b = torch.empty_like(a) if gb <= 4 else None
    if b is not None:
        for _ in range(2): b.copy_(a)
        torch.cuda.synchronize(); e0.record()
        for _ in range(5): b.copy_(a)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"copy {gb} GB: {ms:.3f} ms = read {gb * 1.0737 / ms:.2f} + write {gb * 1.0737 / ms:.2f} TB/s")
    del a, b
