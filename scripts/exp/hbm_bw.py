import torch, time
for n in (129*1000*1000, 258*1000*1000):
    a = torch.randn(n, device='cuda'); b = torch.empty_like(a)
    for _ in range(3): b.copy_(a)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): b.copy_(a)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print('copy %d MB: %.3f ms  %.2f TB/s (read+write)' % (n*4/1e6, dt*1e3, 2*n*4/dt/1e12))
    for _ in range(3): a.add_(1.0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): a.add_(1.0)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print('inplace add %d MB: %.3f ms  %.2f TB/s' % (n*4/1e6, dt*1e3, 2*n*4/dt/1e12))
    s = 0
    for _ in range(3): s = a.sum()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): s = a.sum()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print('sum (read only) %d MB: %.3f ms  %.2f TB/s' % (n*4/1e6, dt*1e3, n*4/dt/1e12))
