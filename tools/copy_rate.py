import torch, time
n = 1 << 26
x = torch.randint(0, 1 << 30, (n,), dtype=torch.int32, device="cuda")
y = torch.empty_like(x)
for name, fn in (("copy_", lambda: y.copy_(x)), ("add_scalar", lambda: torch.add(x, 1, out=y)), ("fill", lambda: y.fill_(7))):
    for _ in range(50): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 200
    print("%-12s %.4f ms per 2^26 int32" % (name, ms))
