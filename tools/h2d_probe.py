import time, torch, numpy as np
n = 45_000_000
p = torch.empty(n, dtype=torch.float32, pin_memory=True)
d = torch.empty(n, dtype=torch.float32, device="cuda")
v = torch.from_numpy(p.numpy())
for name, src in (("pinned tensor", p), ("numpy view", v)):
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); x = src.to("cuda", non_blocking=True); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(name, ".to:", f"call {1e3*(t1-t0):.2f} ms, done {1e3*(t2-t0):.2f} ms", src.is_pinned())
        torch.cuda.synchronize()
        t0 = time.perf_counter(); d.copy_(src, non_blocking=True); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(name, "copy_:", f"call {1e3*(t1-t0):.2f} ms, done {1e3*(t2-t0):.2f} ms")
