"""debug: achievable HBM rates of plain torch kernels (write-only, read-only, copy) on this GPU"""
import torch, time
x = torch.empty(512 * 1024 * 1024, dtype=torch.float32, device="cuda")   # 2 GiB
y = torch.empty_like(x)
def t(f, n=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
gb = x.numel() * 4 / 1e9
print("fill   (write)      %.2f TB/s" % (gb / t(lambda: x.fill_(1.0)) / 1e3))
print("sum    (read)       %.2f TB/s" % (gb / t(lambda: x.sum()) / 1e3))
print("copy   (read+write) %.2f TB/s total" % (2 * gb / t(lambda: y.copy_(x)) / 1e3))
print("add3   (2r+1w)      %.2f TB/s total" % (3 * gb / t(lambda: torch.add(x, y, out=y)) / 1e3))
z = torch.empty(2, x.numel() // 2, dtype=torch.float32, device="cuda")
print("1r+2w  (sincos-like) %.2f TB/s total" % (1.5 * gb / t(lambda: torch.stack((x[: x.numel() // 2], x[: x.numel() // 2]), out=z)) / 1e3))
