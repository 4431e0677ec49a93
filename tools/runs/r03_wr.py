import torch, time
x = torch.empty(256 * 1024 * 1024, device="cuda")      # 1 GiB
y = torch.empty_like(x)
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
g = 2 ** 30 / 1e9
print("fill 1 GiB: %.3f ms  %.2f TB/s" % (t(lambda: x.fill_(1.0)), g / t(lambda: x.fill_(1.0))))
print("copy 1 GiB: %.3f ms  %.2f TB/s (r+w)" % (t(lambda: y.copy_(x)), 2 * g / t(lambda: y.copy_(x))))
print("sum  1 GiB: %.3f ms  %.2f TB/s" % (t(lambda: x.sum()), g / t(lambda: x.sum())))
