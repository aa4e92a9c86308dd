import torch, time
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
N = 1 << 30  # 2 GiB bf16
a = torch.empty(N, dtype=torch.bfloat16, device="cuda"); b = torch.empty(N, dtype=torch.bfloat16, device="cuda")
ms = t(lambda: a.zero_()); print("fill   %.3f ms  %.1f GB/s" % (ms, 2 * N / ms / 1e6))
ms = t(lambda: b.copy_(a)); print("copy   %.3f ms  %.1f GB/s (R+W)" % (ms, 4 * N / ms / 1e6))
ms = t(lambda: a.sum()); print("read   %.3f ms  %.1f GB/s" % (ms, 2 * N / ms / 1e6))
