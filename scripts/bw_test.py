import torch, time
dev = torch.device("cuda:0")
n = 512 * 1024 * 1024  # 4 GiB of f64
a = torch.empty(n, dtype=torch.float64, device=dev)
b = torch.ones(n, dtype=torch.float64, device=dev)
def t(f, reps=10):
    f(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
ms = t(lambda: a.copy_(b)); print(f"copy 4GiB->4GiB: {ms:.3f} ms  {2*n*8/ms/1e6:.0f} GB/s (r+w)")
ms = t(lambda: a.fill_(2.0)); print(f"fill 4GiB: {ms:.3f} ms  {n*8/ms/1e6:.0f} GB/s (w)")
ms = t(lambda: b.sum()); print(f"sum 4GiB: {ms:.3f} ms  {n*8/ms/1e6:.0f} GB/s (r)")
