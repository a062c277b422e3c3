import torch, time
x = torch.empty(1<<29, dtype=torch.float64, device='cuda')  # 4 GiB
y = torch.empty(1<<29, dtype=torch.float64, device='cuda')
def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
gb = x.numel()*8/1e9
ms = t(lambda: x.zero_()); print(f"zero_ (write only)   {gb/ms*1e3/1e3:.2f} TB/s  ({ms:.3f} ms for {gb:.2f} GB)")
ms = t(lambda: x.fill_(1.5)); print(f"fill_ (write only)   {gb/ms*1e3/1e3:.2f} TB/s")
ms = t(lambda: y.copy_(x)); print(f"copy_ (read + write) {2*gb/ms*1e3/1e3:.2f} TB/s total")
ms = t(lambda: x.sum()); print(f"sum (read only)      {gb/ms*1e3/1e3:.2f} TB/s")
