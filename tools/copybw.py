"""Achievable HBM copy bandwidth on this box (context for the roofline fractions): torch device-to-device copy and a
read-only reduction at the sizes of the headline batch."""
import torch
x = torch.empty(1_000_000 * 32, dtype=torch.float64, device="cuda")
x.normal_()
y = torch.empty_like(x)
for name, fn, nbytes in (("copy 256MB->256MB", lambda: y.copy_(x), 2 * x.numel() * 8),
                         ("read-only sum 256MB", lambda: x.sum(), x.numel() * 8),
                         ("fill 256MB", lambda: y.fill_(1.0), x.numel() * 8)):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    print(f"{name}: {ms*1e3:.1f} us  {nbytes/ms/1e6:.0f} GB/s")
