import torch, time
x = torch.empty(24, 2048, 2048, device="cuda")
for fn, name in ((lambda: x.fill_(1.0), "fill_ 403 MB"), (lambda: x.zero_(), "zero_ (memset)"), (lambda: torch.add(x, 1.0, out=x), "x += 1 (read + write)")):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): fn()
    e.record(); torch.cuda.synchronize()
    us = 1e3 * s.elapsed_time(e) / 20
    print("%-24s %7.1f us  %5.2f TB/s written" % (name, us, x.numel() * 4 / us / 1e6))
