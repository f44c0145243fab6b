"""What this box's memory system sustains for the partition's traffic (160 MB read, 80 MB written, repeatedly on the same
buffers -- i.e. out of the 256 MB Infinity Cache as far as it holds them): plain torch kernels as the yardstick."""
import torch
torch.cuda.set_device(0)
def t(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for mb in (160, 800):
    n = mb * 1000 * 1000 // 4
    x = torch.rand(n, device="cuda"); y = torch.empty(n // 2, device="cuda"); z = torch.empty(n, device="cuda")
    us = t(lambda: torch.sum(x))
    print("read %d MB (sum): %.1f us = %.2f TB/s" % (mb, us, mb / us))
    us = t(lambda: z.copy_(x))
    print("copy %d MB -> %d MB: %.1f us = %.2f TB/s moved" % (mb, mb, us, 2 * mb / us))
    us = t(lambda: torch.add(x[: n // 2], x[n // 2:], out=y))
    print("read %d MB, write %d MB (add halves): %.1f us = %.2f TB/s moved" % (mb, mb // 2, us, 1.5 * mb / us))
    us = t(lambda: y.fill_(1.0))
    print("write %d MB (fill): %.1f us = %.2f TB/s" % (mb // 2, us, mb / 2 / us))
