"""Link speed probe (debug aid): pinned H2D / D2H copies of 64 MB, alone and both at once."""
import time
import torch
n = 64 << 20
h = torch.empty(n, dtype=torch.uint8).pin_memory()
h2 = torch.empty(n, dtype=torch.uint8).pin_memory()
d = torch.empty(n, dtype=torch.uint8, device="cuda:0")
d2 = torch.empty(n, dtype=torch.uint8, device="cuda:0")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
def h2d():
    with torch.cuda.stream(s1):
        d.copy_(h, non_blocking=True)
def d2h():
    with torch.cuda.stream(s2):
        h2.copy_(d2, non_blocking=True)
def both():
    h2d(); d2h()
def two_h2d():
    with torch.cuda.stream(s1):
        d.copy_(h, non_blocking=True)
    with torch.cuda.stream(s2):
        d2.copy_(h2, non_blocking=True)
print("H2D GB/s", n / t(h2d) / 1e9)
print("D2H GB/s", n / t(d2h) / 1e9)
print("both GB/s each", n / t(both) / 1e9)
print("two H2D streams GB/s total", 2 * n / t(two_h2d) / 1e9)
