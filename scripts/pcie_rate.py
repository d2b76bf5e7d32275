"""Development aid: pinned-memory H2D / D2H rates of this box (the ceiling of the host-buffer API)."""
import time, torch
n = 1 << 30
h = torch.empty(n, dtype=torch.uint8).pin_memory()
d = torch.empty(n, dtype=torch.uint8, device="cuda:0")
for name, f in (("H2D", lambda: d.copy_(h, non_blocking=True)), ("D2H", lambda: h.copy_(d, non_blocking=True))):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): f()
    torch.cuda.synchronize()
    print(name, "%.1f GB/s" % (3 * n / (time.perf_counter() - t0) / 1e9))
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
h2 = torch.empty(n, dtype=torch.uint8).pin_memory(); d2 = torch.empty(n, dtype=torch.uint8, device="cuda:0")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3):
    with torch.cuda.stream(s1): d.copy_(h, non_blocking=True)
    with torch.cuda.stream(s2): h2.copy_(d2, non_blocking=True)
torch.cuda.synchronize()
print("both directions at once: %.1f GB/s each" % (3 * n / (time.perf_counter() - t0) / 1e9))
