#!/usr/bin/env python3
"""Host-to-device copy rates on this box: pageable vs pinned source, one stream vs four (what a scene upload can hope for)."""
import time, torch
n = 123 * 1024 * 1024
src = torch.empty(n, dtype=torch.uint8).fill_(1)
pin = torch.empty(n, dtype=torch.uint8).pin_memory().fill_(1)
dst = torch.empty(n, dtype=torch.uint8, device="cuda")
def t(f, reps=5):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best
a = t(lambda: dst.copy_(src))
b = t(lambda: dst.copy_(pin, non_blocking=True))
streams = [torch.cuda.Stream() for _ in range(4)]
q = n // 4
def four():
    for i, s in enumerate(streams):
        with torch.cuda.stream(s):
            dst[i * q:(i + 1) * q].copy_(pin[i * q:(i + 1) * q], non_blocking=True)
c = t(four)
tmp = torch.empty(n, dtype=torch.uint8)
d = t(lambda: tmp.copy_(src))
print(f"{n/1e6:.0f} MB: pageable {a*1e3:.2f} ms ({n/a/1e9:.1f} GB/s), pinned {b*1e3:.2f} ms ({n/b/1e9:.1f} GB/s), pinned on 4 streams {c*1e3:.2f} ms ({n/c/1e9:.1f} GB/s), host memcpy 1 thread {d*1e3:.2f} ms ({n/d/1e9:.1f} GB/s)")
