#!/usr/bin/env python3
"""HBM yardstick for the transform kernels: a plain device copy (read + write) and a read-only reduction at their sizes."""
import time, torch
dev = "cuda:0"
for mb in (840, 1260):
    n = mb * 1000 * 1000 // 4
    a = torch.randn(n, device=dev); b = torch.empty_like(a)
    def timed(f, reps=20):
        f(); torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(reps): f()
        torch.cuda.synchronize(); return (time.perf_counter() - t) / reps
    tc = timed(lambda: b.copy_(a)); tr = timed(lambda: a.sum())
    print(f"{mb} MB: copy {tc*1e3:.3f} ms = {2*n*4/tc/1e12:.2f} TB/s (read+write)   sum {tr*1e3:.3f} ms = {n*4/tr/1e12:.2f} TB/s (read)")
