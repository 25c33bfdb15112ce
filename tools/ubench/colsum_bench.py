#!/usr/bin/env python3
"""socmx_colsum_f32 against torch's sum(0) on the bias-gradient shapes of cfg3."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "soc-matching_amd")]
import torch
from socmx import nets

def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

for R, C in [(25728, 256), (25728, 128), (25728, 64), (25728, 10), (20301, 128), (20301, 100)]:
    x = torch.randn(R, C, device="cuda")
    t_hip = timeit(lambda: nets._colsum(x))
    t_torch = timeit(lambda: x.sum(0))
    print(f"R={R} C={C}: hip {t_hip:.1f} us ({R*C*4/t_hip/1e6:.2f} TB/s)  torch {t_torch:.1f} us")
