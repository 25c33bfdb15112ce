#!/usr/bin/env python3
"""HIP-event timings of the socmx loss kernels at one config (no profiler): python tools/kernel_times.py cfg5r"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "soc-matching_amd")]
import torch
from socmx import loss as L, _lib
from socmx.problems import Problem

CFG = {"cfg3": (10, 200, 128), "cfg5r": (64, 400, 512), "cfg4r": (10, 200, 1024), "cfg2": (2, 50, 128)}
d, K, B = CFG[sys.argv[1] if len(sys.argv) > 1 else "cfg3"]
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
rn = lambda *s: torch.randn(*s, generator=g).to(dev)
pb = Problem(_lib.OU_LINEAR, d, (torch.eye(d) + 0.1 * torch.randn(d, d, generator=g)).to(dev),
             A=(-torch.eye(d)).to(dev), omega=torch.ones(d).to(dev))
ts = torch.linspace(0, 1, K + 1).to(dev)
states, noises, controls = rn(K + 1, B, d), rn(K, B, d), rn(K, B, d)
Np = (K + 1) * (K + 2) // 2
net, dnet = (0.1 * rn(Np, d, d)).requires_grad_(True), (0.1 * rn(Np, d, d)).requires_grad_(True)
gam = torch.tensor(2.0, device=dev, requires_grad=True)
nV = rn(K + 1, B, d).requires_grad_(True)
w = torch.rand(B, generator=g).to(dev) + 0.5
t_vec, s_vec, _, _ = L.pair_times(ts, 1.0, K)
delta = (s_vec - t_vec).contiguous()


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

ops = L.socm_operands_hip(pb, ts, 1.0, states, noises, controls)
print(f"d={d} K={K} B={B} Np={Np}")
print(f"prep        {timeit(lambda: L.socm_operands_hip(pb, ts, 1.0, states, noises, controls)):9.3f} ms")
fl = 4.0 * B * d * d * Np
out = None
def fwd():
    global out
    out = L._TargetResidualNetHip.apply(net, dnet, gam, nV, w, delta, ops, pb, K, 1.0 / ((K + 1) * B))
t = timeit(fwd)
print(f"fwd(+resid) {t:9.3f} ms   {fl / t / 1e9:7.1f} TFLOP/s of the contraction")
def fb():
    fwd(); out.backward()
t2 = timeit(fb)
print(f"fwd+bwd     {t2:9.3f} ms   bwd ~ {t2 - t:9.3f} ms  {fl / max(t2 - t, 1e-6) / 1e9:7.1f} TFLOP/s")
