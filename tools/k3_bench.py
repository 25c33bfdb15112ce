#!/usr/bin/env python3
"""Timing of the pair-grid-network kernels (socmx_mnet_forward_f32 / socmx_mnet_backward_f32) against the library path:
    python tools/k3_bench.py [cfg3|cfg5r|ouq20]      (under rocprofv3 --kernel-trace --stats for the per-kernel split)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "soc-matching_amd"), os.path.join(ROOT, "tests")]
import torch
from socmx import nets, loss as L

CFG = {"cfg3": (10, 200), "cfg5r": (64, 400), "cfg2": (2, 50), "ouq20": (20, 50), "d32": (32, 200)}
name = sys.argv[1] if len(sys.argv) > 1 else "cfg5r"
d, K = CFG[name]
hd = (128, 128)
dev = "cuda:0"
torch.manual_seed(0)
M = nets.SigmoidMLP(dim=d, hdims=hd, gamma=torch.nn.Parameter(torch.tensor([2.0])), scaling_factor=0.1).to(dev)
ts = torch.linspace(0, 1, K + 1).to(dev)
t_vec, s_vec, _, _ = L.pair_times(ts, 1.0, K)
Np = t_vec.shape[0]
c = lambda x: x.detach().to(torch.float32).contiguous()
params = [c(p_) for l in (0, 2, 4) for p_ in (M.sigmoid_layers[l].weight, M.sigmoid_layers[l].bias)]
t_vec, s_vec = c(t_vec), c(s_vec)
gn = torch.randn(Np, d, d, device=dev)
gd = torch.randn(Np, d, d, device=dev)


def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

packed = [None]
def fwd():
    net, dnet, packed[0] = nets.pair_net_forward(d, hd, params, t_vec, s_vec, packed=packed[0])
def bwd():
    nets.pair_net_backward(d, hd, [p_.shape for p_ in params], packed[0], t_vec, s_vec, gn, gd)
def lib():
    M.fused_pair_net = False
    for p_ in M.parameters(): p_.grad = None
    n0, d0 = M.forward_with_ds(t_vec, s_vec, raw=True)
    torch.autograd.backward([n0, d0], [gn, gd])

macs = 2 * hd[0] + hd[0] * hd[1] + hd[1] * d * d
tf, tb = timeit(fwd), timeit(bwd)
fl_f, fl_b = 2 * 2.0 * macs * Np, 2 * 2 * 2.0 * macs * Np + 2 * 2.0 * (2 * hd[0] + hd[0] * hd[1]) * Np
print(f"{name}: d={d} Np={Np}  forward {tf:.3f} ms ({fl_f / tf / 1e9:.1f} TFLOP/s)   backward {tb:.3f} ms ({fl_b / tb / 1e9:.1f} TFLOP/s)")
print(f"   library forward + backward {timeit(lib, 3):.3f} ms")
