#!/usr/bin/env python3
"""Timing of the fused control-network backward (socmx_unet_backward_f32) against library autograd:
    python tools/k2_bench.py [cfg3|cfg5r|cfg2|ouq20]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "soc-matching_amd"), os.path.join(ROOT, "tests")]
import torch
from socmx import nets

CFG = {"cfg3": (10, 200, 128), "cfg5r": (64, 400, 512), "cfg4r": (10, 200, 1024), "cfg2": (2, 50, 128), "ouq20": (20, 50, 128)}
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
d, K, B = CFG[name]
dev = "cuda:0"
torch.manual_seed(0)
net = nets.FullyConnectedUNet(d, [256, 128, 64]).to(dev)
N = (K + 1) * B
x = torch.randn(N, d, device=dev)
ts = torch.linspace(0, 1, K + 1, device=dev)
gout = torch.randn(N, d, device=dev)
tx = torch.cat([ts.reshape(-1, 1, 1).expand(K + 1, B, 1), x.reshape(K + 1, B, d)], -1).reshape(-1, d + 1)


def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def lib():
    for p in net.parameters(): p.grad = None
    net(tx).backward(gout)

macs = sum(p.numel() for n_, p in net.named_parameters() if n_.endswith("weight"))
t_hip = timeit(lambda: nets.unet_backward_hip(net, x, ts, B, gout))
t_lib = timeit(lib)
fl = 6.0 * macs * N          # forward recompute + activation gradients + weight gradients
print(f"{name}: N={N} rows  HIP backward {t_hip:.3f} ms ({fl / t_hip / 1e9:.1f} TFLOP/s of 3x2xMACs)   library fwd+bwd {t_lib:.3f} ms")
