#!/usr/bin/env python3
"""Kernel A of the control-network backward against the number of 16-row tiles (dispatch rounds of 512 workgroups):
rocprofv3 --kernel-trace --stats -- python3 tools/k2_tail.py   (reads nothing; prints the row counts it launched)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "soc-matching_amd"), os.path.join(ROOT, "tests")]
import torch
from socmx import nets
dev = "cuda:0"
torch.manual_seed(0)
d = 10
net = nets.FullyConnectedUNet(d, [256, 128, 64]).to(dev)
for tiles in (72, 256, 512, 584, 1024, 1536, 1608, 2048):
    N = tiles * 16
    x = torch.randn(N, d, device=dev)
    ts = torch.linspace(0, 1, N, device=dev)
    gout = torch.randn(N, d, device=dev)
    for _ in range(3):
        nets.unet_backward_hip(net, x, ts, 1, gout)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        nets.unet_backward_hip(net, x, ts, 1, gout)
    b.record(); torch.cuda.synchronize()
    print(f"tiles={tiles:5d} rows={N:6d}: A+B+C {a.elapsed_time(b) / 10 * 1e3:7.1f} us")
