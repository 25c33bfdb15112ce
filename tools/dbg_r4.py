#!/usr/bin/env python3
"""Developer check: 4-row rollout kernel against the 16-row one on the same injected noise.
   python tools/dbg_r4.py run <out.npz> [K] [B]  (SOCMX_TILE_ROWS picks the kernel);  python tools/dbg_r4.py cmp a.npz b.npz"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "soc-matching_amd"), os.path.join(ROOT, "tests")]
import numpy as np
if sys.argv[1] == "cmp":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    for k in a.files:
        dif = np.abs(a[k] - b[k])
        print(f"{k:10s} shape {a[k].shape} max|diff| {dif.max():.3e}  first bad index {np.argwhere(dif > 1e-3)[:3].tolist()}")
    sys.exit(0)
import torch
from test_host_cpu import build_sde
from socmx import rollout as utils
K = int(sys.argv[3]) if len(sys.argv) > 3 else 2
B = int(sys.argv[4]) if len(sys.argv) > 4 else 8
sde, aux = build_sde("cfg3_double_well_d10_K200", "cuda:0")
torch.manual_seed(1)
ts = aux["ts"][: K + 1]
x0 = torch.randn(B, 10, device="cuda:0")
noise = torch.randn(K, B, 10, device="cuda:0")
r = utils.stochastic_trajectories(sde, x0, ts, aux["lmbd"], noise_in=noise, want_nabla_v=True)
nv = r[8]                                   # the network's output on the grid: the first thing to look at when the shapes disagree
np.savez(sys.argv[2], states=r[0].cpu().numpy(), controls=r[7].cpu().numpy(), lpd=r[4].cpu().numpy(), nabla_v=nv.cpu().numpy())
np.set_printoptions(linewidth=200, precision=4, suppress=True)
print(nv[0].cpu().numpy())
print("saved", sys.argv[2], [tuple(t.shape) for t in r[:3]])
