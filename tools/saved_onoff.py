#!/usr/bin/env python3
"""Replayed SOCM iteration with the saved-activation backward on / off for a few settings (developer timing):
   python tools/saved_onoff.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "soc-matching_amd")]
import torch, bench
from socmx.train import Trainer, make_optimizer
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
CASES = [("double_well", 10, 200, 6.0, 128), ("OU_quadratic_easy", 2, 50, 2.0, 128), ("OU_linear", 10, 100, 2.0, 64),
         ("OU_quadratic_easy", 10, 100, 2.0, 128), ("OU_quadratic_hard", 14, 100, 2.0, 128), ("OU_linear", 14, 100, 2.0, 128),
         ("OU_linear", 3, 100, 2.0, 128), ("double_well", 14, 100, 2.0, 128)]
for setting, d, K, gamma, B in CASES:
    row = []
    for save in (True, False):
        best = 1e9
        for rep in range(2):
            cfg, ts, x0, sde, solver = bench.build(dev, setting, d, K, gamma, B)
            opt = make_optimizer(solver, nabla_V_lr=1e-4, M_lr=1e-3, adam_eps=1e-4)
            tr = Trainer(solver, opt, batch_size=B, normalization_const=1.0, sync_timing=False, hip_graph=True, save_activations=save)
            for _ in range(6): tr.step()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(40): tr.step()
            torch.cuda.synchronize()
            best = min(best, 1e3 * (time.perf_counter() - t0) / 40)
            used = (tr._dev or {}).get("saved") is not None
        row.append((best, used))
    print(f"{setting:18s} d={d:2d} K={K:3d} B={B:3d}: saved {row[0][0]:.3f} ms (active: {row[0][1]})   re-computing {row[1][0]:.3f} ms", flush=True)
