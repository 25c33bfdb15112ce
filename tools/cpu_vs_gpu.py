#!/usr/bin/env python3
"""Is the SOCM iteration host-bound or GPU-bound?  Times the host-side enqueue of N iterations (no sync inside)
against the wall time to completion, with the second-stream overlaps on and off."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "soc-matching_amd")]
import torch
import bench as Bn
from socmx.train import Trainer, make_optimizer

dev = torch.device("cuda:0")
cfg, ts, x0, sde, solver = Bn.build(dev)
opt = make_optimizer(solver, M_lr=1e-3)
tr = Trainer(solver, opt, Bn.BATCH_PER_GPU, sync_timing=False)
for mode in ("overlap", "no-overlap", "overlap"):
    solver.overlap_M = mode == "overlap"
    for _ in range(5):
        tr.step()
    torch.cuda.synchronize()
    N = 30
    t0 = time.perf_counter()
    for _ in range(N):
        tr.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{mode}: host enqueue {1e3*(t1-t0)/N:.3f} ms/it, to completion {1e3*(t2-t0)/N:.3f} ms/it")
