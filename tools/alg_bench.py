#!/usr/bin/env python3
"""Eager vs replayed iteration for every algorithm at BASELINE configs[2] (which default Trainer picks is decided from this).
   python tools/alg_bench.py [nosave]   -- nosave: Trainer(save_activations=False), the re-computing control-network backward"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "soc-matching_amd")]
import torch, bench
from socmx.train import Trainer, make_optimizer
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
SAVE = not (len(sys.argv) > 1 and sys.argv[1] == "nosave")
for alg in ("SOCM", "SOCM_const_M", "SOCM_adjoint", "cross_entropy", "log-variance", "moment", "variance", "rel_entropy"):
    for graph in (False, True):
        cfg, ts, x0, sde, solver = bench.build(dev, "double_well", 10, 200, 6.0, 128)
        solver.gamma = 6.0
        opt = make_optimizer(solver, nabla_V_lr=1e-4, M_lr=1e-3, adam_eps=1e-4, algorithm=alg)
        tr = Trainer(solver, opt, batch_size=128, normalization_const=1.0, algorithm=alg, sync_timing=False, hip_graph="force" if graph else False,
                     save_activations=SAVE)
        n = 30 if alg != "rel_entropy" else 6
        for _ in range(5): tr.step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): tr.step()
        torch.cuda.synchronize()
        print(f"{alg:14s} {'graph' if graph else 'eager'}: {1e3*(time.perf_counter()-t0)/n:8.3f} ms/iteration  (graph mode active: {tr.hip_graph})", flush=True)
