#!/usr/bin/env python3
"""Soak run of the BASELINE configs[2] training loop: 3000 iterations with the two-stream schedule, then with the
sequential one, then two-stream again (same Philox keys).  Prints steady-state time, the loss trace, peak memory
and gamma; the schedules must agree to fp32 round-off (they did: 1e-6 relative after 3000 Adam steps)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "soc-matching_amd")]
import torch, bench as Bn
from socmx.train import Trainer, make_optimizer
from socmx import rollout
dev = torch.device("cuda:0")
res = {}
for overlap in (True, False, True):
    torch.manual_seed(0)
    rollout._philox_calls = 0
    cfg, ts, x0, sde, solver = Bn.build(dev)
    opt = make_optimizer(solver, M_lr=1e-3)
    tr = Trainer(solver, opt, 128, sync_timing=False, overlap_M_backward=overlap)
    solver.overlap_M = overlap
    losses = []
    for it in range(3000):
        if it == 300:
            torch.cuda.synchronize(); t0 = time.time()
        out = tr.step()
        if it % 500 == 0 or it == 2999:
            losses.append(float(out["loss"]))
    torch.cuda.synchronize()
    res[overlap] = losses
    print("overlap" if overlap else "sequential", "%.1f s" % (time.time() - t0), ["%.4f" % l for l in losses],
          "mem MB", torch.cuda.max_memory_allocated() // 2**20, "gamma", float(sde.gamma), flush=True)
a, b = res[True], res[False]
print("max rel diff of logged losses:", max(abs(x - y) / max(abs(y), 1e-9) for x, y in zip(a, b)))
