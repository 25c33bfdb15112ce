#!/usr/bin/env python3
"""Side-by-side CPU timing of ONE SOCM ITERATION of the reference (main.py:280-352: loss + backward + Adam step) and of the
oracle's faithful dense restatement (oracle.socm_loss, derivative="jacrev"), at BASELINE configs[2] (double_well d=10, K=200).

Test tooling for the authoring container (imports /root/reference, never shipped).  Establishes what bench.py's
`cpu_baseline.socm_ms_per_iter` (the oracle, timed on the GPU box's host) stands for relative to the reference itself.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/calibrate_cpu_iteration.py [B ...]      (default: 32 128)
"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, HERE]
import torch

import make_golden as G
from oracle import socm_oracle as O


def one(B, threads):
    utils, method, models, settings = G._import_reference()
    d, K, lmbd, T = 10, 200, 1.0, 1.0
    torch.manual_seed(0)
    torch.set_num_threads(threads)
    kappa, nu = torch.ones(d), torch.ones(d)
    kappa[:3], nu[:3] = 5, 3
    sigma = torch.eye(d)
    sde = settings["DoubleWell"](device="cpu", dim=d, hdims=[256, 128, 64], hdims_M=[128, 128], lmbd=lmbd,
                                 kappa=kappa, nu=nu, sigma=sigma, gamma=6.0, scaling_factor_nabla_V=1.0,
                                 scaling_factor_M=0.1)
    sde.initialize_models()
    ts = torch.linspace(0, T, K + 1)
    x0 = torch.zeros(d)
    solver = method.SOC_Solver(sde, x0, None, T=T, num_steps=K, lmbd=lmbd, d=d, sigma=sigma)
    # main.py:188-230: Adam groups {nabla_V: lr 1e-4; M.sigmoid_layers, gamma: M_lr}, eps 1e-4
    opt = torch.optim.Adam([{"params": sde.nabla_V.parameters()}, {"params": sde.M.sigmoid_layers.parameters(), "lr": 1e-3},
                            {"params": sde.gamma, "lr": 1e-3}], lr=1e-4, eps=1e-4)
    res = {}
    t0 = time.perf_counter()
    out = solver.loss(B, compute_L2_error=False, algorithm="SOCM", use_warm_start=False, use_stopping_time=False)
    (out[0] / 1.0).backward()
    opt.step()
    opt.zero_grad()
    res["reference"] = time.perf_counter() - t0
    # the oracle on its own copies of the same weights
    vp = {k: v.detach().clone().requires_grad_(True) for k, v in sde.nabla_V.state_dict().items()}
    mp = {k: v.detach().clone().requires_grad_(True) for k, v in sde.M.state_dict().items() if k != "gamma"}
    gamma = sde.gamma.detach().clone().requires_grad_(True)
    pb = dict(kind="double_well", sigma=sigma, kappa=kappa, nu=nu)
    noise = torch.randn(K, B, d)
    oopt = torch.optim.Adam([{"params": list(vp.values())}, {"params": list(mp.values()), "lr": 1e-3},
                             {"params": [gamma], "lr": 1e-3}], lr=1e-4, eps=1e-4)
    t0 = time.perf_counter()
    obj, _, _ = O.socm_loss(pb, vp, mp, gamma, x0, ts, T, lmbd, B, noise, derivative="jacrev")
    (obj / 1.0).backward()
    oopt.step()
    oopt.zero_grad()
    res["oracle"] = time.perf_counter() - t0
    return res


def main():
    Bs = [int(a) for a in sys.argv[1:]] or [32, 128]
    for B in Bs:
        for threads in (1, os.cpu_count()):
            r = one(B, threads)
            print(f"B={B} threads={threads}: reference iteration {r['reference']:.2f} s, oracle iteration {r['oracle']:.2f} s, "
                  f"oracle/reference = {r['oracle'] / r['reference']:.3f}", flush=True)


if __name__ == "__main__":
    main()
