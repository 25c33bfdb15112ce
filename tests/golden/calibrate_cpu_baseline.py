#!/usr/bin/env python3
"""Side-by-side CPU timing of the REFERENCE rollout and the oracle's rollout.

Test tooling for the authoring container (lives beside the fixture generator): imports /root/reference (never shipped).  Establishes that the oracle
("port" cpu_baseline of bench.py) costs what the reference costs on the same host, so that the number
bench.py reports on the GPU box stands in for the reference's CPU path.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/calibrate_cpu_baseline.py
"""
import os
import statistics
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, HERE]
import torch

import make_golden as G                       # the fixture generator's reference-import recipe
from oracle import socm_oracle as O


def main():
    utils, method, models, settings = G._import_reference()
    d, K, B, lmbd = 10, 200, 128, 1.0
    torch.manual_seed(0)
    kappa, nu = torch.ones(d), torch.ones(d)
    kappa[:3], nu[:3] = 5, 3
    sigma = torch.eye(d)
    sde = settings["DoubleWell"](device="cpu", dim=d, hdims=[256, 128, 64], hdims_M=[128, 128], u=None, lmbd=lmbd,
                                 kappa=kappa, nu=nu, sigma=sigma, gamma=6.0, scaling_factor_nabla_V=1.0,
                                 scaling_factor_M=0.1)
    sde.initialize_models()
    ts = torch.linspace(0, 1.0, K + 1)
    x0 = torch.zeros(B, d)
    vp = {k: v.detach() for k, v in sde.nabla_V.state_dict().items()}
    pb = dict(kind="double_well", sigma=sigma, kappa=kappa, nu=nu)
    noise = torch.randn(K, B, d)
    rows = []
    for threads in (1, os.cpu_count()):
        torch.set_num_threads(threads)
        res = {}
        for name, fn in (("reference", lambda: utils.stochastic_trajectories(sde, x0, ts, lmbd)),
                         ("oracle", lambda: O.stochastic_trajectories(pb, vp, x0, ts, lmbd, noise))):
            with torch.no_grad():
                for _ in range(3):
                    fn()
                t = []
                for _ in range(15):
                    t0 = time.perf_counter()
                    fn()
                    t.append(time.perf_counter() - t0)
            res[name] = (statistics.median(t), min(t), max(t))
        rows.append((threads, res))
        r, o = res["reference"][0], res["oracle"][0]
        print(f"threads={threads}: reference {1e3*r:.1f} ms (min {1e3*res['reference'][1]:.1f}, max "
              f"{1e3*res['reference'][2]:.1f}); oracle {1e3*o:.1f} ms (min {1e3*res['oracle'][1]:.1f}, max "
              f"{1e3*res['oracle'][2]:.1f}); oracle/reference = {o/r:.3f}")


if __name__ == "__main__":
    main()
