#!/usr/bin/env python3
"""Golden vectors for the double-well ground-truth control (reference double_well.py:132-233 PDE solve and
models.py:98-150 LowDimControl lookup), at a coarse grid so the fixture stays small.
Run here only:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_dw_pde.py"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import _import_reference

utils, method, models, classes = _import_reference()
d = 4
kappa = torch.tensor([5.0, 5.0, 1.0, 1.0]); nu = torch.tensor([3.0, 1.0, 3.0, 1.0])
sde = classes["DoubleWell"](device="cpu", dim=d, kappa=kappa, nu=nu, sigma=torch.eye(d), lmbd=1.0)
T, delta_t, delta_x, xb = 1.0, 0.02, 0.05, 2.75
uts = [sde.compute_reference_solution(T=T, delta_t=delta_t, xb=xb, delta_x=delta_x, lmbd=1.0, idx=j) for j in range(d)]
ut = torch.stack([torch.from_numpy(u) for u in uts], dim=2)
ctrl = models.LowDimControl(ut, T, xb, d, delta_t, delta_x)
torch.manual_seed(0)
ts = torch.linspace(0, T, 11)
xs = 1.5 * torch.randn(11, 6, d)
u3 = ctrl(ts, xs, t_is_tensor=True)
u2 = ctrl(ts[3], xs[3])
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "dw_pde_reference.npz"),
                    kappa=kappa.numpy(), nu=nu.numpy(), params=np.array([T, delta_t, delta_x, xb]), ut=ut.numpy(),
                    ts=ts.numpy(), xs=xs.numpy(), u_tensor=u3.numpy(), u_scalar=u2.numpy(), t_scalar=ts[3].numpy())
print("ut", ut.shape, float(ut.abs().max()))
