"""`DoubleWell` setting: b = -4 kappa x (x^2 - 1), f = 0, g = sum nu (x^2-1)^2.

Constructor signature of reference experiment_settings/double_well.py:12-41.  (The
reference file also holds the 1-D PDE ground-truth solver, double_well.py:99-233;
that is SURVEY row f2 and not part of this module yet.)"""
import torch

from SOC_matching import method
from socmx import _lib
from socmx.problems import Problem


class DoubleWell(method.NeuralSDE):
    def __init__(self, device="cuda", dim=2, hdims=[256, 128, 64], hdims_M=[128, 128], u=None, lmbd=1.0,
                 kappa=torch.ones(2), nu=torch.ones(2), sigma=torch.eye(2), gamma=3.0,
                 scaling_factor_nabla_V=1.0, scaling_factor_M=1.0):
        super().__init__(device=device, dim=dim, hdims=hdims, hdims_M=hdims_M, u=u, lmbd=lmbd, sigma=sigma,
                         gamma=gamma, scaling_factor_nabla_V=scaling_factor_nabla_V,
                         scaling_factor_M=scaling_factor_M)
        self.kappa, self.nu = kappa, nu
        self.problem = Problem(_lib.DOUBLE_WELL, dim, sigma, kappa=kappa, nu=nu)

    def potential(self, x):
        return (self.kappa * (x * x - 1.0) ** 2).sum(-1)
