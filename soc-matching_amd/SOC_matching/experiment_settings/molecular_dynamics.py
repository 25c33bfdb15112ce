"""`MolecularDynamics` setting: double-well drift, f = 1, g = 0, stops when
Phi(x) = -x_0 turns negative.

Constructor signature of reference experiment_settings/molecular_dynamics.py:11-47.
Having a `Phi` attribute is what switches the rollout's stopping logic on
(reference utils.py:33)."""
import torch

from SOC_matching import method
from socmx import _lib
from socmx.problems import Problem


class MolecularDynamics(method.NeuralSDE):
    def __init__(self, device="cuda", dim=2, hdims=[256, 128, 64], hdims_M=[128, 128], u=None, lmbd=1.0,
                 kappa=torch.ones(2), sigma=torch.eye(2), gamma=3.0, scaling_factor_nabla_V=1.0,
                 scaling_factor_M=1.0, T=1.0, u_warm_start=None, use_warm_start=False,
                 use_stopping_time=False):
        super().__init__(device=device, dim=dim, hdims=hdims, hdims_M=hdims_M, u=u, lmbd=lmbd, sigma=sigma,
                         gamma=gamma, scaling_factor_nabla_V=scaling_factor_nabla_V,
                         scaling_factor_M=scaling_factor_M, T=T, u_warm_start=u_warm_start,
                         use_warm_start=use_warm_start, use_stopping_time=use_stopping_time)
        self.kappa = kappa
        self.problem = Problem(_lib.MOLECULAR_DYNAMICS, dim, sigma, kappa=kappa)

    def Phi(self, x):
        return self.problem.Phi(x)
