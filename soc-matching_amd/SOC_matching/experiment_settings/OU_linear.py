"""`OU_Linear` setting: b = A x, f = 0, g = omega . x.

Constructor signature of reference experiment_settings/OU_linear.py:11-40."""
import torch

from SOC_matching import method
from socmx import _lib
from socmx.problems import Problem


class OU_Linear(method.NeuralSDE):
    def __init__(self, device="cuda", dim=2, u=None, hdims=[256, 128, 64], hdims_M=[128, 128], lmbd=1.0,
                 A=torch.eye(2), sigma=torch.eye(2), omega=torch.ones(2), gamma=3.0,
                 scaling_factor_nabla_V=1.0, scaling_factor_M=1.0):
        super().__init__(device=device, dim=dim, hdims=hdims, hdims_M=hdims_M, u=u, lmbd=lmbd, sigma=sigma,
                         gamma=gamma, scaling_factor_nabla_V=scaling_factor_nabla_V,
                         scaling_factor_M=scaling_factor_M)
        self.A, self.omega = A, omega
        self.problem = Problem(_lib.OU_LINEAR, dim, sigma, A=A, omega=omega)
