"""`OU_Quadratic` setting: b = A x, f = x'Px, g = x'Qx.

Constructor signature of reference experiment_settings/OU_quadratic.py:11-49; the
math itself lives in `socmx.problems.Problem` (kind OU_QUADRATIC) and, fused, in
the HIP kernels."""
import torch

from SOC_matching import method
from socmx import _lib
from socmx.problems import Problem


class OU_Quadratic(method.NeuralSDE):
    def __init__(self, device="cuda", dim=2, hdims=[256, 128, 64], hdims_M=[128, 128], u=None, lmbd=1.0,
                 A=torch.eye(2), P=torch.eye(2), Q=torch.eye(2), sigma=torch.eye(2), gamma=3.0,
                 scaling_factor_nabla_V=1.0, scaling_factor_M=1.0, T=1.0, u_warm_start=None,
                 use_warm_start=False):
        super().__init__(device=device, dim=dim, hdims=hdims, hdims_M=hdims_M, u=u, lmbd=lmbd, sigma=sigma,
                         gamma=gamma, scaling_factor_nabla_V=scaling_factor_nabla_V,
                         scaling_factor_M=scaling_factor_M, T=T, u_warm_start=u_warm_start,
                         use_warm_start=use_warm_start)
        self.A, self.P, self.Q = A, P, Q
        self.problem = Problem(_lib.OU_QUADRATIC, dim, sigma, A=A, P=P, Q=Q)
