"""`NeuralSDE`: the controlled SDE object the rollout and the loss operate on.

Mirrors the attribute surface of reference SOC_matching/method.py:15-143
(`device, dim, hdims, hdims_M, u, lmbd, sigma, gamma[,2,3], use_learned_control,
T, u_warm_start, use_warm_start, use_stopping_time, nabla_V, M`, methods
`control`, `initialize_models`, and the setting protocol `b, nabla_b, f, nabla_f,
g, nabla_g[, Phi]`), but carries the setting as a `Problem` descriptor
(`self.problem`) instead of per-subclass Python math.  Subclasses in
`SOC_matching/experiment_settings/` only translate constructor arguments.
"""
import torch

from . import nets
from .problems import Problem


class NeuralSDE(torch.nn.Module):
    noise_type = "diagonal"
    sde_type = "ito"

    def __init__(self, device="cuda", dim=2, hdims=(256, 128, 64), hdims_M=(128, 128), u=None, lmbd=1.0,
                 sigma=None, gamma=1.0, gamma2=1.0, gamma3=1.0, scaling_factor_nabla_V=1.0,
                 scaling_factor_M=1.0, T=1.0, u_warm_start=None, use_warm_start=False,
                 use_stopping_time=False, problem=None):
        super().__init__()
        self.device = device
        self.dim = dim
        self.hdims = list(hdims)
        self.hdims_M = list(hdims_M)
        self.u = u
        self.lmbd = lmbd
        self.sigma = sigma if sigma is not None else torch.eye(dim)
        self.gamma, self.gamma2, self.gamma3 = gamma, gamma2, gamma3
        self.scaling_factor_nabla_V = scaling_factor_nabla_V
        self.scaling_factor_M = scaling_factor_M
        self.use_learned_control = False
        self.T = T
        self.u_warm_start = u_warm_start
        self.use_warm_start = use_warm_start
        self.use_stopping_time = use_stopping_time
        self.problem = problem  # None for foreign subclasses that override b/f/g themselves

    # ---- setting protocol, served by the descriptor ------------------------------------
    def b(self, t, x):
        return self.problem.b(t, x)

    def nabla_b(self, t, x):
        return self.problem.nabla_b(t, x)

    def f(self, t, x):
        return self.problem.f(t, x)

    def nabla_f(self, t, x):
        return self.problem.nabla_f(t, x)

    def g(self, x):
        return self.problem.g(x)

    def nabla_g(self, x):
        return self.problem.nabla_g(x)

    # ---- control (method.py:58-107) ---------------------------------------------------------
    def control(self, t, x, verbose=False):
        if verbose:
            print(f"self.use_learned_control: {self.use_learned_control}, self.u: {self.u}")
        if not self.use_learned_control:
            return None if self.u is None else self.u(t, x)
        if x.dim() == 2:
            tx = torch.cat([t.reshape(-1, 1).expand(x.shape[0], 1), x], dim=-1)
        else:  # (Kp, B, d) with t of shape (Kp,)
            tx = torch.cat([t.reshape(-1, 1, 1).expand(x.shape[0], x.shape[1], 1), x], dim=-1)
        grad_v = self.nabla_V(tx.reshape(-1, tx.shape[-1])).reshape(x.shape)
        learned = -(grad_v @ self.sigma)  # -sigma^T grad_v, row-vector form
        if verbose:
            print(f"self.use_warm_start: {self.use_warm_start}, self.u_warm_start: {self.u_warm_start}")
        if self.use_warm_start and self.u_warm_start:
            return learned + self.u_warm_start(t, x).detach()
        return learned

    # ---- models (method.py:109-143) -----------------------------------------------------------
    def initialize_models(self):
        self.nabla_V = nets.FullyConnectedUNet(
            dim=self.dim, hdims=self.hdims, scaling_factor=self.scaling_factor_nabla_V).to(self.device)
        print(f"initialize_models, self.use_stopping_time: {self.use_stopping_time}")
        as_param = lambda v: torch.nn.Parameter(torch.tensor([v]).to(self.device))
        self.gamma = as_param(self.gamma)
        if self.use_stopping_time:
            self.gamma2 = as_param(self.gamma2)
            self.gamma3 = as_param(self.gamma3)
            self.M = nets.TwoBoundarySigmoidMLP(
                dim=self.dim, hdims=self.hdims_M, gamma=self.gamma, gamma2=self.gamma2, gamma3=self.gamma3,
                scaling_factor=self.scaling_factor_M).to(self.device)
        else:
            self.M = nets.SigmoidMLP(
                dim=self.dim, hdims=self.hdims_M, gamma=self.gamma,
                scaling_factor=self.scaling_factor_M).to(self.device)
        self.use_learned_control = True


def make_problem_sde(kind, cls=NeuralSDE, **kw):
    """Helper for the experiment_settings subclasses: split constants from NeuralSDE kwargs."""
    const = {k: kw.pop(k) for k in ("A", "P", "Q", "omega", "kappa", "nu") if k in kw}
    sde = cls.__new__(cls)
    NeuralSDE.__init__(sde, **kw)
    sde.problem = Problem(kind, sde.dim, sde.sigma, **const)
    for k, v in const.items():
        setattr(sde, k, v)
    return sde
