"""Closed-form optimal controls for the linear settings.

* LQ (OU_quadratic): backward Riccati recursion F' = -(A^T F + F A - 2 F S S^T F + P), F(T)=Q on the
  simulation grid, u*(t,x) = -2 S^T F(t) x  (reference utils.py:234-254, models.py:10-52).
* OU_linear: u*(t) = -S^T exp(A^T (T-t)) omega  (reference settings.py:52-63, models.py:55-95).
Both are exposed as `sde.u(t, x, t_is_tensor=False)` callables on an un-learned NeuralSDE, which the
eager rollout path consumes (the fused kernel handles the learned control only).
"""
import torch


class LinearControl:
    """u(t,x) = U(t) x with U tabulated on the grid; lookup idx = floor((n-1) t / T) exactly as the
    reference does (models.py:15-23), fp round-down included."""

    def __init__(self, U, ts, T):
        self.U, self.ts, self.T = U, ts, T

    def _index(self, t):
        return torch.floor((self.U.shape[0] - 1) * t / self.T).to(torch.int64)

    def __call__(self, t, x, t_is_tensor=False):
        if not t_is_tensor:
            Ut = self.U[self._index(t)]
            Ut = Ut if Ut.dim() == 2 else Ut[0]
            return x @ Ut.T
        Ut = self.U[self._index(t)]
        if x.dim() == 2:
            return torch.einsum("bij,bj->bi", Ut, x)
        return torch.einsum("aij,abj->abi", Ut, x)


class ConstantControl:
    """u(t,x) = c(t), independent of x."""

    def __init__(self, C, ts, T):
        self.C, self.ts, self.T = C, ts, T

    def __call__(self, t, x, t_is_tensor=False):
        n = self.C.shape[0]
        if not t_is_tensor:
            # quirk kept from the reference (models.py:72-75): the scalar-time lookup scales by n, not n-1
            idx = torch.floor(n * t / self.T).to(torch.int64)
            return self.C[idx].unsqueeze(0).repeat(x.shape[0], 1)
        idx = torch.floor((n - 1) * t / self.T).to(torch.int64)
        return self.C[idx, :].unsqueeze(1).repeat(1, x.shape[1], 1)


def riccati(sigma, A, P, Q, ts):
    """F on the grid, backward Euler in the reference's discretisation (utils.py:234-248)."""
    R_inv = sigma @ sigma.T
    F = Q
    out = [F]
    dts = ts[1:] - ts[:-1]
    for dt in dts:   # the reference walks the (uniform) grid forward while integrating backward in time
        F = F - dt * (-(A.T @ F) - F @ A + 2 * F @ R_inv @ F - P)
        out.append(F)
    out.reverse()
    return torch.stack(out)


def lq_optimal_sde(cls, ts, sigma, A, P, Q, cfg):
    U = -2 * torch.einsum("ij,bjk->bik", sigma.T, riccati(sigma, A, P, Q, ts))
    sde = cls(device=cfg.method.device, dim=cfg.method.d, u=LinearControl(U, ts, cfg.method.T), lmbd=cfg.method.lmbd,
              A=A, P=P, Q=Q, sigma=sigma)
    return sde


def linear_optimal_sde(cls, ts, sigma, A, omega, cfg):
    T = cfg.method.T
    expo = torch.matrix_exp((T - ts).reshape(-1, 1, 1) * A.T.unsqueeze(0))     # exp(A^T (T-t))
    C = -torch.einsum("ij,aj->ai", sigma.T, expo @ omega)
    return cls(device=cfg.method.device, dim=cfg.method.d, u=ConstantControl(C, ts, T), lmbd=cfg.method.lmbd,
               A=A, omega=omega, sigma=sigma)
