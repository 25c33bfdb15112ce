"""Ground-truth optimal controls.

* LQ (OU_quadratic): backward Riccati recursion F' = -(A^T F + F A - 2 F S S^T F + P), F(T)=Q on the
  simulation grid, u*(t,x) = -2 S^T F(t) x  (reference utils.py:234-254, models.py:10-52).
* OU_linear: u*(t) = -S^T exp(A^T (T-t)) omega  (reference settings.py:52-63, models.py:55-95).
* double_well: the problem decouples per coordinate; each coordinate's value function follows from a 1-D
  linear PDE for psi = exp(-V) solved backward in time with an implicit (tridiagonal) step on a symmetrised
  finite-volume generator, u* = -(2/beta) sigma_ii d/dx(-log psi)  (reference double_well.py:132-233,
  table lookup models.py:98-150).  Vectorised here: the reference fills the tridiagonal matrix and the
  control table with Python double loops (5.5 M iterations per coordinate at the README resolution) and
  repeats the solve for every coordinate; coordinates sharing (kappa_i, nu_i, sigma_ii) share one solve.
All are exposed as `sde.u(t, x, t_is_tensor=False)` callables on an un-learned NeuralSDE, which the
eager rollout path consumes (the fused kernel handles the learned control only).
"""
import numpy as np
import torch


class LinearControl:
    """u(t,x) = U(t) x with U tabulated on the grid; lookup idx = floor((n-1) t / T) exactly as the
    reference does (models.py:15-23), fp round-down included."""

    def __init__(self, U, ts, T):
        self.U, self.ts, self.T = U, ts, T

    def _index(self, t):
        return torch.floor((self.U.shape[0] - 1) * t / self.T).to(torch.int64)

    def hip_descriptor(self, ts):
        """(kind, table, per-step table row, n_x, xb, delta_x) for socmx_rollout_control_f32: the scalar-time lookup of
        __call__ evaluated for every step of the grid with the same fp32 arithmetic."""
        from . import _lib
        return _lib.CTRL_LINEAR, self.U, self._index(ts[:-1]), 0, 0.0, 1.0

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

    def hip_descriptor(self, ts):
        from . import _lib
        n = self.C.shape[0]
        return _lib.CTRL_CONSTANT, self.C, torch.floor(n * ts[:-1] / self.T).to(torch.int64), 0, 0.0, 1.0

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


# ---- double well ------------------------------------------------------------------------------------------

def double_well_table_1d(kappa_i, nu_i, sigma_ii, T, delta_t, delta_x, xb):
    """u*(t_n, x_i) on the reference's grid: (N+1, nx-1) float64.  beta = 2 as in the reference."""
    from scipy.linalg import solve_banded
    beta = 2.0
    nx = int(2.0 * xb / delta_x)
    N = int(T / delta_t)
    V = lambda x: kappa_i * (x * x - 1.0) ** 2
    i = np.arange(nx)
    xc = -xb + (i + 0.5) * delta_x                      # cell centres
    # off-diagonal between cells i and i+1 (symmetric), interface at -xb + (i+1) dx
    xi = -xb + (i[:-1] + 1) * delta_x
    off = -np.exp(beta * 0.5 * (V(xc[1:]) + V(xc[:-1]) - 2 * V(xi))) / delta_x**2
    diag = np.zeros(nx)
    diag[1:] += np.exp(beta * (V(xc[1:]) - V(-xb + i[1:] * delta_x))) / delta_x**2          # left interface
    diag[:-1] += np.exp(beta * (V(xc[:-1]) - V(-xb + (i[:-1] + 1) * delta_x))) / delta_x**2  # right interface
    off, diag = -off / beta, -diag / beta
    band = -delta_t * np.vstack([np.append([0.0], off), diag - N / T, np.append(off, [0.0])])
    xv = np.linspace(-xb, xb, nx, endpoint=True)
    D, Dinv = np.exp(beta * V(xv) / 2), np.exp(-beta * V(xv) / 2)
    psi = np.zeros((N + 1, nx))
    psi[N] = np.exp(-nu_i * (xv * xv - 1.0) ** 2)
    for n in range(N - 1, -1, -1):
        psi[n] = D * solve_banded((1, 1), band, Dinv * psi[n + 1])
    logp = np.log(psi)
    return -2.0 / beta * sigma_ii * (logp[:, :-1] - logp[:, 1:]) / delta_x


class LowDimControl:
    """Per-coordinate table lookup u_j(t, x_j) (reference models.py:98-150): time index ceil(t/delta_t), space
    index floor((x+xb)/delta_x) clamped to the table."""

    def __init__(self, ut, T, xb, dim, delta_t, delta_x):
        self.ut, self.T, self.xb, self.dim, self.delta_t, self.delta_x = ut, T, xb, dim, delta_t, delta_x

    def hip_descriptor(self, ts):
        from . import _lib
        ti = torch.ceil(ts[:-1] / self.delta_t).to(torch.int64)
        return _lib.CTRL_TABLE, self.ut, ti, int(self.ut.shape[1]), float(self.xb), float(self.delta_x)

    def _lookup(self, t_idx, x):
        ix = torch.floor((x + self.xb) / self.delta_x).to(torch.int64).clamp_(0, self.ut.shape[1] - 1)
        j = torch.arange(self.dim, device=x.device).expand_as(ix)
        return self.ut[t_idx.expand_as(ix), ix, j]

    def __call__(self, t, x, t_is_tensor=False):
        if not t_is_tensor:
            x2 = x.reshape(-1, self.dim)
            ti = torch.ceil(torch.as_tensor(t, device=x.device).reshape(1, 1) / self.delta_t).to(torch.int64)
            return self._lookup(ti, x2)
        ti = torch.ceil(t.reshape(-1, 1, 1) / self.delta_t).to(torch.int64)
        return self._lookup(ti, x)


def double_well_optimal_sde(cls, kappa, nu, sigma, cfg, xb=2.75):
    m = cfg.method
    tables, cache = [], {}
    for j in range(m.d):
        key = (float(kappa[j]), float(nu[j]), float(sigma[j, j]))
        if key not in cache:
            cache[key] = torch.from_numpy(double_well_table_1d(*key, m.T, m.delta_t_optimal, m.delta_x_optimal, xb))
        tables.append(cache[key])
    ut = torch.stack(tables, dim=2).to(torch.float32).to(m.device)   # fp32 like every other rollout operand
    print(f"ut_discrete.shape: {ut.shape}")
    sde = cls(device=m.device, dim=m.d, lmbd=m.lmbd, kappa=kappa, nu=nu, sigma=sigma)
    sde.u = LowDimControl(ut, m.T, xb, m.d, m.delta_t_optimal, m.delta_x_optimal)
    return sde
