"""Problem descriptors: the closed-form pieces of each `method.setting`.

The reference expresses a setting as a `NeuralSDE` subclass with Python
callbacks `b, nabla_b, f, nabla_f, g, nabla_g[, Phi]`
(experiment_settings/OU_quadratic.py:51-83, OU_linear.py:43-96,
double_well.py:44-97, molecular_dynamics.py:49-99).  Here a setting is DATA -- a
kind tag plus a few constant tensors -- so the HIP kernels can fuse the math, and
the same descriptor provides the torch closed forms used by the compat classes.

`nabla_b_T_apply(x, v)` returns (nabla_b(x))^T-contracted v, i.e.
out_l = sum_n d b_n / d x_l * v_n, without ever building the (…,d,d) Jacobian the
reference materialises (OU_quadratic.py:55-63, double_well.py:51-61).
"""
import torch

from . import _lib

KIND_OF_SETTING = {
    "OU_quadratic_easy": _lib.OU_QUADRATIC,
    "OU_quadratic_hard": _lib.OU_QUADRATIC,
    "OU_linear": _lib.OU_LINEAR,
    "double_well": _lib.DOUBLE_WELL,
    "molecular_dynamics": _lib.MOLECULAR_DYNAMICS,
}


class Problem:
    def __init__(self, kind, d, sigma, A=None, P=None, Q=None, omega=None, kappa=None, nu=None):
        self.kind = int(kind)
        self.d = int(d)
        self.sigma = sigma
        self.A, self.P, self.Q, self.omega, self.kappa, self.nu = A, P, Q, omega, kappa, nu
        self._cache = {}

    def __getstate__(self):  # ctypes structs / device-side copies are rebuilt on demand; keep the solver picklable
        st = self.__dict__.copy()
        st["_cache"] = {}
        return st

    # ---- constants -------------------------------------------------------
    @property
    def has_phi(self):
        return self.kind == _lib.MOLECULAR_DYNAMICS

    def sigma_inv_t(self):
        key = ("sit", self.sigma.device)
        if key not in self._cache:
            self._cache[key] = torch.transpose(torch.inverse(self.sigma), 0, 1).contiguous()
        return self._cache[key]

    def tensors(self):
        return dict(sigma=self.sigma, A=self.A, P=self.P, Q=self.Q, omega=self.omega, kappa=self.kappa,
                    nu=self.nu)

    def to(self, device):
        kw = {k: (v.to(device) if v is not None else None) for k, v in self.tensors().items()}
        return Problem(self.kind, self.d, **kw)

    def c_struct(self):
        """socmx_problem for the C ABI; keeps the fp32-contiguous device tensors alive on self."""
        dev = self.sigma.device
        key = ("c", dev)
        if key not in self._cache:
            keep = {}
            for k, v in self.tensors().items():
                keep[k] = None if v is None else v.detach().to(torch.float32).contiguous()
            keep["sigma_inv_t"] = self.sigma_inv_t().to(torch.float32).contiguous()
            ident = bool(torch.equal(keep["sigma"].cpu(), torch.eye(self.d)))   # one host sync, at setup time
            s = _lib.Problem(kind=self.kind, d=self.d, flags=_lib.SIGMA_IDENTITY if ident else 0)
            for k, v in keep.items():
                setattr(s, k, _lib.ptr(v))
            self._cache[key] = (s, keep)
        return self._cache[key][0]

    # ---- closed forms (any leading batch shape) ----------------------------------
    def b(self, t, x):
        if self.kind in (_lib.OU_QUADRATIC, _lib.OU_LINEAR):
            return x @ self.A.T
        return -4.0 * self.kappa * x * (x * x - 1.0)

    def nabla_b(self, t, x):
        """Dense Jacobian with the reference's index convention out[..., l, n] = d b_n / d x_l
        (only for API compatibility; the loss never calls it)."""
        if self.kind in (_lib.OU_QUADRATIC, _lib.OU_LINEAR):
            return self.A.T.expand(*x.shape[:-1], self.d, self.d)
        return torch.diag_embed(-(12.0 * self.kappa * x * x - 4.0 * self.kappa))

    def nabla_b_T_apply(self, x, v):
        if self.kind in (_lib.OU_QUADRATIC, _lib.OU_LINEAR):
            return v @ self.A                      # sum_n A[n,l] v_n
        return -(12.0 * self.kappa * x * x - 4.0 * self.kappa) * v

    def f(self, t, x):
        if self.kind == _lib.OU_QUADRATIC:
            return (x * (x @ self.P.T)).sum(-1)
        if self.kind == _lib.MOLECULAR_DYNAMICS:
            return torch.ones_like(x[..., 0])
        return torch.zeros_like(x[..., 0])

    def nabla_f(self, t, x):
        if self.kind == _lib.OU_QUADRATIC:
            return 2.0 * (x @ self.P.T)
        return torch.zeros_like(x)

    def g(self, x):
        if self.kind == _lib.OU_QUADRATIC:
            return (x * (x @ self.Q.T)).sum(-1)
        if self.kind == _lib.OU_LINEAR:
            return x @ self.omega
        if self.kind == _lib.DOUBLE_WELL:
            return (self.nu * (x * x - 1.0) ** 2).sum(-1)
        return torch.zeros_like(x[..., 0])

    def nabla_g(self, x):
        if self.kind == _lib.OU_QUADRATIC:
            return 2.0 * (x @ self.Q.T)
        if self.kind == _lib.OU_LINEAR:
            return self.omega.expand_as(x).clone()
        if self.kind == _lib.DOUBLE_WELL:
            return 4.0 * self.nu * x * (x * x - 1.0)
        return torch.zeros_like(x)

    def Phi(self, x):
        return -x[..., 0]
