"""`define_variables(cfg, ts)`: the `method.setting` string dispatch of reference
experiment_settings/settings.py:210-301, returning `(x0, sigma, optimal_sde, neural_sde,
u_warm_start)`.

Constants and the ORDER of RNG draws follow the reference (x0 / xi are drawn right after seeding,
before the networks are initialised) so that a run with the reference's seed starts from the same
problem instance and the same weights.  Ground-truth controls (socmx.ground_truth): closed forms for the OU
settings, the vectorised 1-D PDE solve for double_well; `None` for molecular_dynamics as in the reference
(settings.py:112-114).  Warm start (splines) is out of
scope: `u_warm_start` is always None and `method.use_warm_start=True` raises.
"""
import torch

from . import ground_truth


def define_variables(cfg, ts):
    from SOC_matching.experiment_settings.OU_quadratic import OU_Quadratic
    from SOC_matching.experiment_settings.OU_linear import OU_Linear
    from SOC_matching.experiment_settings.double_well import DoubleWell
    from SOC_matching.experiment_settings.molecular_dynamics import MolecularDynamics

    m = cfg.method
    dev, d = m.device, m.d
    if m.use_warm_start:
        raise NotImplementedError("method.use_warm_start=True (spline warm start) is out of scope")
    common = dict(device=dev, dim=d, hdims=list(cfg.arch.hdims), hdims_M=list(cfg.arch.hdims_M), lmbd=m.lmbd,
                  gamma=m.gamma, scaling_factor_nabla_V=m.scaling_factor_nabla_V,
                  scaling_factor_M=m.scaling_factor_M)
    eye = torch.eye(d).to(dev)
    setting = m.setting
    optimal_sde = None
    if setting in ("OU_quadratic_easy", "OU_quadratic_hard"):
        x0 = torch.tensor([0.4, 0.6]).to(dev) if d == 2 else 0.5 * torch.randn(d).to(dev)
        print(f"x0: {x0}")
        a, p, q = (1.0, 1.0, 0.5) if setting == "OU_quadratic_hard" else (0.2, 0.2, 0.1)
        sigma, A, P, Q = eye, a * eye, p * eye, q * eye
        optimal_sde = ground_truth.lq_optimal_sde(OU_Quadratic, ts, sigma, A, P, Q, cfg)
        sde = OU_Quadratic(A=A, P=P, Q=Q, sigma=sigma, u_warm_start=None, use_warm_start=False, **common)
    elif setting == "OU_linear":
        x0 = torch.zeros(d).to(dev)
        xi = 0.1 * torch.randn(d, d).to(dev)
        omega = torch.ones(d).to(dev)
        A, sigma = -eye + xi, eye + xi
        optimal_sde = ground_truth.linear_optimal_sde(OU_Linear, ts, sigma, A, omega, cfg)
        sde = OU_Linear(A=A, omega=omega, sigma=sigma, **common)
    elif setting == "double_well":
        print("double_well")
        x0 = torch.zeros(d).to(dev)
        kappa, nu = torch.ones(d).to(dev), torch.ones(d).to(dev)
        kappa[:3] = 5
        nu[:3] = 3
        sigma = eye
        optimal_sde = ground_truth.double_well_optimal_sde(DoubleWell, kappa, nu, sigma, cfg)
        sde = DoubleWell(kappa=kappa, nu=nu, sigma=sigma, **common)
    elif setting == "molecular_dynamics":
        print("molecular_dynamics")
        x0 = -torch.ones(d).to(dev)
        kappa, sigma = torch.ones(d).to(dev), eye
        sde = MolecularDynamics(kappa=kappa, sigma=sigma, use_stopping_time=m.use_stopping_time, **common)
    else:
        raise ValueError(f"unknown method.setting {setting!r} (supported: OU_quadratic_easy, OU_quadratic_hard, "
                         "OU_linear, double_well, molecular_dynamics)")
    sde.initialize_models()
    return x0, sigma, optimal_sde, sde, None
