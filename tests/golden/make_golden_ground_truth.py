#!/usr/bin/env python3
"""Golden vectors for rollouts under the reference's GROUND-TRUTH controls.

Test tooling, not product code; runs only where `/root/reference` exists:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_ground_truth.py

For each setting the unmodified reference builds its optimal SDE with
`experiment_settings.settings.ground_truth_control(cfg, ts, x0, ...)` (settings.py:25-114:
`LinearControl` from `optimal_control_LQ` / `solution_Ricatti`, utils.py:234-258;
`ConstantControlLinear` from `exponential_t_A`; `LowDimControl` from the double-well PDE solve,
double_well.py:132-233) and then runs, on injected noise,
  * `utils.stochastic_trajectories(optimal_sde, state0, ts, lmbd)` with DISTINCT initial rows
    (utils.py:17-128 through `NeuralSDE.control`'s non-learned branch, method.py:103-107), and
  * `utils.control_objective(optimal_sde, x0, ts, lmbd, batch, total_n_samples)` (utils.py:131-163,
    the call of main.py:137-150).
Stored: the problem constants, the reference's control TABLE (`ut`), inputs and all outputs.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import _import_reference, _NoiseFeeder  # noqa: E402


def _cfg(setting, d, T, lmbd, delta_t=0.02, delta_x=0.05):
    m = types.SimpleNamespace(setting=setting, d=d, T=T, lmbd=lmbd, device="cpu", delta_t_optimal=delta_t,
                              delta_x_optimal=delta_x)
    return types.SimpleNamespace(method=m)


def make_one(name, setting, d, K, B, seed, n_batches=3, T=1.0, lmbd=1.0):
    utils, method, models, classes = _import_reference()
    from SOC_matching.experiment_settings import settings as ref_settings
    torch.manual_seed(seed)
    torch.set_num_threads(1)
    cfg = _cfg(setting, d, T, lmbd)
    ts = torch.linspace(0, T, K + 1)
    out = dict(meta_setting=np.array(setting), meta=np.array([d, K, B, seed, n_batches], dtype=np.int64),
               meta_f=np.array([T, lmbd, cfg.method.delta_t_optimal, cfg.method.delta_x_optimal, 2.75]))
    # constants exactly as settings.py:215-267 draws them
    if setting in ("OU_quadratic_easy", "OU_quadratic_hard"):
        x0 = torch.tensor([0.4, 0.6]) if d == 2 else 0.5 * torch.randn(d)
        sigma = torch.eye(d)
        c = (1.0, 1.0, 0.5) if setting == "OU_quadratic_hard" else (0.2, 0.2, 0.1)
        kw = dict(sigma=sigma, A=c[0] * torch.eye(d), P=c[1] * torch.eye(d), Q=c[2] * torch.eye(d))
    elif setting == "OU_quadratic_dense":
        # not a reference preset: dense non-symmetric matrices through the reference's own LQ solver and LinearControl
        x0 = 0.5 * torch.randn(d)
        sigma = torch.eye(d) + 0.2 * torch.randn(d, d)
        P = 0.3 * torch.randn(d, d)
        Q = 0.3 * torch.randn(d, d)
        kw = dict(sigma=sigma, A=-0.5 * torch.eye(d) + 0.3 * torch.randn(d, d), P=P @ P.T, Q=Q @ Q.T)
        cfg.method.setting = "OU_quadratic_easy"
    elif setting == "OU_linear":
        x0 = torch.zeros(d)
        xi = 0.1 * torch.randn(d, d)
        kw = dict(sigma=torch.eye(d) + xi, A=-torch.eye(d) + xi, omega=torch.ones(d))
    elif setting == "double_well":
        x0 = torch.zeros(d)
        kappa, nu = torch.ones(d), torch.ones(d)
        kappa[:3], nu[:3] = 5, 3
        kw = dict(sigma=torch.eye(d), kappa=kappa, nu=nu)
    else:
        raise ValueError(setting)
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        optimal_sde = ref_settings.ground_truth_control(cfg, ts, x0, **kw)
    assert not optimal_sde.use_learned_control
    u = optimal_sde.u
    table = u.u if hasattr(u, "u") else u.ut
    out["ut"] = table.detach().to(torch.float32).numpy().copy()
    out["ut_dtype"] = np.array(str(table.dtype))
    for k, v in kw.items():
        out["const_" + k] = v.numpy().copy()
    out["const_x0"] = x0.numpy().copy()
    out["ts"] = ts.numpy().copy()

    # ---- one rollout from distinct rows -----------------------------------------------------------
    state0 = x0 + 0.3 * torch.randn(B, d)
    noise = torch.randn(K, B, d)
    out["state0"] = state0.numpy().copy()
    out["noise_in"] = noise.numpy().copy()
    with _NoiseFeeder(noise), torch.no_grad():
        r = utils.stochastic_trajectories(optimal_sde, state0, ts, lmbd)
    names = ["states", "noises", "stop_indicators", "fractional_timesteps", "lpd", "lps", "ltw", "controls"]
    for n, v in zip(names, r):
        out["roll_" + n] = v.to(torch.float32).numpy().copy()
    # the tensor-time lookup the L2-error metric uses (method.py:858-873 / utils.py:196-199)
    with torch.no_grad():
        out["u_on_trajectory"] = u(ts, r[0], t_is_tensor=True).to(torch.float32).numpy().copy()

    # ---- the optimal-control burst of main.py:137-150 ---------------------------------------------
    Bb = max(4, B // 2)
    burst_noise = torch.randn(K * n_batches, Bb, d)        # consumed batch after batch, step-major inside a batch
    with _NoiseFeeder(burst_noise), torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        m, e = utils.control_objective(optimal_sde, x0, ts, lmbd, Bb, total_n_samples=Bb * n_batches)
    out["burst_noise"] = burst_noise.numpy().copy()
    out["burst_batch"] = np.array(Bb, dtype=np.int64)
    out["burst_mean"] = np.array(float(m))
    out["burst_std_err"] = np.array(float(e))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB  table {tuple(table.shape)} {table.dtype}  "
          f"burst mean {float(m):.6g}")


FIXTURES = [
    ("gt_ou_quadratic_easy_d2", "OU_quadratic_easy", 2, 50, 37, 21),
    ("gt_ou_quadratic_hard_d20", "OU_quadratic_hard", 20, 30, 40, 22),
    ("gt_ou_quadratic_dense_d5", "OU_quadratic_dense", 5, 24, 19, 26),
    ("gt_ou_linear_d10", "OU_linear", 10, 40, 33, 23),
    ("gt_ou_linear_d64", "OU_linear", 64, 25, 20, 24),
    ("gt_double_well_d4", "double_well", 4, 30, 50, 25),
]

if __name__ == "__main__":
    only = sys.argv[1:]
    for name, setting, d, K, B, seed in FIXTURES:
        if only and name not in only:
            continue
        make_one(name, setting, d, K, B, seed)
