#!/usr/bin/env python3
"""Generate golden input/output vectors by importing the *reference* here.

Test tooling, not product code.  Runs only in the authoring container, where
`/root/reference` exists; the reference source never travels -- only the small
`.npz` files this script writes next to itself do.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What is pinned (all by the unmodified reference, CPU fp32):
  * `SOC_matching.utils.stochastic_trajectories`  (utils.py:17-128) -- the full
    8-tuple, with the per-step `torch.randn_like` draw replaced by an iterator
    over a pre-drawn `(K,B,d)` tensor so the vectors do not depend on the RNG.
  * `SOC_Solver.loss(algorithm="SOCM")` (method.py:223-262, 272-287, 480-720,
    897-906): objective, mean/std of the importance weight, and the gradient of
    the objective w.r.t. every nabla_V / M / gamma parameter.
  * `SigmoidMLP` on the pair grid and its d/ds by `functorch.jacrev`
    (method.py:510-515, 533-547, 565-573; models.py:245-275).

Two stub modules stand in for packages that are not installed here and are not
used on this path (`omegaconf`: utils.py:11; `nvidia_smi`: method.py:10).
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def _import_reference():
    sys.dont_write_bytecode = True
    om = types.ModuleType("omegaconf")
    om.DictConfig = dict
    om.OmegaConf = type("OmegaConf", (), {"create": staticmethod(lambda d: d)})
    sys.modules["omegaconf"] = om
    sys.modules["nvidia_smi"] = types.ModuleType("nvidia_smi")
    sys.path.insert(0, REF)
    from SOC_matching import utils, method, models  # noqa: F401
    from SOC_matching.experiment_settings.OU_quadratic import OU_Quadratic
    from SOC_matching.experiment_settings.OU_linear import OU_Linear
    from SOC_matching.experiment_settings.double_well import DoubleWell
    from SOC_matching.experiment_settings.molecular_dynamics import MolecularDynamics

    return utils, method, models, dict(
        OU_Quadratic=OU_Quadratic,
        OU_Linear=OU_Linear,
        DoubleWell=DoubleWell,
        MolecularDynamics=MolecularDynamics,
    )


class _NoiseFeeder:
    """Replaces torch.randn_like for the duration of one rollout."""

    def __init__(self, noise):
        self.noise = noise
        self.k = 0
        self.orig = torch.randn_like

    def __enter__(self):
        def fake(x, *a, **kw):
            out = self.noise[self.k].clone()
            assert out.shape == x.shape
            self.k += 1
            return out

        torch.randn_like = fake
        return self

    def __exit__(self, *exc):
        torch.randn_like = self.orig


def build_setting(classes, setting, d, hdims, hdims_M, gamma, sf_V, sf_M, lmbd, T,
                  use_stopping_time=False):
    """Constants follow experiment_settings/settings.py:215-282."""
    consts = {}
    dev = "cpu"
    if setting in ("OU_quadratic_easy", "OU_quadratic_hard"):
        if d == 2:
            x0 = torch.tensor([0.4, 0.6])
        else:
            x0 = 0.5 * torch.randn(d)
        sigma = torch.eye(d)
        c = (1.0, 1.0, 0.5) if setting == "OU_quadratic_hard" else (0.2, 0.2, 0.1)
        A, P, Q = c[0] * torch.eye(d), c[1] * torch.eye(d), c[2] * torch.eye(d)
        sde = classes["OU_Quadratic"](
            device=dev, dim=d, hdims=hdims, hdims_M=hdims_M, lmbd=lmbd, A=A, P=P, Q=Q,
            sigma=sigma, gamma=gamma, scaling_factor_nabla_V=sf_V, scaling_factor_M=sf_M,
            u_warm_start=None, use_warm_start=False,
        )
        consts.update(A=A, P=P, Q=Q)
    elif setting == "OU_quadratic_dense":
        # not a reference preset: dense non-symmetric A/P/Q/sigma through the
        # reference's OU_Quadratic class, to pin index conventions (transposes).
        x0 = 0.5 * torch.randn(d)
        sigma = torch.eye(d) + 0.2 * torch.randn(d, d)
        A = -0.5 * torch.eye(d) + 0.3 * torch.randn(d, d)
        P = 0.3 * torch.randn(d, d)
        Q = 0.3 * torch.randn(d, d)
        sde = classes["OU_Quadratic"](
            device=dev, dim=d, hdims=hdims, hdims_M=hdims_M, lmbd=lmbd, A=A, P=P, Q=Q,
            sigma=sigma, gamma=gamma, scaling_factor_nabla_V=sf_V, scaling_factor_M=sf_M,
            u_warm_start=None, use_warm_start=False,
        )
        consts.update(A=A, P=P, Q=Q)
    elif setting == "OU_linear":
        x0 = torch.zeros(d)
        xi = 0.1 * torch.randn(d, d)
        omega = torch.ones(d)
        A = -torch.eye(d) + xi
        sigma = torch.eye(d) + xi
        sde = classes["OU_Linear"](
            device=dev, dim=d, hdims=hdims, hdims_M=hdims_M, lmbd=lmbd, A=A, omega=omega,
            sigma=sigma, gamma=gamma, scaling_factor_nabla_V=sf_V, scaling_factor_M=sf_M,
        )
        consts.update(A=A, omega=omega)
    elif setting == "double_well":
        x0 = torch.zeros(d)
        kappa = torch.ones(d)
        nu = torch.ones(d)
        kappa[:3] = 5
        nu[:3] = 3
        sigma = torch.eye(d)
        sde = classes["DoubleWell"](
            device=dev, dim=d, hdims=hdims, hdims_M=hdims_M, lmbd=lmbd, kappa=kappa, nu=nu,
            sigma=sigma, gamma=gamma, scaling_factor_nabla_V=sf_V, scaling_factor_M=sf_M,
        )
        consts.update(kappa=kappa, nu=nu)
    elif setting == "molecular_dynamics":
        x0 = -torch.ones(d)
        kappa = torch.ones(d)
        sigma = torch.eye(d)
        sde = classes["MolecularDynamics"](
            device=dev, dim=d, hdims=hdims, hdims_M=hdims_M, lmbd=lmbd, kappa=kappa,
            sigma=sigma, gamma=gamma, scaling_factor_nabla_V=sf_V, scaling_factor_M=sf_M,
            T=T, use_stopping_time=use_stopping_time,
        )
        consts.update(kappa=kappa)
    else:
        raise ValueError(setting)
    sde.initialize_models()
    consts.update(x0=x0, sigma=sigma)
    return sde, x0, sigma, consts


def make_one(name, setting, d, K, B, hdims, hdims_M, gamma, seed, T=1.0, lmbd=1.0,
             sf_V=1.0, sf_M=0.1, use_stopping_time=False, with_loss=True, with_pairs=True, with_algs=False):
    utils, method, models, classes = _import_reference()
    torch.manual_seed(seed)
    torch.set_num_threads(1)
    sde, x0, sigma, consts = build_setting(
        classes, setting, d, hdims, hdims_M, gamma, sf_V, sf_M, lmbd, T, use_stopping_time
    )
    ts = torch.linspace(0, T, K + 1)
    noise = torch.randn(K, B, d)
    out = {}
    out["meta_setting"] = np.array(setting)
    out["meta"] = np.array([d, K, B, seed, int(use_stopping_time)], dtype=np.int64)
    out["meta_f"] = np.array([T, lmbd, gamma, sf_V, sf_M], dtype=np.float64)
    out["hdims"] = np.array(hdims, dtype=np.int64)
    out["hdims_M"] = np.array(hdims_M, dtype=np.int64)
    for k, v in consts.items():
        out["const_" + k] = v.numpy().copy()
    for k, v in sde.nabla_V.state_dict().items():
        out["nablaV." + k] = v.detach().numpy().copy()
    for k, v in sde.M.state_dict().items():
        out["M." + k] = v.detach().numpy().copy()
    out["gamma"] = sde.gamma.detach().numpy().copy()
    if use_stopping_time:
        out["gamma2"] = sde.gamma2.detach().numpy().copy()
        out["gamma3"] = sde.gamma3.detach().numpy().copy()
    out["ts"] = ts.numpy().copy()
    out["noise_in"] = noise.numpy().copy()

    # ---- rollout: utils.py:17-128 --------------------------------------
    state0 = x0.repeat(B, 1)
    with _NoiseFeeder(noise):
        with torch.no_grad():
            r = utils.stochastic_trajectories(sde, state0, ts, lmbd, detach=True)
    names = ["states", "noises", "stop_indicators", "fractional_timesteps",
             "lpd", "lps", "ltw", "controls"]
    for n, v in zip(names, r):
        out["roll_" + n] = v.to(torch.float32).numpy().copy()
    assert np.array_equal(out["roll_noises"], out["noise_in"])

    if with_loss:
        # ---- SOCM loss + gradients: method.py:223-906 ------------------
        solver = method.SOC_Solver(sde, x0, None, T=T, num_steps=K, lmbd=lmbd, d=d, sigma=sigma)
        with _NoiseFeeder(noise):
            res = solver.loss(
                B, compute_L2_error=False, optimal_control=None,
                compute_control_objective=False, algorithm="SOCM", verbose=False,
                u_warm_start=None, use_warm_start=False, use_stopping_time=use_stopping_time,
            )
        objective, _, _, _, _, w_mean, w_std, stop_ind = res
        objective.backward()
        out["loss_objective"] = objective.detach().numpy().copy()
        out["loss_weight_mean"] = w_mean.detach().numpy().copy()
        out["loss_weight_std"] = w_std.detach().numpy().copy()
        for k, p in sde.nabla_V.named_parameters():
            out["grad_nablaV." + k] = p.grad.numpy().copy()
        for k, p in sde.M.sigmoid_layers.named_parameters():
            out["grad_M.sigmoid_layers." + k] = p.grad.numpy().copy()
        out["grad_gamma"] = sde.gamma.grad.numpy().copy()
        if use_stopping_time:
            out["grad_gamma2"] = sde.gamma2.grad.numpy().copy()
            out["grad_gamma3"] = (
                sde.gamma3.grad.numpy().copy() if sde.gamma3.grad is not None else np.zeros(1, np.float32)
            )

        if with_pairs and not use_stopping_time:
            # ---- M and dM/ds on the pair grid: method.py:510-515, 533-573
            import functorch

            s_vec, t_vec = [], []
            for k, t in enumerate(ts):
                s_vec.append(torch.linspace(t, T, K + 1 - k))
                t_vec.append(t * torch.ones(K + 1 - k))
            s_vec, t_vec = torch.cat(s_vec), torch.cat(t_vec)
            with torch.no_grad():
                M_all = sde.M(t_vec, s_vec)
            sum_M = lambda t, s: sde.M(t, s).sum(dim=0)
            dM0 = functorch.jacrev(sum_M, argnums=1)
            dM_all = torch.transpose(torch.transpose(dM0(t_vec, s_vec), 1, 2), 0, 1)
            out["pairs_t"] = t_vec.numpy().copy()
            out["pairs_s"] = s_vec.numpy().copy()
            out["pairs_M"] = M_all.detach().numpy().copy()
            out["pairs_dM"] = dM_all.detach().numpy().copy()

    if with_algs:
        # ---- the other losses on the same rollout (method.py:264-270, 289-478, 722-856): objective + nabla_V grads
        for alg in ("SOCM_const_M", "SOCM_exp", "SOCM_adjoint", "cross_entropy", "log-variance", "variance",
                    "moment", "rel_entropy"):
            for p_ in sde.parameters():
                p_.grad = None
            solver = method.SOC_Solver(sde, x0, None, T=T, num_steps=K, lmbd=lmbd, d=d, sigma=sigma)
            with torch.no_grad():
                solver.y0.fill_(0.37)
            solver.gamma = torch.nn.Parameter(torch.tensor([gamma])) if alg == "SOCM_exp" else gamma
            with _NoiseFeeder(noise):
                res = solver.loss(B, compute_L2_error=False, optimal_control=None, compute_control_objective=False,
                                  algorithm=alg, verbose=False, u_warm_start=None, use_warm_start=False,
                                  use_stopping_time=False)
            res[0].backward()
            out[f"alg.{alg}.objective"] = res[0].detach().numpy().copy()
            for k, p_ in sde.nabla_V.named_parameters():
                out[f"alg.{alg}.grad_nablaV.{k}"] = p_.grad.numpy().copy()
            if alg == "SOCM_exp":
                out[f"alg.{alg}.grad_gamma"] = solver.gamma.grad.numpy().copy()
            if alg == "moment":
                out[f"alg.{alg}.grad_y0"] = solver.y0.grad.numpy().copy()

    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {os.path.getsize(path)/1024:.1f} KiB"
          + (f"  objective={float(out['loss_objective']):.6g}" if with_loss else ""))


TINY = dict(hdims=[32, 16, 8], hdims_M=[16, 16])
DEFAULT = dict(hdims=[256, 128, 64], hdims_M=[128, 128])

FIXTURES = [
    # name, setting, d, K, B, arch, gamma, seed, extra
    ("tiny_ou_quadratic_easy_d2", "OU_quadratic_easy", 2, 12, 8, TINY, 2.0, 1, {}),
    ("tiny_ou_quadratic_hard_d4", "OU_quadratic_hard", 4, 12, 8, TINY, 1.0, 2, dict(with_algs=True)),
    ("tiny_ou_quadratic_dense_d3", "OU_quadratic_dense", 3, 10, 8, TINY, 1.5, 7, {}),
    ("tiny_ou_linear_d6", "OU_linear", 6, 14, 8, TINY, 2.0, 3, dict(with_algs=True)),
    ("tiny_double_well_d10", "double_well", 10, 40, 8, TINY, 6.0, 4, dict(with_algs=True)),
    ("tiny_molecular_dynamics_d1", "molecular_dynamics", 1, 24, 16, TINY, 1.0, 5,
     dict(T=2.0, lmbd=2.0)),
    ("tiny_molecular_dynamics_d1_stopping", "molecular_dynamics", 1, 24, 16, TINY, 1.0, 5,
     dict(T=2.0, lmbd=2.0, use_stopping_time=True)),
    ("tiny_molecular_dynamics_d2", "molecular_dynamics", 2, 20, 16, TINY, 1.0, 6, {}),
    # d = 2 with the stopping-time loss: per-sample 2 x 2 pair matrices (TwoBoundarySigmoidMLP), off-diagonal terms
    ("tiny_molecular_dynamics_d2_stopping", "molecular_dynamics", 2, 20, 16, TINY, 1.0, 6,
     dict(T=2.0, lmbd=2.0, use_stopping_time=True)),
    # d = 5 and d = 10 with stopping times: the entry-wise (8- and 16-wide) instantiations of the stopping-time kernels, B = 66
    # rows (two 64-sample blocks, ragged)
    ("tiny_molecular_dynamics_d5_stopping", "molecular_dynamics", 5, 14, 66, TINY, 1.0, 11,
     dict(T=1.2, lmbd=2.0, use_stopping_time=True)),
    ("tiny_molecular_dynamics_d10_stopping", "molecular_dynamics", 10, 8, 12, TINY, 1.5, 12,
     dict(T=1.5, lmbd=1.0, use_stopping_time=True)),
    # default architecture at the BASELINE configs, small batch (weights dominate the size)
    ("cfg1_ou_quadratic_easy_d2_K50", "OU_quadratic_easy", 2, 50, 8, DEFAULT, 2.0, 0,
     dict(with_pairs=False)),
    ("cfg3_double_well_d10_K200", "double_well", 10, 200, 8, DEFAULT, 6.0, 0,
     dict(with_pairs=False)),
    # a 20-row batch so that a 16-row tile plus a ragged 4-row tail is exercised
    ("tiny_ou_linear_d5_B20", "OU_linear", 5, 9, 20, TINY, 2.0, 8, {}),
    # d > 16 (BASELINE configs[4] is d = 64): dense sigma, several 16-wide blocks per pair matrix, the general SDE step
    ("tiny_ou_linear_d20", "OU_linear", 20, 8, 8, TINY, 2.0, 9, dict(with_pairs=False)),
    ("tiny_ou_linear_d64", "OU_linear", 64, 5, 4, dict(hdims=[32, 16, 8], hdims_M=[8, 8]), 2.0, 10,
     dict(with_pairs=False)),
    # DEFAULT widths at the dimensions that select the other two constexpr rollout instantiations and the LDS-staged
    # contraction kernels: BASELINE configs[4] (OU_linear d = 64 -> StaticNet<80,256,128,64,64>) and soc.yaml's own
    # default d = 20 (OU_quadratic_easy -> StaticNet<32,256,128,64,32>)
    ("cfg5_ou_linear_d64_K20", "OU_linear", 64, 20, 8, DEFAULT, 2.0, 0, dict(with_pairs=False)),
    ("ouq20_ou_quadratic_easy_d20_K12", "OU_quadratic_easy", 20, 12, 8, DEFAULT, 2.0, 0, dict(with_pairs=False)),
    # LARGE BATCHES (round 3): the contraction kernels the BASELINE configs select by batch size -- d = 64 at B >= 256 takes
    # the LDS-staged forward (socm_target_lds4_kernel) and the transposed-tile backward (socm_target_bwd_lds2_kernel);
    # d <= 16 at B >= 512 the two-pairs-per-wave backward.  K is small so that the reference's own (Kp,Kp,B,d,d)
    # intermediates stay at 67 MB (d = 64) / 10 MB (d = 10); T is shortened with it (time steps of 0.05 / 0.01 as at
    # the other fixtures: with T = 1 the six-step double well diverges to NaN in the reference itself).
    ("cfg5_ou_linear_d64_B256_K3", "OU_linear", 64, 3, 256, DEFAULT, 2.0, 0, dict(with_pairs=False, T=0.15)),
    ("cfg4_double_well_d10_B512_K6", "double_well", 10, 6, 512, DEFAULT, 6.0, 0, dict(with_pairs=False, T=0.06)),
    # FULL-SIZE PINS (round 4).  (1) The headline config at its own size -- README.md:51 / BASELINE configs[2]: double_well
    # d = 10, K = 200, B = 128, default widths (the reference holds ~10.6 GB of (Kp,Kp,B,d,d) intermediates and needs a few
    # minutes on one thread): the 128-workgroup launch (one row per workgroup) bench.py times faces the reference itself.
    ("cfg3_full_double_well_d10_K200_B128", "double_well", 10, 200, 128, DEFAULT, 6.0, 0, dict(with_pairs=False)),
    # (2) README.md:60's molecular_dynamics run as written: d = 1, K = 150, B = 64, default control-network widths,
    # arch.hdims_M=[64,64], gamma = 2 (gamma2 = gamma3 = 1: MolecularDynamics does not forward them, method.py:29-30),
    # use_stopping_time=True -- the constexpr STOPPING rollout instantiations and stopping_target_kernel<1> at K = 150.
    ("md_default_d1_K150_B64_stopping", "molecular_dynamics", 1, 150, 64,
     dict(hdims=[256, 128, 64], hdims_M=[64, 64]), 2.0, 0, dict(use_stopping_time=True)),
    # (3) the other eight losses at the DEFAULT widths and K = 200 (SOCM_adjoint's costate recursion over 200 steps, the
    # HIP rollout -> socmx_baselines kernels -> constexpr control-network backward): same seed and draws as
    # cfg3_double_well_d10_K200 (only the alg.* outputs are kept next to the inputs)
    ("cfg3_algs_double_well_d10_K200", "double_well", 10, 200, 8, DEFAULT, 6.0, 0,
     dict(with_pairs=False, with_loss=False, with_algs=True)),
    # (4) BASELINE configs[1] at its own size (README.md:15: OU_quadratic_easy d = 2, K = 50, B = 128): the OU form of the one-row
    # rollout kernel (A x drift, x'Px running cost) and the whole loss at the batch bench.py's secondary entry times
    ("cfg1_full_ou_quadratic_easy_d2_K50_B128", "OU_quadratic_easy", 2, 50, 128, DEFAULT, 2.0, 0, dict(with_pairs=False)),
    # (5) the README's Linear OU run at its own size (OU_linear d = 10, K = 100, B = 64, default widths): a DENSE sigma at
    # d <= 15 -- the dense forms of the one-row rollout kernel (u = -sigma^T nabla_V, sigma u, sigma eps on the serial chain) and,
    # through the tests that re-launch its rows in larger batches, of the 4-row / 16-row / two-tile kernels
    ("oul10_ou_linear_d10_K100_B64", "OU_linear", 10, 100, 64, DEFAULT, 2.0, 0, dict(with_pairs=False)),
    # (6) d % 4 != 0 beyond d = 22 (default widths): rows of d*d = 900 floats in the pair-grid network's WIDE kernels (56 whole
    # 16-wide blocks + one quad), several 16-wide blocks per pair matrix in the contraction, the 32-wide rollout instantiation
    ("oul30_ou_linear_d30_K10_B16", "OU_linear", 30, 10, 16, DEFAULT, 2.0, 0, dict(with_pairs=False)),
    # (7) the eight other losses at the DEFAULT widths in the README's two other sweep settings (README.md:15-45 run every
    # algorithm at each setting): Quadratic OU at d = 20 (the 32-wide instantiations, A / P matrices in the costate recursion) and
    # Linear OU at d = 10 (dense sigma)
    ("ouq20_algs_ou_quadratic_easy_d20_K12", "OU_quadratic_easy", 20, 12, 8, DEFAULT, 2.0, 0,
     dict(with_pairs=False, with_loss=False, with_algs=True)),
    ("oul10_algs_ou_linear_d10_K20", "OU_linear", 10, 20, 8, DEFAULT, 2.0, 0,
     dict(with_pairs=False, with_loss=False, with_algs=True)),
]


if __name__ == "__main__":
    only = sys.argv[1:]
    for name, setting, d, K, B, arch, gamma, seed, extra in FIXTURES:
        if only and name not in only:
            continue
        make_one(name, setting, d, K, B, arch["hdims"], arch["hdims_M"], gamma, seed, **extra)
