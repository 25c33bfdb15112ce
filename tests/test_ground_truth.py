"""Row f2: rollouts under the reference's ground-truth controls, pinned to reference-generated fixtures
(tests/golden/make_golden_ground_truth.py: the unmodified reference's `ground_truth_control` +
`stochastic_trajectories` + `control_objective` on injected noise).

CPU: the product's own control tables (socmx.ground_truth: Riccati / expm / vectorised PDE) against the reference's tables,
and the device-agnostic eager rollout against the reference's outputs.
GPU (`-m gpu`): ONE launch of socmx_rollout_control_f32 (csrc/socmx_rollout_ctrl.hip) against the same outputs.
"""
import glob
import os
import types

import numpy as np
import pytest
import torch

from test_host_cpu import GOLDEN

GT = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "gt_*.npz")))
NAMES = ["states", "noises", "stop_indicators", "fractional_timesteps", "lpd", "lps", "ltw", "controls"]


def build_optimal_sde(name, device="cpu", table="reference"):
    """The product's un-learned setting object with `u` = the product's control class; `table`: "reference" feeds it the
    reference's table from the fixture, "product" lets socmx.ground_truth compute its own."""
    from SOC_matching.experiment_settings.OU_quadratic import OU_Quadratic
    from SOC_matching.experiment_settings.OU_linear import OU_Linear
    from SOC_matching.experiment_settings.double_well import DoubleWell
    from socmx import ground_truth as G
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    setting = str(z["meta_setting"])
    d, K, B, seed, n_batches = [int(v) for v in z["meta"]]
    T, lmbd, delta_t, delta_x, xb = [float(v) for v in z["meta_f"]]
    c = lambda k: torch.from_numpy(z["const_" + k].copy()).to(device)
    ts = torch.from_numpy(z["ts"].copy()).to(device)
    ref_table = torch.from_numpy(z["ut"].copy()).to(device)
    cfg = types.SimpleNamespace(method=types.SimpleNamespace(device=device, d=d, T=T, lmbd=lmbd, delta_t_optimal=delta_t,
                                                             delta_x_optimal=delta_x))
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        if setting.startswith("OU_quadratic"):
            sde = G.lq_optimal_sde(OU_Quadratic, ts, c("sigma"), c("A"), c("P"), c("Q"), cfg)
            own = sde.u.U
            if table == "reference":
                sde.u = G.LinearControl(ref_table, ts, T)
        elif setting == "OU_linear":
            sde = G.linear_optimal_sde(OU_Linear, ts, c("sigma"), c("A"), c("omega"), cfg)
            own = sde.u.C
            if table == "reference":
                sde.u = G.ConstantControl(ref_table, ts, T)
        else:
            sde = G.double_well_optimal_sde(DoubleWell, c("kappa"), c("nu"), c("sigma"), cfg, xb=xb)
            own = sde.u.ut
            if table == "reference":
                sde.u = G.LowDimControl(ref_table, T, xb, d, delta_t, delta_x)
    aux = dict(z=z, d=d, K=K, B=B, T=T, lmbd=lmbd, ts=ts, x0=c("x0"), n_batches=n_batches, own_table=own,
               ref_table=ref_table, setting=setting,
               state0=torch.from_numpy(z["state0"].copy()).to(device),
               noise=torch.from_numpy(z["noise_in"].copy()).to(device))
    return sde, aux


def _burst_noise(z, K, n_batches, device):
    """The reference consumes the burst noise batch after batch (K draws each); the product's burst takes
    (K, n_batches * batch, d) with batch k in columns [k*batch, (k+1)*batch)."""
    bn = torch.from_numpy(z["burst_noise"].copy())
    return torch.cat([bn[k * K:(k + 1) * K] for k in range(n_batches)], dim=1).to(device)


def _check_rollout(got, z, rtol=2e-4):
    for n, a in zip(NAMES, got):
        want = z["roll_" + n]
        assert tuple(a.shape) == want.shape, n
        scale = max(1.0, float(np.abs(want).max()))
        np.testing.assert_allclose(a.detach().to("cpu", torch.float32).numpy(), want, rtol=rtol, atol=rtol * scale,
                                   err_msg=n)


@pytest.mark.parametrize("name", GT)
def test_product_control_tables_match_the_reference_tables(name):
    """socmx.ground_truth's own Riccati / matrix-exponential / PDE tables against the tables the reference built."""
    sde, aux = build_optimal_sde(name, table="product")
    own, ref = aux["own_table"].to(torch.float32).numpy(), aux["ref_table"].numpy()
    assert own.shape == ref.shape
    if aux["setting"] == "double_well":
        # the reference assembles its tridiagonal system in fp32-cast pieces; ours is fp64 throughout
        np.testing.assert_allclose(own, ref, rtol=2e-3, atol=2e-3 * np.abs(ref).max())
    else:
        np.testing.assert_allclose(own, ref, rtol=1e-5, atol=1e-6 * max(1.0, np.abs(ref).max()))


@pytest.mark.parametrize("name", GT)
def test_eager_ground_truth_rollout_matches_reference(name):
    from SOC_matching import utils
    sde, aux = build_optimal_sde(name)
    with torch.no_grad():
        got = utils.stochastic_trajectories(sde, aux["state0"], aux["ts"], aux["lmbd"], noise_in=aux["noise"])
    _check_rollout(got, aux["z"], rtol=1e-4)
    with torch.no_grad():
        u3 = sde.u(aux["ts"], got[0], t_is_tensor=True)
    np.testing.assert_allclose(u3.to(torch.float32).numpy(), aux["z"]["u_on_trajectory"], rtol=1e-4, atol=1e-4)
    z, K = aux["z"], aux["K"]
    Bb = int(z["burst_batch"])
    m, e = utils.control_objective(sde, aux["x0"], aux["ts"], aux["lmbd"], Bb, total_n_samples=Bb * aux["n_batches"],
                                   noise_in=_burst_noise(z, K, aux["n_batches"], "cpu"))
    np.testing.assert_allclose(float(m), float(z["burst_mean"]), rtol=1e-4)
    np.testing.assert_allclose(float(e), float(z["burst_std_err"]), rtol=1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("name", GT)
def test_ground_truth_rollout_kernel_vs_reference(name):
    """socmx_rollout_control_f32 (LinearControl / ConstantControlLinear / LowDimControl evaluated inside the kernel,
    models.py:10-150) against the reference's own optimal-SDE rollout, and the fused costs-only burst against the
    reference's control_objective (main.py:137-150)."""
    from SOC_matching import utils
    from socmx import rollout as R
    dev = "cuda:0"
    sde, aux = build_optimal_sde(name, dev)
    assert R._eligible_for_hip_control(sde, aux["state0"], True)
    got = utils.stochastic_trajectories(sde, aux["state0"], aux["ts"], aux["lmbd"], noise_in=aux["noise"])
    _check_rollout(got, aux["z"])
    z, K = aux["z"], aux["K"]
    Bb = int(z["burst_batch"])
    m, e = utils.control_objective(sde, aux["x0"], aux["ts"], aux["lmbd"], Bb, total_n_samples=Bb * aux["n_batches"],
                                   noise_in=_burst_noise(z, K, aux["n_batches"], dev), chunk_rows=2 * Bb - 3)
    np.testing.assert_allclose(float(m), float(z["burst_mean"]), rtol=2e-4)
    np.testing.assert_allclose(float(e), float(z["burst_std_err"]), rtol=1e-3)


@pytest.mark.gpu
def test_control_table_index_out_of_range_raises():
    """A time index past the table (a grid longer than the table) must fail loudly, not read out of bounds."""
    from socmx import rollout as R, ground_truth as G
    dev = "cuda:0"
    sde, aux = build_optimal_sde("gt_ou_linear_d10", dev)
    short = G.ConstantControl(aux["ref_table"][:5].contiguous(), aux["ts"], aux["T"])
    short.hip_descriptor = lambda ts, _h=sde.u.hip_descriptor: _h(ts)[:1] + (aux["ref_table"][:5].contiguous(),) + _h(ts)[2:]
    sde.u = short
    with pytest.raises(Exception):
        R.stochastic_trajectories(sde, aux["state0"], aux["ts"], aux["lmbd"], noise_in=aux["noise"])
        torch.cuda.synchronize()
