"""The oracle (oracle/socm_oracle.py) against the reference-generated golden vectors."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import socm_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ALL = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "*.npz"))
             if not os.path.basename(p).startswith(("dw_pde", "train_", "gt_")))
TINY = [n for n in ALL if n.startswith("tiny_")]
NON_STOPPING_LOSS = [n for n in TINY if not n.endswith("_stopping")]


def _close(a, b, rtol=1e-5, atol=1e-6):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


@pytest.mark.parametrize("name", ALL)
def test_rollout_matches_reference(name):
    torch.set_num_threads(1)
    pb, vp, mp, gamma, aux = O.load_fixture(os.path.join(GOLDEN, name + ".npz"))
    z = aux["z"]
    with torch.no_grad():
        r = O.stochastic_trajectories(pb, vp, aux["x0"].repeat(aux["B"], 1), aux["ts"], aux["lmbd"], aux["noise"])
    names = ["states", "noises", "stop_indicators", "fractional_timesteps", "lpd", "lps", "ltw", "controls"]
    for n, v in zip(names, r):
        assert v.shape == z["roll_" + n].shape, n
        # same op order as the reference => agreement at fp32 round-off
        _close(v, z["roll_" + n], rtol=2e-5, atol=2e-6)
    # exact for the integer-like outputs
    assert np.array_equal(r[2].numpy(), z["roll_stop_indicators"])


@pytest.mark.parametrize("name", NON_STOPPING_LOSS)
def test_pairs_M_and_dM(name):
    pb, vp, mp, gamma, aux = O.load_fixture(os.path.join(GOLDEN, name + ".npz"))
    z = aux["z"]
    if "pairs_t" not in z.files:
        pytest.skip("fixture generated without the pair-grid arrays (with_pairs=False)")
    t_vec, s_vec = O.pair_grid(aux["ts"], aux["T"], aux["K"])
    assert np.array_equal(t_vec.numpy(), z["pairs_t"])
    assert np.array_equal(s_vec.numpy(), z["pairs_s"])
    with torch.no_grad():
        M = O.sigmoid_mlp(mp, gamma, t_vec, s_vec, aux["d"])
        dM = O.dM_ds_analytic(mp, gamma, t_vec, s_vec, aux["d"])
    _close(M, z["pairs_M"], rtol=1e-5, atol=1e-6)
    _close(dM, z["pairs_dM"], rtol=1e-4, atol=2e-6)   # analytic vs the reference's jacrev
    # M(t,t) = I  (models.py:268-275)
    diag = np.isclose(z["pairs_t"], z["pairs_s"])
    _close(M[torch.from_numpy(diag)], np.broadcast_to(np.eye(aux["d"], dtype=np.float32), (diag.sum(), aux["d"], aux["d"])))


@pytest.mark.parametrize("name", NON_STOPPING_LOSS)
@pytest.mark.parametrize("derivative", ["jacrev", "analytic"])
def test_socm_loss_and_grads(name, derivative):
    torch.set_num_threads(1)
    pb, vp, mp, gamma, aux = O.load_fixture(os.path.join(GOLDEN, name + ".npz"), requires_grad=True)
    z = aux["z"]
    obj, wm, ws = O.socm_loss(pb, vp, mp, gamma, aux["x0"], aux["ts"], aux["T"], aux["lmbd"], aux["B"],
                              aux["noise"], derivative=derivative)
    _close(obj, z["loss_objective"], rtol=2e-5)
    _close(wm, z["loss_weight_mean"], rtol=2e-5)
    _close(ws, z["loss_weight_std"], rtol=2e-5)
    obj.backward()
    for k, p in vp.items():
        g = z["grad_nablaV." + k]
        _close(p.grad, g, rtol=1e-3, atol=1e-5 * max(1.0, np.abs(g).max()))
    for k, p in mp.items():
        g = z["grad_M." + k]
        _close(p.grad, g, rtol=1e-3, atol=1e-5 * max(1.0, np.abs(g).max()))
    _close(gamma.grad, z["grad_gamma"], rtol=1e-3, atol=1e-5 * max(1.0, np.abs(z["grad_gamma"]).max()))


@pytest.mark.parametrize("name", ["cfg1_ou_quadratic_easy_d2_K50", "cfg1_full_ou_quadratic_easy_d2_K50_B128",
                                  "ouq20_ou_quadratic_easy_d20_K12",
                                  "cfg5_ou_linear_d64_K20", "cfg5_ou_linear_d64_B256_K3",
                                  "cfg4_double_well_d10_B512_K6",
                                  # the README's Linear OU at its own size: dense sigma, d = 10, K = 100, B = 64
                                  "oul10_ou_linear_d10_K100_B64",
                                  # d = 30 (d*d = 900 outputs of the pair-grid network: not a multiple of 16)
                                  "oul30_ou_linear_d30_K10_B16"])
def test_socm_loss_default_arch(name):
    torch.set_num_threads(4)
    pb, vp, mp, gamma, aux = O.load_fixture(os.path.join(GOLDEN, name + ".npz"), requires_grad=True)
    z = aux["z"]
    obj, wm, ws = O.socm_loss(pb, vp, mp, gamma, aux["x0"], aux["ts"], aux["T"], aux["lmbd"], aux["B"],
                              aux["noise"], derivative="analytic")
    _close(obj, z["loss_objective"], rtol=5e-5)
    obj.backward()
    num = sum(float(((p.grad - torch.from_numpy(z["grad_nablaV." + k])) ** 2).sum()) for k, p in vp.items())
    den = sum(float((z["grad_nablaV." + k] ** 2).sum()) for k in vp)
    assert (num / den) ** 0.5 < 1e-3


@pytest.mark.parametrize("name", ["tiny_molecular_dynamics_d1_stopping", "tiny_molecular_dynamics_d2_stopping",
                                  "tiny_molecular_dynamics_d5_stopping", "tiny_molecular_dynamics_d10_stopping",
                                  # README.md:60 as written (default widths, hdims_M=[64,64], K = 150, B = 64)
                                  "md_default_d1_K150_B64_stopping"])
def test_socm_loss_stopping_time(name):
    torch.set_num_threads(4 if ("d5" in name or "md_default" in name) else 1)
    pb, vp, mp, gamma, aux = O.load_fixture(os.path.join(GOLDEN, name + ".npz"), requires_grad=True)
    z = aux["z"]
    obj, wm, ws = O.socm_loss_stopping(pb, vp, mp, gamma, aux["gamma2"], aux["gamma3"], aux["x0"], aux["ts"],
                                       aux["T"], aux["lmbd"], aux["B"], aux["noise"])
    _close(obj, z["loss_objective"], rtol=2e-5)
    obj.backward()
    _close(gamma.grad, z["grad_gamma"], rtol=1e-3, atol=1e-6)
    _close(aux["gamma2"].grad, z["grad_gamma2"], rtol=1e-3, atol=1e-6)
    for k, p in vp.items():
        g = z["grad_nablaV." + k]
        _close(p.grad, g, rtol=1e-3, atol=1e-5 * max(1.0, np.abs(g).max()))


def test_philox_known_answers():
    # Random123 kat_vectors: philox4x32-10
    out = O.philox4x32_10(np.array([0, 0, 0, 0]), np.array([0, 0]))
    assert [hex(int(v)) for v in out] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    out = O.philox4x32_10(np.array([0xFFFFFFFF] * 4), np.array([0xFFFFFFFF] * 2))
    assert [hex(int(v)) for v in out] == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]
    out = O.philox4x32_10(np.array([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344]),
                          np.array([0xA4093822, 0x299F31D0]))
    assert [hex(int(v)) for v in out] == ["0xd16cfe09", "0x94fdcceb", "0x5001e420", "0x24126ea1"]


def test_philox_normals_moments():
    x = np.concatenate([O.philox_normals(1234, 0, r, 0, 8) for r in range(4000)])
    assert abs(x.mean()) < 0.02 and abs(x.std() - 1) < 0.02
