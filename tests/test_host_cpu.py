"""Host-side logic on CPU: the drop-in surface (`SOC_matching.*` names) running the device-agnostic
torch path (BASELINE config 0, "plumbing, no GPU") against the reference-generated golden vectors."""
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
LOSS = [n for n in TINY if not n.endswith("_stopping")]


def build_sde(name, device="cpu"):
    """Instantiate the product's setting class from a fixture (weights loaded from the fixture)."""
    from SOC_matching.experiment_settings.OU_quadratic import OU_Quadratic
    from SOC_matching.experiment_settings.OU_linear import OU_Linear
    from SOC_matching.experiment_settings.double_well import DoubleWell
    from SOC_matching.experiment_settings.molecular_dynamics import MolecularDynamics

    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    setting = str(z["meta_setting"])
    d, K, B, seed, stopping = [int(v) for v in z["meta"]]
    T, lmbd, gamma, sfV, sfM = [float(v) for v in z["meta_f"][:5]]
    c = lambda k: torch.from_numpy(z["const_" + k].copy()).to(device)
    common = dict(device=device, dim=d, hdims=[int(h) for h in z["hdims"]], hdims_M=[int(h) for h in z["hdims_M"]],
                  lmbd=lmbd, sigma=c("sigma"), gamma=gamma, scaling_factor_nabla_V=sfV, scaling_factor_M=sfM)
    if setting.startswith("OU_quadratic"):
        sde = OU_Quadratic(A=c("A"), P=c("P"), Q=c("Q"), **common)
    elif setting == "OU_linear":
        sde = OU_Linear(A=c("A"), omega=c("omega"), **common)
    elif setting == "double_well":
        sde = DoubleWell(kappa=c("kappa"), nu=c("nu"), **common)
    else:
        sde = MolecularDynamics(kappa=c("kappa"), T=T, use_stopping_time=bool(stopping), **common)
    sde.initialize_models()
    sde.nabla_V.load_state_dict({k[len("nablaV."):]: torch.from_numpy(z[k].copy()) for k in z.files
                                 if k.startswith("nablaV.")})
    sde.M.sigmoid_layers.load_state_dict({k[len("M.sigmoid_layers."):]: torch.from_numpy(z[k].copy())
                                          for k in z.files if k.startswith("M.sigmoid_layers.")})
    with torch.no_grad():
        sde.gamma.copy_(torch.from_numpy(z["gamma"].copy()))
        if stopping:
            sde.gamma2.copy_(torch.from_numpy(z["gamma2"].copy()))
            sde.gamma3.copy_(torch.from_numpy(z["gamma3"].copy()))
    aux = dict(z=z, d=d, K=K, B=B, T=T, lmbd=lmbd, x0=c("x0"), ts=torch.from_numpy(z["ts"].copy()).to(device),
               noise=torch.from_numpy(z["noise_in"].copy()).to(device), stopping=bool(stopping))
    return sde, aux


@pytest.mark.parametrize("name", TINY)
def test_eager_rollout_matches_reference(name):
    from SOC_matching import utils
    sde, aux = build_sde(name)
    z = aux["z"]
    with torch.no_grad():
        r = utils.stochastic_trajectories(sde, aux["x0"].repeat(aux["B"], 1), aux["ts"], aux["lmbd"],
                                          noise_in=aux["noise"])
    names = ["states", "noises", "stop_indicators", "fractional_timesteps", "lpd", "lps", "ltw", "controls"]
    for n, v in zip(names, r):
        assert v.shape == z["roll_" + n].shape, n
        np.testing.assert_allclose(v.to(torch.float32).numpy(), z["roll_" + n], rtol=1e-4, atol=1e-5, err_msg=n)


@pytest.mark.parametrize("name", LOSS)
def test_socm_loss_matches_reference(name):
    from SOC_matching.method import SOC_Solver
    sde, aux = build_sde(name)
    z = aux["z"]
    solver = SOC_Solver(sde, aux["x0"], None, T=aux["T"], num_steps=aux["K"], lmbd=aux["lmbd"], d=aux["d"],
                        sigma=sde.sigma)
    solver.noise_in = aux["noise"]
    out = solver.loss(aux["B"], algorithm="SOCM", use_warm_start=False, use_stopping_time=False)
    obj, wm, ws = out[0], out[5], out[6]
    np.testing.assert_allclose(obj.item(), z["loss_objective"], rtol=1e-4)
    np.testing.assert_allclose(wm.item(), z["loss_weight_mean"], rtol=1e-4)
    np.testing.assert_allclose(ws.item(), z["loss_weight_std"], rtol=1e-4)
    assert out[7].shape == z["roll_stop_indicators"].shape
    obj.backward()
    for k, p in sde.nabla_V.named_parameters():
        g = z["grad_nablaV." + k]
        np.testing.assert_allclose(p.grad.numpy(), g, rtol=1e-3, atol=1e-5 * max(1.0, np.abs(g).max()), err_msg=k)
    for k, p in sde.M.sigmoid_layers.named_parameters():
        g = z["grad_M.sigmoid_layers." + k]
        np.testing.assert_allclose(p.grad.numpy(), g, rtol=1e-3, atol=1e-5 * max(1.0, np.abs(g).max()), err_msg=k)
    np.testing.assert_allclose(sde.gamma.grad.numpy(), z["grad_gamma"], rtol=1e-3,
                               atol=1e-5 * max(1.0, np.abs(z["grad_gamma"]).max()))


@pytest.mark.parametrize("name", LOSS)
def test_pair_times_match_reference_linspace(name):
    from socmx import loss as L
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    if "pairs_t" not in z.files:
        pytest.skip("fixture generated without the pair-grid arrays (with_pairs=False)")
    ts = torch.from_numpy(z["ts"].copy())
    t, s, ii, jj = L.pair_times(ts, float(z["meta_f"][0]), int(z["meta"][1]))
    assert np.array_equal(t.numpy(), z["pairs_t"])
    np.testing.assert_allclose(s.numpy(), z["pairs_s"], rtol=0, atol=1.2e-7)


def test_state_dict_keys_and_seeded_init_match_reference():
    """Same construction order => same RNG consumption => a reference seed reproduces the same weights."""
    name = "tiny_double_well_d10"
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    from SOC_matching.experiment_settings.double_well import DoubleWell
    d, K, B, seed, _ = [int(v) for v in z["meta"]]
    torch.manual_seed(seed)
    sde = DoubleWell(device="cpu", dim=d, hdims=[32, 16, 8], hdims_M=[16, 16], lmbd=1.0,
                     kappa=torch.from_numpy(z["const_kappa"].copy()), nu=torch.from_numpy(z["const_nu"].copy()),
                     sigma=torch.eye(d), gamma=6.0, scaling_factor_nabla_V=1.0, scaling_factor_M=0.1)
    sde.initialize_models()
    for k, v in sde.nabla_V.state_dict().items():
        assert np.array_equal(v.numpy(), z["nablaV." + k]), k
    for k, v in sde.M.state_dict().items():
        if k.startswith("sigmoid_layers"):
            assert np.array_equal(v.numpy(), z["M." + k]), k


def test_stopping_time_socm_loss_matches_reference():
    """molecular_dynamics with use_stopping_time=True (per-sample TwoBoundarySigmoidMLP), README's MD command."""
    from SOC_matching.method import SOC_Solver
    name = "tiny_molecular_dynamics_d1_stopping"
    sde, aux = build_sde(name)
    z = aux["z"]
    solver = SOC_Solver(sde, aux["x0"], None, T=aux["T"], num_steps=aux["K"], lmbd=aux["lmbd"], d=aux["d"],
                        sigma=sde.sigma)
    solver.noise_in = aux["noise"]
    out = solver.loss(aux["B"], algorithm="SOCM", use_warm_start=False, use_stopping_time=True)
    np.testing.assert_allclose(out[0].item(), z["loss_objective"], rtol=1e-4)
    out[0].backward()
    for k, p in sde.nabla_V.named_parameters():
        g = z["grad_nablaV." + k]
        np.testing.assert_allclose(p.grad.numpy(), g, rtol=1e-3, atol=1e-5 * max(1.0, np.abs(g).max()), err_msg=k)
    for k, p in sde.M.sigmoid_layers.named_parameters():
        g = z["grad_M.sigmoid_layers." + k]
        np.testing.assert_allclose(p.grad.numpy(), g, rtol=2e-3, atol=1e-5 * max(1.0, np.abs(g).max()), err_msg=k)
    np.testing.assert_allclose(sde.gamma.grad.numpy(), z["grad_gamma"], rtol=2e-3, atol=1e-6)
    np.testing.assert_allclose(sde.gamma2.grad.numpy(), z["grad_gamma2"], rtol=2e-3, atol=1e-6)


OTHER_ALGS = ("SOCM_const_M", "SOCM_exp", "SOCM_adjoint", "cross_entropy", "log-variance", "variance", "moment",
              "rel_entropy")


@pytest.mark.parametrize("name", ["tiny_ou_quadratic_hard_d4", "tiny_ou_linear_d6", "tiny_double_well_d10",
                                  # default widths, K = 200 (SOCM_adjoint's costate recursion over 200 steps)
                                  "cfg3_algs_double_well_d10_K200",
                                  # the README's two other sweep settings at default widths (d = 20; dense sigma at d = 10)
                                  "ouq20_algs_ou_quadratic_easy_d20_K12", "oul10_algs_ou_linear_d10_K20"])
@pytest.mark.parametrize("alg", OTHER_ALGS)
def test_other_losses_match_reference(name, alg):
    """Row f4: the reference's eight other losses, objective and nabla_V gradients, same injected noise."""
    from SOC_matching.method import SOC_Solver
    sde, aux = build_sde(name)
    z = aux["z"]
    solver = SOC_Solver(sde, aux["x0"], None, T=aux["T"], num_steps=aux["K"], lmbd=aux["lmbd"], d=aux["d"],
                        sigma=sde.sigma)
    with torch.no_grad():
        solver.y0.fill_(0.37)
    gamma0 = float(z["meta_f"][2])
    solver.gamma = torch.nn.Parameter(torch.tensor([gamma0])) if alg == "SOCM_exp" else gamma0
    solver.noise_in = aux["noise"]
    out = solver.loss(aux["B"], algorithm=alg, use_warm_start=False, use_stopping_time=False)
    want = z[f"alg.{alg}.objective"]
    np.testing.assert_allclose(out[0].item(), want, rtol=2e-4, atol=1e-7)
    out[0].backward()
    num = den = 0.0
    for k, p in sde.nabla_V.named_parameters():
        g = z[f"alg.{alg}.grad_nablaV.{k}"]
        num += float(((p.grad.numpy() - g) ** 2).sum())
        den += float((g ** 2).sum())
    assert (num / max(den, 1e-30)) ** 0.5 < 2e-3, (alg, (num / max(den, 1e-30)) ** 0.5)
    if alg == "SOCM_exp":
        np.testing.assert_allclose(solver.gamma.grad.numpy(), z[f"alg.{alg}.grad_gamma"], rtol=2e-3, atol=1e-6)
    if alg == "moment":
        np.testing.assert_allclose(solver.y0.grad.numpy(), z[f"alg.{alg}.grad_y0"], rtol=2e-3)


def test_unknown_algorithm_fails_loudly():
    from SOC_matching.method import SOC_Solver
    sde, aux = build_sde("tiny_ou_quadratic_easy_d2")
    solver = SOC_Solver(sde, aux["x0"], None, T=aux["T"], num_steps=aux["K"], lmbd=aux["lmbd"], d=aux["d"],
                        sigma=sde.sigma)
    with pytest.raises(NotImplementedError):
        solver.loss(4, algorithm="SOCM_typo")


def test_trainer_gradient_telemetry_follows_the_reference_bookkeeping():
    """Trainer.step reports grad_norm_sqd / EMA_grad_norm_sqd / sqd_norm_EMA_grad exactly as main.py:325-345 computes
    them (per-parameter norms, compute_EMA on every gradient), here re-derived with plain per-tensor ops."""
    from SOC_matching.method import SOC_Solver
    from socmx.train import Trainer, make_optimizer, compute_EMA
    sde, aux = build_sde("tiny_ou_linear_d6", "cpu")
    solver = SOC_Solver(sde, aux["x0"], None, T=aux["T"], num_steps=aux["K"], lmbd=aux["lmbd"], d=aux["d"], sigma=sde.sigma)
    opt = make_optimizer(solver, M_lr=1e-3)
    tr = Trainer(solver, opt, aux["B"], sync_timing=False)
    ema_grad = ema_norm = None
    for itr in range(4):
        seen = {}
        hook = opt.register_step_pre_hook(lambda *_: seen.update(
            g=[p.grad.detach().clone() for p in sde.nabla_V.parameters()]))
        solver.noise_in = torch.randn(aux["K"], aux["B"], aux["d"], generator=torch.Generator().manual_seed(itr))
        step = tr.step()
        hook.remove()
        grad = seen["g"]
        gns = sum(torch.norm(g) ** 2 for g in grad)
        if itr == 0:
            ema_grad, ema_norm = grad, gns
        else:
            ema_grad = [compute_EMA(g, e, EMA_coeff=0.01, itr=itr) for g, e in zip(grad, ema_grad)]
            ema_norm = compute_EMA(gns, ema_norm, EMA_coeff=0.01, itr=itr)
        np.testing.assert_allclose(step["grad_norm_sqd"].item(), gns.item(), rtol=1e-5)
        np.testing.assert_allclose(step["EMA_grad_norm_sqd"].item(), ema_norm.item(), rtol=1e-5)
        np.testing.assert_allclose(step["sqd_norm_EMA_grad"].item(), sum(torch.norm(e) ** 2 for e in ema_grad).item(), rtol=1e-5)


TRAIN = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "train_*.npz")))


def run_training_fixture(name, device, **trainer_kw):
    """Replays a reference-generated training fixture (tests/golden/make_golden_train.py) through Trainer.step."""
    from SOC_matching.method import SOC_Solver
    from socmx.train import Trainer, make_optimizer
    sde, aux = build_sde(name, device)
    z = aux["z"]
    lr_V, lr_M, eps, norm0 = [float(v) for v in z["meta_f"][5:9]]
    solver = SOC_Solver(sde, aux["x0"], None, T=aux["T"], num_steps=aux["K"], lmbd=aux["lmbd"], d=aux["d"],
                        sigma=sde.sigma)
    opt = make_optimizer(solver, nabla_V_lr=lr_V, M_lr=lr_M, adam_eps=eps)
    tr = Trainer(solver, opt, aux["B"], normalization_const=norm0, sync_timing=False, **trainer_kw)
    rec = dict(loss=[], weight_mean=[], norm=[])
    static = torch.empty_like(aux["noise"][0]) if trainer_kw.get("hip_graph") else None   # a captured graph reads ONE buffer
    for it in range(int(z["train_iters"])):
        if static is not None:
            static.copy_(aux["noise"][it])
        solver.noise_in = aux["noise"][it] if static is None else static
        out = tr.step()
        rec["loss"].append(float(out["loss"]))
        rec["weight_mean"].append(float(out["weight_mean"]))
        rec["norm"].append(float(tr.normalization_const))
    tr.join()
    return sde, z, rec


def check_training_fixture(sde, z, rec, rtol):
    np.testing.assert_allclose(rec["loss"], z["train_loss"], rtol=rtol)
    np.testing.assert_allclose(rec["weight_mean"], z["train_weight_mean"], rtol=rtol)
    np.testing.assert_allclose(rec["norm"], z["train_norm_const"], rtol=rtol)
    num = den = 0.0
    for prefix, mod in (("final_nablaV.", sde.nabla_V), ("final_M.", sde.M)):
        for k, v in mod.state_dict().items():
            w = z[prefix + k]
            num += float(((v.detach().cpu().numpy() - w) ** 2).sum())
            den += float((w ** 2).sum())
    assert (num / den) ** 0.5 < 10 * rtol, (num / den) ** 0.5
    np.testing.assert_allclose(sde.gamma.detach().cpu().numpy(), z["final_gamma"], rtol=10 * rtol)


@pytest.mark.parametrize("name", TRAIN)
def test_training_iterations_match_reference_on_cpu(name):
    """main.py:280-359 replayed: the same losses, importance-weight means, normaliser recursion and final
    parameters as the reference's own loop (SOC_Solver.loss -> /normaliser -> backward -> Adam groups -> EMA)."""
    torch.set_num_threads(1)
    sde, z, rec = run_training_fixture(name, "cpu")
    check_training_fixture(sde, z, rec, rtol=2e-4)
