"""The SOCM iteration as one replayed hipGraph (Trainer(hip_graph=True)): device-resident Philox key, iteration counter,
EMA normaliser and Adam step counters.  Checked against the eager Trainer on the same Philox stream, and against the
reference-generated training fixtures (injected noise through a static buffer)."""
import os

import numpy as np
import pytest
import torch

from test_host_cpu import build_sde, GOLDEN

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _np(t):
    return t.detach().to("cpu", torch.float32).numpy()


def _run(name, graph, steps, B, seed=77, save_activations=True):
    from SOC_matching.method import SOC_Solver
    from socmx.rollout import PhiloxKey
    from socmx.train import Trainer, make_optimizer
    sde, aux = build_sde(name, DEV)
    solver = SOC_Solver(sde, aux["x0"], None, T=aux["T"], num_steps=aux["K"], lmbd=aux["lmbd"], d=aux["d"], sigma=sde.sigma)
    solver.philox_key = PhiloxKey(torch.device(DEV), seed=seed, offset=5)
    opt = make_optimizer(solver, M_lr=1e-3)
    tr = Trainer(solver, opt, B, normalization_const=0.7, sync_timing=False, hip_graph=graph, graph_warmup=2,
                 overlap_M_backward=False, save_activations=save_activations)
    rec = []
    for _ in range(steps):
        info = tr.step()
        rec.append([float(info[k]) for k in ("loss", "weight_mean", "weight_std", "grad_norm_sqd", "EMA_grad_norm_sqd",
                                             "sqd_norm_EMA_grad")] + [float(tr.normalization_const)])
    tr.join()                 # (hipGraph mode: applies the pair-grid network's outstanding update)
    torch.cuda.synchronize()
    return np.array(rec), {k: _np(v) for k, v in sde.state_dict().items()}, tr, solver


@pytest.mark.parametrize("name,B", [("cfg1_ou_quadratic_easy_d2_K50", 128), ("cfg3_double_well_d10_K200", 64),
                                    ("cfg3_full_double_well_d10_K200_B128", 128),
                                    ("cfg1_full_ou_quadratic_easy_d2_K50_B128", 128),
                                    ("oul10_ou_linear_d10_K100_B64", 64),      # dense sigma: the one-row kernel's dense form
                                    ("ouq20_ou_quadratic_easy_d20_K12", 40), ("tiny_molecular_dynamics_d2_stopping", 48),
                                    ("oul30_ou_linear_d30_K10_B16", 16)])      # pair matrices of 900 floats: the wide pair-net kernels
def test_graph_replay_equals_the_eager_iteration(name, B):
    """7 iterations = 2 eager warm-up + the captured one + 4 replays, fresh Philox noise in every one; the eager Trainer on
    the same device key is the reference: losses, weight statistics, gradient telemetry, normaliser and the final
    parameters."""
    # (save_activations=False on both sides: the re-computing backward.  The default -- the rollout saves the control network's activations
    #  where its one-row kernel applies, in the eager autograd iteration and in the replayed body alike -- is compared below, eager against
    #  replayed at the same tolerances; between the two FORMS the weight gradients see another kernel's fp32 rounding of the same activations
    #  and a handful of ReLU signs at rounding distance of zero, which test_gpu_saved.py bounds)
    rec_e, par_e, _, _ = _run(name, False, 7, B, save_activations=False)
    rec_g, par_g, tr, solver = _run(name, True, 7, B, save_activations=False)
    assert tr.hip_graph and len([k for k in tr._graphs if not (isinstance(k, tuple) and k and k[0] == "warm")]) == 1
    if "molecular" not in name:
        assert any(k and k[0] == "manual" for k in tr._graphs)   # the autograd-free body, deferred pair-grid-network update
    np.testing.assert_allclose(rec_g, rec_e, rtol=2e-5, atol=1e-7)
    assert len(set(np.round(rec_g[:, 0], 9))) == 7                    # every replay drew fresh noise
    assert solver.philox_key.key.cpu().tolist()[1] == 5 + 7
    for k in par_e:
        # (the replayed body steps the control network with socmx_adam_step_f32, the eager one with torch's fused Adam: the
        #  same rule, last-bit differences in the bias corrections -- a few 1e-7 on parameters of size 1e-2 after 7 steps)
        #  At the headline size (B = 128: 25,728 rows per weight gradient) a handful of entries whose gradient sums nearly
        #  cancel differ by a few 1e-6 after seven Adam steps (eps = 1e-4 divides a last-bit difference of the sum by 1e-4):
        #  element-wise 1e-5 there (and at K = 100, B = 64: 6,464 rows), and norm-wise agreement to 2e-6 everywhere.
        np.testing.assert_allclose(par_g[k], par_e[k], rtol=2e-5, atol=5e-7 if B * solver.num_steps < 6000 else 1e-5, err_msg=k)
        # (the pair-grid network and gamma are stepped by socmx_adam_step_f32 in the replayed body too -- one launch on the second
        #  stream instead of torch's four -- at M_lr = 1e-2, a hundred times the control network's rate: the same last-bit
        #  differences in the bias corrections weigh a hundred times more in the update; 4e-6 norm-wise there)
        nw = 4e-6 if (k.startswith("M.") or "gamma" in k) else 2e-6
        assert np.linalg.norm(par_g[k] - par_e[k]) <= nw * max(np.linalg.norm(par_e[k]), 1e-12) + 1e-9, k
    rec_s, par_s, tr_s, _ = _run(name, True, 7, B)
    if (tr_s._dev or {}).get("saved") is not None:                  # the rollout saved the activations (d <= 15, whole 16-row tiles)
        rec_es, par_es, _, _ = _run(name, False, 7, B)               # ... and so does the eager iteration's (socmx/solver.py)
        np.testing.assert_allclose(rec_s, rec_es, rtol=2e-5, atol=1e-7)
        for k in par_es:
            np.testing.assert_allclose(par_s[k], par_es[k], rtol=2e-5, atol=5e-7 if B * solver.num_steps < 6000 else 1e-5, err_msg=k)
            nw = 4e-6 if (k.startswith("M.") or "gamma" in k) else 2e-6
            assert np.linalg.norm(par_s[k] - par_es[k]) <= nw * max(np.linalg.norm(par_es[k]), 1e-12) + 1e-9, k
        # the two forms against each other
        np.testing.assert_allclose(rec_s[:, :3], rec_e[:, :3], rtol=5e-5, atol=1e-7)       # loss, weight mean / std
        for k in par_e:
            assert np.linalg.norm(par_s[k] - par_e[k]) <= 5e-5 * max(np.linalg.norm(par_e[k]), 1e-12) + 1e-9, k
    else:
        assert np.array_equal(rec_s, rec_g)


@pytest.mark.parametrize("name,B", [("cfg3_double_well_d10_K200", 64), ("oul10_ou_linear_d10_K100_B64", 64),
                                    ("ouq20_ou_quadratic_easy_d20_K12", 40)])
def test_training_is_bit_reproducible_run_to_run(name, B):
    """Two runs of the replayed iteration from the same device key, with other work on the chip in between (dirty LDS, caches):
    weight statistics, gradient telemetry and the final parameters are BIT-identical -- no float atomics anywhere on the path
    that feeds the update (rollout, loss gradients, backward, Adam: fixed summation orders).  Only the reported loss value may
    move in its last bits (one float atomicAdd per workgroup of the contraction; it feeds nothing back)."""
    rec_a, par_a, _, _ = _run(name, True, 5, B)
    junk = torch.randn(2048, 2048, device=DEV)
    (junk @ junk).sum().item()
    _run("tiny_double_well_d10", True, 3, 16)
    rec_b, par_b, _, _ = _run(name, True, 5, B)
    assert np.array_equal(rec_a[:, 1:], rec_b[:, 1:])
    np.testing.assert_allclose(rec_a[:, 0], rec_b[:, 0], rtol=2e-6)
    for k in par_a:
        assert np.array_equal(par_a[k], par_b[k]), k


@pytest.mark.parametrize("alg", ["SOCM_const_M", "SOCM_exp", "SOCM_adjoint", "cross_entropy", "log-variance", "variance",
                                 "moment", "rel_entropy"])
def test_graph_replay_of_the_other_losses_equals_the_eager_iteration(alg):
    """The README sweeps run nine algorithms (README.md:15-60); with `backend.hip_graph: True` as the default every one of
    them replays its iteration as one captured graph -- the eight other losses the captured AUTOGRAD body (HIP rollout on the
    device Philox key, fused loss kernels, control-network backward, Adam incl. the y0 / gamma groups, EMA normaliser).
    Eager Trainer on the same device key = the reference: losses, weight statistics, normaliser, final parameters.
    rel_entropy differentiates through the eager rollout, whose noise comes from torch's generator (other draws under a
    capture): there the run must capture, replay and stay finite."""
    from SOC_matching.method import SOC_Solver
    from socmx.rollout import PhiloxKey
    from socmx.train import Trainer, make_optimizer
    name, B, steps = ("tiny_double_well_d10", 16, 7)
    out = {}
    for graph in (False, True):
        sde, aux = build_sde(name, DEV)
        solver = SOC_Solver(sde, aux["x0"], None, T=aux["T"], num_steps=aux["K"], lmbd=aux["lmbd"], d=aux["d"], sigma=sde.sigma)
        if alg == "SOCM_exp":
            solver.gamma = torch.nn.Parameter(torch.tensor([2.0], device=DEV))
        else:
            solver.gamma = 2.0
        with torch.no_grad():
            solver.y0.fill_(0.37)                 # (method.py:172 draws it: the same start for both runs)
        solver.philox_key = PhiloxKey(torch.device(DEV), seed=21, offset=3)
        opt = make_optimizer(solver, M_lr=1e-3, algorithm=alg)
        logs = []
        tr = Trainer(solver, opt, B, normalization_const=0.7, algorithm=alg, sync_timing=False,
                     hip_graph="force" if graph else False, graph_warmup=2, log=logs.append)
        rec = []
        for _ in range(steps):
            info = tr.step()
            rec.append([float(info["loss"]), float(info["weight_mean"]), float(info["weight_std"]), float(tr.normalization_const)])
        tr.join()
        torch.cuda.synchronize()
        if graph:
            assert tr.hip_graph and not logs, logs                        # captured, no fall-back
            assert len([k for k in tr._graphs if not (isinstance(k, tuple) and k and k[0] == "warm")]) == 1
        out[graph] = (np.array(rec), {k: _np(v) for k, v in sde.state_dict().items()},
                      float(solver.y0.detach()) if alg == "moment" else None)
    assert np.isfinite(out[True][0]).all()
    if alg == "rel_entropy":
        return
    np.testing.assert_allclose(out[True][0], out[False][0], rtol=5e-5, atol=1e-7)
    for k in out[False][1]:
        np.testing.assert_allclose(out[True][1][k], out[False][1][k], rtol=5e-5, atol=1e-6, err_msg=k)
    if alg == "moment":
        np.testing.assert_allclose(out[True][2], out[False][2], rtol=1e-5)


@pytest.mark.parametrize("name", ["train_ou_quadratic_easy_d2", "train_double_well_d10"])
def test_graph_mode_replays_the_reference_training_fixture(name):
    """Reference-generated TRAINING fixture (main.py:280-359 replayed by the reference itself): the noise of iteration n is
    copied into a static buffer before step n -- the captured graph reads that buffer -- and the losses, normaliser
    values and final parameters must be the reference's."""
    from test_host_cpu import run_training_fixture, check_training_fixture
    sde, z, rec = run_training_fixture(name, DEV, hip_graph=True, graph_warmup=1)
    check_training_fixture(sde, z, rec, rtol=1e-3)


def test_graph_mode_falls_back_to_eager_for_checkpoint_iterations():
    """compute_control_objective=True (a burst with host-side statistics) runs eagerly and keeps the device-side state
    (normaliser, iteration counter, telemetry EMAs) in step."""
    rec, par, tr, solver = _run("tiny_double_well_d10", True, 4, 16)
    out = tr.step(compute_control_objective=True, total_n_samples=64)
    assert out["out"][2] is not None and torch.isfinite(out["out"][2])
    after = tr.step()                # (runs the manual body eagerly once: the flush consumed the outstanding update)
    again = tr.step()                # replays again
    assert torch.isfinite(after["loss"]) and torch.isfinite(again["loss"]) and tr.itr == 7
    assert float(tr._dev["itr"]) == 7.0


def test_graph_mode_with_the_ground_truth_L2_error():
    """main.py passes compute_L2_error + the ground-truth control every iteration (main.py:298-309): the manual graph body
    computes the weighted L2 error itself; values equal the eager Trainer's on the same Philox stream."""
    import contextlib, io
    from socmx.config import load_config
    from socmx.settings import define_variables
    from socmx.rollout import PhiloxKey
    from socmx.train import Trainer, make_optimizer
    from SOC_matching.method import SOC_Solver
    recs = []
    for graph in (False, True):
        cfg = load_config(["method.setting=OU_quadratic_easy", "method.d=2", "method.num_steps=50", "method.gamma=2.0",
                           "method.scaling_factor_M=0.1", "optim.M_lr=1e-3"])
        cfg.method.device = DEV
        torch.manual_seed(0)
        ts = torch.linspace(0, 1.0, 51).to(DEV)
        with contextlib.redirect_stdout(io.StringIO()):
            x0, sigma, optimal_sde, sde, _ = define_variables(cfg, ts)
        solver = SOC_Solver(sde, x0, optimal_sde.u, T=1.0, num_steps=50, lmbd=1.0, d=2, sigma=sigma)
        solver.philox_key = PhiloxKey(torch.device(DEV), seed=5, offset=0)
        tr = Trainer(solver, make_optimizer(solver, M_lr=1e-3), 64, sync_timing=False, hip_graph=graph, overlap_M_backward=False)
        rec = []
        for _ in range(6):
            info = tr.step(compute_L2_error=True, optimal_control=optimal_sde.u)
            rec.append([float(info["loss"]), float(info["out"][1])])
        tr.join()
        if graph:
            assert any(k and k[0] == "manual" for k in tr._graphs if isinstance(k, tuple))
        recs.append(np.array(rec))
    np.testing.assert_allclose(recs[1], recs[0], rtol=2e-5, atol=1e-7)
    assert (recs[0][:, 1] > 0).all()


@pytest.mark.gpu
def test_fused_adam_step_matches_torch_adam():
    """socmx_adam_step_f32 (flat gradient -> Adam update + telemetry sums in one launch) against torch.optim.Adam(fused=True) and
    the EMA / squared-norm formulas of main.py:325-345, over several steps (bias corrections, EMA warm-up branch)."""
    from socmx import _lib
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(5)
    shapes = [(256, 16), (256,), (128, 256), (128,), (3, 7), (1,)]
    ref = [torch.randn(*s, generator=g).to(dev).requires_grad_(True) for s in shapes]
    mine = [p.detach().clone().requires_grad_(True) for p in ref]
    opt_ref = torch.optim.Adam(ref, lr=1e-2, eps=1e-4, fused=True)
    opt_mine = torch.optim.Adam(mine, lr=1e-2, eps=1e-4, fused=True)
    total = sum(p.numel() for p in ref)
    # one torch step creates the state the fused entry point updates in place
    g0 = [torch.randn(*s, generator=g).to(dev) for s in shapes]
    for params, opt in ((ref, opt_ref), (mine, opt_mine)):
        for p, gg in zip(params, g0):
            p.grad = gg.clone()
        opt.step()
    rows, off = [], 0
    for p in mine:
        st = opt_mine.state[p]
        rows.append([p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), st["step"].data_ptr(), p.numel(), off])
        off += p.numel()
    table = torch.tensor(rows, dtype=torch.int64, device=dev)
    scratch, sums = torch.zeros(4 + 2 * ((total + 1023) // 1024), device=dev), torch.zeros(2, device=dev)
    ema = torch.zeros(total, device=dev)
    ema_ref = torch.zeros(total, device=dev)
    itr = torch.zeros(1, device=dev)
    L, f = _lib.lib(), _lib.ptr
    for it in range(4):
        grads = [torch.randn(*s, generator=g).to(dev) for s in shapes]
        flat = torch.cat([x.reshape(-1) for x in grads]).contiguous()
        for p, gg in zip(ref, grads):
            p.grad = gg.clone()
        opt_ref.step()
        itr.fill_(float(it))
        with _lib.on_device(dev):
            _lib.check(L.socmx_adam_step_f32(table.data_ptr(), len(mine), total, f(flat), f(ema), f(itr), 0.4, 1e-2, 0.9, 0.999, 1e-4,
                                            f(scratch), f(sums), _lib.stream_ptr(dev)), "socmx_adam_step_f32")
        # EMA coefficient c = 0.4: itr = 0 -> copy, itr <= floor(1/c) = 2 -> running mean, then c g + (1 - c) ema
        if it == 0:
            ema_ref = flat.clone()
        elif it <= 2:
            ema_ref = (it * ema_ref + flat) / (it + 1)
        else:
            ema_ref = 0.4 * flat + 0.6 * ema_ref
        torch.cuda.synchronize()
        np.testing.assert_allclose(sums[0].item(), float(flat.double().pow(2).sum()), rtol=1e-5)
        np.testing.assert_allclose(sums[1].item(), float(ema_ref.double().pow(2).sum()), rtol=1e-5)
        np.testing.assert_allclose(ema.cpu().numpy(), ema_ref.cpu().numpy(), rtol=1e-5, atol=1e-7)
        for a, b in zip(mine, ref):
            np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=1e-5, atol=1e-6)
        assert float(opt_mine.state[mine[0]]["step"]) == float(opt_ref.state[ref[0]]["step"]) == it + 2
    assert float(scratch[0]) == 0.0              # the ticket is re-armed; the rest holds per-workgroup partial sums


@pytest.mark.gpu
@pytest.mark.parametrize("telemetry", [True, False])
def test_adam_step_with_scalars_equals_the_two_launches(telemetry):
    """socmx_adam_step_scalars_f32 = socmx_adam_step_f32 followed by phase 1 of socmx_iteration_scalars_f32 on its sums: parameters,
    Adam state, EMA buffers, the seven outputs, the normaliser and the iteration counter bit for bit, over the EMA warm-up branches."""
    from socmx import _lib
    dev = torch.device("cuda", 0)
    L, f = _lib.lib(), _lib.ptr
    g = torch.Generator().manual_seed(9)
    shapes = [(64, 16), (64,), (33, 64), (33,), (5, 33), (5,)]
    total = sum(int(np.prod(s)) for s in shapes)

    def setup():
        gg = torch.Generator().manual_seed(9)
        ps = [torch.randn(*s, generator=gg).to(dev).requires_grad_(True) for s in shapes]
        opt = torch.optim.Adam(ps, lr=1e-2, eps=1e-4, fused=True)
        for p in ps:
            p.grad = torch.randn(p.shape, generator=gg).to(dev)
        opt.step()
        rows, off = [], 0
        for p in ps:
            st = opt.state[p]
            rows.append([p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), st["step"].data_ptr(), p.numel(), off])
            off += p.numel()
        z = lambda n, v=0.0: torch.full((n,), v, device=dev)
        return dict(ps=ps, opt=opt, table=torch.tensor(rows, dtype=torch.int64, device=dev),
                    scratch=z(4 + 2 * ((total + 1023) // 1024)), sums=z(2), ema=z(total), itr=z(1), norm=z(1, 1.0),
                    ema_gn=z(1), out=z(7))

    A, B = setup(), setup()
    for it in range(5):
        flat = torch.randn(total, generator=g).to(dev)
        w_mean, w_std, obj = (torch.rand(1, generator=g).to(dev) + 0.5 for _ in range(3))
        with _lib.on_device(dev):
            S = A
            _lib.check(L.socmx_adam_step_f32(S["table"].data_ptr(), len(shapes), total, f(flat), f(S["ema"]) if telemetry else None,
                                            f(S["itr"]), 0.4, 1e-2, 0.9, 0.999, 1e-4, f(S["scratch"]), f(S["sums"]),
                                            _lib.stream_ptr(dev)), "socmx_adam_step_f32")
            _lib.check(L.socmx_iteration_scalars_f32(1, f(S["itr"]), f(S["norm"]), f(S["ema_gn"]) if telemetry else None, f(w_mean),
                                                    f(w_std), f(obj), f(S["sums"][0:1]) if telemetry else None,
                                                    f(S["sums"][1:2]) if telemetry else None, 0.3, 0.4, None, f(S["out"]),
                                                    _lib.stream_ptr(dev)), "socmx_iteration_scalars_f32")
            S = B
            _lib.check(L.socmx_adam_step_scalars_f32(S["table"].data_ptr(), len(shapes), total, f(flat),
                                                    f(S["ema"]) if telemetry else None, f(S["itr"]), 0.4, 1e-2, 0.9, 0.999, 1e-4,
                                                    f(S["scratch"]), f(S["sums"]), f(S["norm"]), f(S["ema_gn"]) if telemetry else None,
                                                    f(w_mean), f(w_std), f(obj), 0.3, f(S["out"]), _lib.stream_ptr(dev)),
                       "socmx_adam_step_scalars_f32")
        torch.cuda.synchronize()
        for k in ("out", "norm", "itr", "ema_gn", "ema", "sums"):
            assert torch.equal(A[k], B[k]), (it, k, A[k], B[k])
        for a, b in zip(A["ps"], B["ps"]):
            assert torch.equal(a, b)
        assert float(B["itr"]) == it + 1 and float(B["scratch"][0]) == 0.0
        assert (float(B["out"][3]) > 0) == telemetry


@pytest.mark.parametrize("fused_adam", [True, False])
def test_graph_mode_survives_an_optimizer_reload_and_an_lr_change(fused_adam):
    """the captured iteration bakes in the Adam state's data pointers and the groups' hyper-parameters.  A
    resume (`optimizer.load_state_dict`: new state tensors) or an lr change must drop the device table and the captured
    graphs -- the run must continue exactly like an eager Trainer that went through the same reload -- with the fused Adam
    table (socmx_adam_step_f32) and without it (torch's capturable Adam inside the graph: the signature of the optimiser
    state is checked for every captured body)."""
    import copy
    out = {}
    for graph in (False, True):
        from SOC_matching.method import SOC_Solver
        from socmx.rollout import PhiloxKey
        from socmx.train import Trainer, make_optimizer
        sde, aux = build_sde("cfg1_ou_quadratic_easy_d2_K50", DEV)
        solver = SOC_Solver(sde, aux["x0"], None, T=aux["T"], num_steps=aux["K"], lmbd=aux["lmbd"], d=aux["d"], sigma=sde.sigma)
        solver.philox_key = PhiloxKey(torch.device(DEV), seed=3, offset=0)
        opt = make_optimizer(solver, M_lr=1e-3)
        tr = Trainer(solver, opt, 32, normalization_const=0.9, sync_timing=False, hip_graph=graph, overlap_M_backward=False,
                     fused_adam=fused_adam)
        rec = []
        for it in range(12):
            if it == 5:
                tr.join()
                torch.cuda.synchronize()
                opt.load_state_dict(copy.deepcopy(opt.state_dict()))      # resume: every state tensor is a new allocation
            if it == 8:
                tr.join()
                for g in opt.param_groups:
                    g["lr"] = g["lr"] * 0.5
            rec.append(float(tr.step()["loss"]))
        tr.join()
        torch.cuda.synchronize()
        if graph:
            assert any(k and k[0] == "manual" for k in tr._graphs if isinstance(k, tuple))
        out[graph] = (np.array(rec), {k: _np(v) for k, v in sde.state_dict().items()})
    np.testing.assert_allclose(out[True][0], out[False][0], rtol=5e-5, atol=1e-7)
    for k in out[False][1]:
        np.testing.assert_allclose(out[True][1][k], out[False][1][k], rtol=5e-5, atol=1e-6, err_msg=k)


def test_keyed_rollouts_and_bursts_draw_from_disjoint_philox_streams():
    """hipGraph mode's device key must not re-use the offsets the host-side counter hands to the evaluation
    bursts under the same seed (iteration n would train on the rows of the n-th burst)."""
    from socmx import rollout as R
    from socmx.train import KEYED_OFFSET_BASE
    from SOC_matching.method import SOC_Solver
    from socmx.train import Trainer, make_optimizer
    sde, aux = build_sde("tiny_double_well_d10", DEV)
    solver = SOC_Solver(sde, aux["x0"], None, T=aux["T"], num_steps=aux["K"], lmbd=aux["lmbd"], d=aux["d"], sigma=sde.sigma)
    tr = Trainer(solver, make_optimizer(solver, M_lr=1e-3), 16, sync_timing=False, hip_graph=True)
    before = R._philox_calls
    for _ in range(3):
        tr.step()
    tr.join()
    key = solver.philox_key.key.cpu().tolist()
    assert key[1] == KEYED_OFFSET_BASE + 3 and KEYED_OFFSET_BASE >= 1 << 31
    assert R._philox_calls - before < 1 << 20          # the host counter stays far below the keyed stream's base
    # the same seed at host offset 0 and at the keyed stream's first offset gives different noise
    x0 = aux["x0"].repeat(16, 1)
    a = R.stochastic_trajectories(sde, x0, aux["ts"], aux["lmbd"], seed=torch.initial_seed(), offset=0)[1]
    b = R.stochastic_trajectories(sde, x0, aux["ts"], aux["lmbd"], seed=torch.initial_seed(), offset=KEYED_OFFSET_BASE)[1]
    assert not torch.equal(a, b)
