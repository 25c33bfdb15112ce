"""GPU parity of the SAVED control-network backward (round 6): the one-row rollout writes the network's activations and ReLU signs of
every trajectory row where the backward reads them (socmx_rollout_ex_f32: act_workspace / act_records), and
socmx_unet_backward_saved_f32 runs the five backward stages only.  Checked here: the export changes nothing the rollout returns (bit for
bit), the slabs are the activations kernel A would re-compute, the sign records decode -- with an independent reading of the format in
csrc/socmx_unet.h -- to the signs of those activations, the gradients of the two backward entries agree, the query refuses what the kernel
cannot do, and the reference's own training runs come out the same with the switch on and off.  Semantics: models.py:233-242, method.py:272-278."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [HERE]
from test_host_cpu import build_sde, run_training_fixture, check_training_fixture  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
WIDTHS = [("X", 16), ("R1", 256), ("R2", 128), ("R3", 64), ("O2", 128), ("A1", 256)]


def _slab_rows(work, N, name):
    """(N, W) matrix of slab tensor `name` from the backward's workspace: tensor t at 16 ntiles prefix(t), [tile][unit][16 rows]."""
    off = 0
    for nm, w in WIDTHS:
        if nm == name:
            return work[off * N:(off + w) * N].reshape(N // 16, w, 16).permute(0, 2, 1).reshape(N, w)
        off += w
    raise KeyError(name)


def _decode_records(rec):
    """The six sign masks of every row from its 32-dword record -- written from the format's description (csrc/socmx_unet.h), not from
    act_record_nibble."""
    r = rec.cpu().numpy().astype(np.uint32).reshape(-1, 8, 4)
    N = r.shape[0]
    out = {}
    u = np.arange(128)
    out["R2"] = (r[:, u >> 4, 0] >> (u & 15)) & 1
    out["U2"] = (r[:, u >> 4, 0] >> (16 + (u & 15))) & 1
    u = np.arange(256)
    out["A1"] = (r[:, u >> 5, 1] >> (u & 31)) & 1

    def lane_of(e):                                   # element 16 a + 4 b + i of a 64-chunk sits in lane 16 b + 4 a + i
        a, b, i = (e >> 4) & 3, (e >> 2) & 3, e & 3
        return 16 * b + 4 * a + i
    u = np.arange(256)
    x = lane_of(u & 63)
    out["R1"] = (r[:, 1 + (u >> 6), 2 + (x >> 5)] >> (x & 31)) & 1
    u = np.arange(64)
    x = lane_of(u)
    out["R3"] = (r[:, 5, 2 + (x >> 5)] >> (x & 31)) & 1
    u = np.arange(16)
    out["U0"] = (r[:, 0, 2][:, None] >> u[None, :]) & 1
    assert out["R1"].shape == (N, 256)
    return out


def _export_rollout(name):
    from socmx import rollout, _lib
    sde, aux = build_sde(name, DEV)
    K, d = aux["K"], aux["d"]
    B = aux["B"]
    if (K + 1) * B % 16 == 0:
        kw = dict(noise_in=aux["noise"])                    # the fixture's noise
    else:
        B, kw = 16, dict(seed=7, offset=0)                  # (whole 16-row tiles: the kernel's own Philox noise)
    state0 = aux["x0"].repeat(B, 1)
    net = sde.nabla_V
    N = (K + 1) * B
    ws_n, ng = _lib.C.c_int64(0), _lib.C.c_int64(0)
    _lib.check(net.hip_lib().socmx_unet_backward_sizes(d, _lib.i3(net.hdims), N, _lib.C.byref(ws_n), _lib.C.byref(ng)), "sizes")
    work = torch.full((ws_n.value,), float("nan"), dtype=torch.float32, device=DEV)
    rec = torch.zeros(N, 32, dtype=torch.int32, device=DEV)
    assert rollout.saves_activations(sde, state0, B, K)
    plain = rollout.hip_trajectories(sde, state0, aux["ts"], aux["lmbd"], want_nabla_v=True, **kw)
    out = rollout.hip_trajectories(sde, state0, aux["ts"], aux["lmbd"], want_nabla_v=True, act_export=(work, rec), **kw)
    torch.cuda.synchronize()
    return sde, aux, B, N, work, rec, plain, out


# the default-width fixtures the one-row kernel takes: MODE 0 / DMAX 11, MODE 2 / DMAX 3, a dense sigma (MODE 6 / DMAX 11), and a small batch
CASES = ["cfg3_full_double_well_d10_K200_B128", "cfg1_full_ou_quadratic_easy_d2_K50_B128", "oul10_ou_linear_d10_K100_B64",
         "cfg3_double_well_d10_K200", "md_default_d1_K150_B64_stopping"]       # (the last: with a stopping time, MODE 1)


@pytest.mark.parametrize("name", CASES)
def test_rollout_saves_what_the_backward_recomputes(name):
    from socmx import nets, _lib
    sde, aux, B, N, work, rec, plain, out = _export_rollout(name)
    K, d = aux["K"], aux["d"]
    net = sde.nabla_V
    for a, b, nm in zip(plain, out, "states noises stop frac lpd lps ltw controls nabla_v".split()):
        assert torch.equal(a, b), nm                              # the export changes nothing the rollout returns
    states, nabla_v = out[0], out[8]
    x = states.reshape(N, d)
    gout = torch.randn(N, d, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
    # kernel A's own activations: the re-computing entry on a workspace of its own
    ws_n, ng = _lib.C.c_int64(0), _lib.C.c_int64(0)
    L = net.hip_lib()
    _lib.check(L.socmx_unet_backward_sizes(d, _lib.i3(net.hdims), N, _lib.C.byref(ws_n), _lib.C.byref(ng)), "sizes")
    work2 = torch.empty(ws_n.value, dtype=torch.float32, device=DEV)
    flat_ref = torch.empty(ng.value, dtype=torch.float32, device=DEV)
    ts = aux["ts"].to(device=DEV, dtype=torch.float32).contiguous()
    with _lib.on_device(torch.device(DEV)):
        _lib.check(L.socmx_unet_backward_scaled_f32(_lib.ptr(net.packed()), _lib.ptr(net.packed_bwd()), d, _lib.i3(net.hdims), _lib.ptr(x),
                                                    _lib.ptr(ts), B, N, _lib.ptr(gout), None, _lib.ptr(work2), _lib.ptr(flat_ref),
                                                    _lib.stream_ptr(torch.device(DEV))), "scaled")
    torch.cuda.synchronize()
    acts = {}
    for nm, _ in WIDTHS[1:]:
        a, b = _slab_rows(work, N, nm), _slab_rows(work2, N, nm)
        assert torch.isfinite(a).all(), nm
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-5, atol=2e-6, err_msg=nm)
        acts[nm] = b.cpu().numpy()
    # the records against the signs of the re-computed activations (where the value is not within rounding of zero)
    m = _decode_records(rec)
    for nm in ("R1", "R2", "R3", "A1"):
        clear = np.abs(acts[nm]) > 1e-6
        assert np.array_equal(m[nm][clear], (acts[nm] > 0)[clear].astype(np.uint32)), nm
        assert not m[nm][(acts[nm] == 0) & (np.abs(_slab_rows(work, N, nm).cpu().numpy()) == 0)].any(), nm    # relu = 0 <-> sign bit clear
    with torch.no_grad():
        res2 = torch.nn.functional.linear(torch.from_numpy(acts["R2"]).to(DEV), net.res_2[0].weight, net.res_2[0].bias).cpu().numpy()
        up2 = acts["O2"] - res2                                       # relu(up_2 R3 + b): models.py:238
        clear = np.abs(up2) > 1e-4
        assert np.array_equal(m["U2"][clear], (up2 > 0)[clear].astype(np.uint32))
        tx = torch.cat([ts.reshape(-1, 1, 1).expand(K + 1, B, 1), states], -1).reshape(N, d + 1)
        res0 = torch.nn.functional.linear(tx, net.res_0[0].weight, net.res_0[0].bias)
        up0 = (nabla_v.reshape(N, d) - res0).cpu().numpy()            # relu(up_0 o1 + b): models.py:241
        clear = np.abs(up0) > 1e-4
        assert np.array_equal(m["U0"][:, :d][clear], (up0 > 0)[clear].astype(np.uint32))
    # the gradients of the two entries
    _, flat_s = nets.unet_backward_hip(net, x, ts, B, gout, return_flat=True, saved=(work, rec))
    torch.cuda.synchronize()
    # Every sign agrees -> 1e-7.  A unit whose pre-activation is within rounding of zero can come out on the other side in the rollout's VALU
    # arithmetic than in kernel A's MFMA arithmetic (both are fp32 evaluations of the same network; the gradient is discontinuous there):
    # a handful of units among the N x 848 -- counted here -- and then the two entries differ like two fp32 runs of autograd would.
    flips = sum(int((m[nm] != (acts[nm] > 0).astype(np.uint32)).sum()) for nm in ("R1", "R2", "R3", "A1"))
    err = float((flat_s - flat_ref).norm() / flat_ref.norm())
    assert flips <= 2e-6 * N * 848 + 2, flips
    assert err < (1e-5 if flips == 0 else 2e-3), (err, flips)
    # ... and against library autograd (the tolerance of the other gradient tests)
    for p in net.parameters():
        p.grad = None
    tx = torch.cat([ts.reshape(-1, 1, 1).expand(K + 1, B, 1), states], -1).reshape(N, d + 1)
    net(tx).backward(gout)
    flat_t = torch.cat([torch.cat([getattr(net, n)[0].weight.grad.reshape(-1), getattr(net, n)[0].bias.grad.reshape(-1)]) for n in _lib.UNET_LAYERS])
    assert float((flat_s - flat_t).norm() / flat_t.norm()) < 1e-3


def test_the_query_refuses_what_the_kernel_cannot_save():
    from socmx import rollout, _lib
    sde, aux = build_sde("cfg3_double_well_d10_K200", DEV)
    K, d = aux["K"], aux["d"]
    x16 = aux["x0"].repeat(16, 1)
    assert rollout.saves_activations(sde, x16, 16, K)
    assert rollout.saves_activations(sde, x16, 256, 15)
    s3, a3 = build_sde("tiny_double_well_d10", DEV)
    assert not rollout.saves_activations(s3, a3["x0"].repeat(16, 1), 16, 15)     # hidden widths the one-row kernel is not built for
    assert not rollout.saves_activations(sde, x16, 272, 15)                 # beyond the one-row kernel's batches
    assert not rollout.saves_activations(sde, x16, 10, 4)                   # 50 rows: a ragged last tile
    for other in ("tiny_ou_linear_d20", "tiny_molecular_dynamics_d2_stopping", "tiny_ou_linear_d64"):
        s2, a2 = build_sde(other, DEV)
        assert not rollout.saves_activations(s2, a2["x0"].repeat(16, 1), 16, 15), other     # 32-wide network / stopping time / d = 64
    # a launch that is asked for it anyway says so
    s2, a2 = build_sde("tiny_ou_linear_d20", DEV)
    N = (a2["K"] + 1) * 16
    work = torch.empty(N * 4096, dtype=torch.float32, device=DEV)
    rec = torch.zeros(N, 32, dtype=torch.int32, device=DEV)
    with pytest.raises(_lib.SocmxError):
        rollout.hip_trajectories(s2, a2["x0"].repeat(16, 1), a2["ts"], a2["lmbd"], seed=0, offset=0, want_nabla_v=True, act_export=(work, rec))
    # ... and the saved entry refuses shapes it was not built for
    from socmx import nets
    net = s2.nabla_V
    x = torch.zeros(N, a2["d"], device=DEV)
    with pytest.raises(_lib.SocmxError):
        nets.unet_backward_hip(net, x, a2["ts"], 16, torch.zeros_like(x), saved=(work, rec))


@pytest.mark.parametrize("graph", [True])          # (the autograd-free body is what a captured iteration runs; the eager autograd iteration re-computes)
@pytest.mark.parametrize("name,B", [("cfg3_double_well_d10_K200", 32), ("cfg1_full_ou_quadratic_easy_d2_K50_B128", 128),
                                    ("oul10_ou_linear_d10_K100_B64", 64)])
def test_training_is_the_same_with_and_without_saved_activations(name, B, graph):
    """Six SOCM iterations of Trainer.step (default widths: the one-row kernel) with the switch on (the default) and off, same Philox
    key: the backward's data path sees the same signs up to units at rounding distance of zero, only the activations feeding the weight
    gradients come from another kernel's fp32 arithmetic -- losses and parameters agree far inside the tolerance the re-computing path is
    held to against the reference (tests/test_gpu_parity.py, test_gpu_graph.py: the training fixtures, 1e-3)."""
    from SOC_matching.method import SOC_Solver
    from socmx.train import Trainer, make_optimizer

    def run(save):
        sde, aux = build_sde(name, DEV)
        solver = SOC_Solver(sde, aux["x0"], None, T=aux["T"], num_steps=aux["K"], lmbd=aux["lmbd"], d=aux["d"], sigma=sde.sigma)
        torch.manual_seed(11)
        tr = Trainer(solver, make_optimizer(solver, nabla_V_lr=1e-4, M_lr=1e-3), B, normalization_const=0.8, sync_timing=False,
                     hip_graph=graph, save_activations=save)
        losses = [float(tr.step()["loss"]) for _ in range(6)]
        tr.join()
        assert ((tr._dev or {}).get("saved") is not None) == (save and graph)     # (the autograd-free body: what a captured iteration runs)
        return losses, {k: v.detach().cpu().numpy().copy() for k, v in sde.nabla_V.state_dict().items()}, \
            {k: v.detach().cpu().numpy().copy() for k, v in sde.M.state_dict().items()}

    l_on, v_on, m_on = run(True)
    l_off, v_off, m_off = run(False)
    np.testing.assert_allclose(l_on, l_off, rtol=2e-5)
    # (Adam divides by the gradient's running magnitude: where a gradient is near zero its rounding noise moves the weight by a visible
    #  fraction of lr per step -- the bound is 2 % of what six steps can move a weight at all)
    for k in v_on:
        np.testing.assert_allclose(v_on[k], v_off[k], rtol=0, atol=0.02 * 6 * 1e-4, err_msg=k)
    for k in m_on:
        np.testing.assert_allclose(m_on[k], m_off[k], rtol=0, atol=0.02 * 6 * 1e-3, err_msg=k)


def test_the_trainer_takes_the_saved_path_where_it_can():
    from SOC_matching.method import SOC_Solver
    from socmx.train import Trainer, make_optimizer
    sde, aux = build_sde("cfg3_double_well_d10_K200", DEV)
    solver = SOC_Solver(sde, aux["x0"], None, T=aux["T"], num_steps=aux["K"], lmbd=aux["lmbd"], d=aux["d"], sigma=sde.sigma)
    tr = Trainer(solver, make_optimizer(solver, M_lr=1e-3), 32, sync_timing=False, hip_graph=True)
    for _ in range(5):
        info = tr.step()
    tr.join()
    assert "graph" in info["mode"]
    saved = (tr._dev or {}).get("saved")
    assert saved is not None and saved[1].shape == ((aux["K"] + 1) * 32, 32)
    assert int((saved[1] != 0).sum()) > 0                        # the rollout wrote records
    # a batch whose rows do not fill 16-row tiles keeps the re-computing backward
    sde2, aux2 = build_sde("cfg3_double_well_d10_K200", DEV)
    solver2 = SOC_Solver(sde2, aux2["x0"], None, T=aux2["T"], num_steps=aux2["K"], lmbd=aux2["lmbd"], d=aux2["d"], sigma=sde2.sigma)
    tr2 = Trainer(solver2, make_optimizer(solver2, M_lr=1e-3), 24, sync_timing=False)
    for _ in range(2):
        tr2.step()
    tr2.join()
    assert (tr2._dev or {}).get("saved") is None
