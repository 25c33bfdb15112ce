"""Batch sharding over ranks (world_size 2, gloo, CPU): the sharded SOCM step with the flat-buffer all-reduce
reproduces the single-process objective, weight statistics and gradients on the same noise."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from test_host_cpu import build_sde


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _worker(rank, world, port, name, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from SOC_matching.method import SOC_Solver
    from socmx.dist import Shard
    sde, aux = build_sde(name)
    solver = SOC_Solver(sde, aux["x0"], None, T=aux["T"], num_steps=aux["K"], lmbd=aux["lmbd"], d=aux["d"],
                        sigma=sde.sigma)
    solver.shard = Shard()
    B = aux["B"]
    Bl, row0 = solver.shard.local_rows(B)
    solver.noise_in = aux["noise"][:, row0:row0 + Bl].contiguous()
    out = solver.loss(B, algorithm="SOCM", use_warm_start=False, use_stopping_time=aux["stopping"])
    out[0].backward()
    params = list(sde.nabla_V.parameters()) + list(sde.M.sigmoid_layers.parameters()) + [sde.gamma]
    if aux["stopping"]:
        params.append(sde.gamma2)
    (obj,) = solver.shard.allreduce_gradients(params, extra=[out[0].detach()])
    if rank == 0:
        ret["objective"] = float(obj)
        ret["w_mean"], ret["w_std"] = float(out[5]), float(out[6])
        ret["grads"] = [p.grad.numpy().copy() for p in params]
        ret["rows"] = (Bl, row0)
    dist.destroy_process_group()


@pytest.mark.parametrize("name", ["tiny_ou_linear_d5_B20", "tiny_double_well_d10", "tiny_molecular_dynamics_d2_stopping"])
def test_sharded_step_equals_single_process(name):
    world = 2
    port = _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, port, name, ret), nprocs=world, join=True)
        ret = dict(ret)
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"))
    np.testing.assert_allclose(ret["objective"], z["loss_objective"], rtol=1e-4)
    np.testing.assert_allclose(ret["w_mean"], z["loss_weight_mean"], rtol=1e-4)
    np.testing.assert_allclose(ret["w_std"], z["loss_weight_std"], rtol=1e-4)
    sde, aux = build_sde(name)
    names = ["grad_nablaV." + k for k, _ in sde.nabla_V.named_parameters()] + \
            ["grad_M.sigmoid_layers." + k for k, _ in sde.M.sigmoid_layers.named_parameters()] + ["grad_gamma"]
    if name.endswith("_stopping"):        # (the stopping-time loss's normaliser sum(stop_indicators) is all-reduced before the backward)
        names.append("grad_gamma2")
    for g, n in zip(ret["grads"], names):
        np.testing.assert_allclose(g, z[n], rtol=1e-3, atol=1e-5 * max(1.0, np.abs(z[n]).max()), err_msg=n)


def test_local_rows_partition():
    from socmx.dist import Shard
    for B in (8, 9, 20, 1024):
        for G in (1, 2, 3, 8):
            if B < G:
                continue
            rows = [Shard(rank=r, world_size=G).local_rows(B) for r in range(G)]
            assert sum(b for b, _ in rows) == B
            assert rows[0][1] == 0
            for (b0, r0), (b1, r1) in zip(rows[:-1], rows[1:]):
                assert r1 == r0 + b0


def test_combine_stats_is_exact_pooling():
    from socmx import loss as L
    g = torch.Generator().manual_seed(0)
    w = torch.rand(37, generator=g)
    parts = [w[:10], w[10:29], w[29:]]
    stats = torch.stack([torch.stack([p.sum(), ((p - p.mean()) ** 2).sum(), torch.tensor(float(len(p)))]) for p in parts])
    mean, std = L.mean_std_from_stats(L.combine_stats(stats))
    np.testing.assert_allclose(mean.item(), w.mean().item(), rtol=1e-6)
    np.testing.assert_allclose(std.item(), w.std().item(), rtol=1e-5)


def _train_worker(rank, world, port, name, ret):
    """Three-four `Trainer.step` iterations of a REFERENCE-generated training fixture (main.py:280-359 replayed by the
    reference itself, tests/golden/make_golden_train.py) with the batch split over two ranks."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from SOC_matching.method import SOC_Solver
    from socmx.dist import Shard
    from socmx.train import Trainer, make_optimizer
    sde, aux = build_sde(name)
    z = aux["z"]
    lr_V, lr_M, eps, norm0 = [float(v) for v in z["meta_f"][5:9]]
    solver = SOC_Solver(sde, aux["x0"], None, T=aux["T"], num_steps=aux["K"], lmbd=aux["lmbd"], d=aux["d"], sigma=sde.sigma)
    solver.shard = Shard()
    counts = {"all_reduce": 0, "all_gather": 0}
    real_ar, real_ag = dist.all_reduce, dist.all_gather
    dist.all_reduce = lambda *a, **k: (counts.__setitem__("all_reduce", counts["all_reduce"] + 1), real_ar(*a, **k))[1]
    dist.all_gather = lambda *a, **k: (counts.__setitem__("all_gather", counts["all_gather"] + 1), real_ag(*a, **k))[1]
    opt = make_optimizer(solver, nabla_V_lr=lr_V, M_lr=lr_M, adam_eps=eps)
    tr = Trainer(solver, opt, aux["B"], normalization_const=norm0, sync_timing=False)
    Bl, row0 = solver.shard.local_rows(aux["B"])
    rec = dict(loss=[], weight_mean=[], weight_std=[], norm=[])
    n_it = int(z["train_iters"])
    for it in range(n_it):
        solver.noise_in = aux["noise"][it][:, row0:row0 + Bl].contiguous()
        out = tr.step()
        rec["loss"].append(float(out["loss"]))
        rec["weight_mean"].append(float(out["weight_mean"]))
        rec["weight_std"].append(float(out["weight_std"]))
        rec["norm"].append(float(tr.normalization_const))
    tr.join()
    dist.all_reduce, dist.all_gather = real_ar, real_ag
    state = {("V." + k): v.detach().numpy().copy() for k, v in sde.nabla_V.state_dict().items()}
    state.update({("M." + k): v.detach().numpy().copy() for k, v in sde.M.state_dict().items()})
    state["gamma"] = sde.gamma.detach().numpy().copy()
    ret[rank] = dict(rec=rec, state=state, collectives=dict(counts), iters=n_it, rows=(Bl, row0))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("name", ["train_ou_quadratic_easy_d2", "train_double_well_d10"])
def test_sharded_trainer_steps_equal_the_reference_training_run(name, world):
    """world_size 2 and 3 (gloo): `Trainer.step` with a batch shard -- loss, backward, ONE flat all-reduce (gradients + loss
    value + the ranks' weight-statistics slots), Adam with the reference's groups, EMA normaliser -- reproduces the reference's own training
    run on the same noise, all ranks end with identical parameters, and exactly one collective is issued per iteration.
    The fixtures' batch is 16 rows: three ranks hold 6 / 5 / 5 of them (a split that does not divide)."""
    port = _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_train_worker, args=(world, port, name, ret), nprocs=world, join=True)
        ret = {k: v for k, v in ret.items()}
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"))
    r0, r1 = ret[0], ret[world - 1]
    assert sorted(ret) == list(range(world))
    assert r0["collectives"] == {"all_reduce": r0["iters"], "all_gather": 0}, r0["collectives"]
    assert [ret[r]["rows"] for r in range(world)] == ([(8, 0), (8, 8)] if world == 2 else [(6, 0), (5, 6), (5, 11)])
    np.testing.assert_allclose(r0["rec"]["loss"], z["train_loss"], rtol=2e-4)
    np.testing.assert_allclose(r0["rec"]["weight_mean"], z["train_weight_mean"], rtol=2e-4)
    np.testing.assert_allclose(r0["rec"]["norm"], z["train_norm_const"], rtol=2e-4)
    for k in ("loss", "weight_mean", "weight_std", "norm"):
        assert r0["rec"][k] == r1["rec"][k], k                        # every rank reports the same reduced values
    num = den = 0.0
    for prefix, tag in (("final_nablaV.", "V."), ("final_M.", "M.")):
        for k in [k for k in z.files if k.startswith(prefix)]:
            a = r0["state"][tag + k[len(prefix):]]
            num += float(((a - z[k]) ** 2).sum())
            den += float((z[k] ** 2).sum())
    assert (num / den) ** 0.5 < 2e-3, (num / den) ** 0.5
    np.testing.assert_allclose(r0["state"]["gamma"], z["final_gamma"], rtol=2e-3)
    for k in r0["state"]:
        assert np.array_equal(r0["state"][k], r1["state"][k]), k      # replicas stay bit-identical


def test_weight_stat_slots_pool_exactly():
    """socmx/dist.py: each rank fills ITS (n, mean, M2) slot, the all-reduce(SUM) of slots that are zero elsewhere is exact, Chan's
    rule pools them in fp64 -- one-process accuracy at any scale of the weights (1e-3 relative spread around 0.1, and weights of
    size 1e-6: the cases where rounds 4-5's sums shifted by a far-off constant cancelled)."""
    from socmx.dist import weight_stat_slots, mean_std_from_slots
    g = torch.Generator().manual_seed(1)
    for w in (0.1 + 1e-4 * torch.rand(300, generator=g), 1e-6 * torch.rand(300, generator=g), 30.0 + torch.randn(300, generator=g)):
        parts = [w[:100], w[100:170], w[170:]]
        total = sum(weight_stat_slots(p, r, 3) for r, p in enumerate(parts))
        for r, p in enumerate(parts):                      # the sum IS the concatenation of the ranks' own slots
            assert torch.equal(total[3 * r:3 * r + 3], weight_stat_slots(p, r, 3)[3 * r:3 * r + 3])
        mean, std = mean_std_from_slots(total)
        np.testing.assert_allclose(mean.item(), w.double().mean().item(), rtol=2e-7)
        np.testing.assert_allclose(std.item(), w.double().std().item(), rtol=2e-6)


def _agree_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from socmx.dist import Shard
    sh = Shard()
    # every rank ok -> True; one rank not ok -> False on EVERY rank (the ranks then take the same fall-back together)
    ret[rank] = (sh.agree(True), sh.agree(rank != 1), sh.agree(False))
    dist.destroy_process_group()


def test_ranks_agree_on_a_capture_before_anyone_replays():
    """train.py: a sharded run that opts in to capturing its iteration (hip_graph="sharded") replays only if EVERY rank
    captured -- `Shard.agree` is the one all_reduce(MIN) that decides it (world size 2, gloo)."""
    port = _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_agree_worker, args=(2, port, ret), nprocs=2, join=True)
        ret = dict(ret)
    assert ret[0] == (True, False, False) and ret[1] == (True, False, False)


def test_a_sharded_run_captures_only_over_its_own_communicators():
    """The rule of Trainer.__init__ (no GPU needed: the flags are decided before anything is captured): several ranks capture
    their iteration iff the shard's transport is the package's own RCCL communicators (`Shard.capturable`); over torch's process
    group or the host-staged transport the autograd-free body runs eagerly whatever the config key says -- and says so."""
    from types import SimpleNamespace
    from socmx.train import Trainer

    class FakeShard:
        def __init__(self, w, transport):
            self.world_size, self.transport = w, transport
            self.capturable = w == 1 or transport == "rccl"

    def flags(hip_graph, world, cuda=True, transport="rccl"):
        logs = []
        solver = SimpleNamespace(x0=SimpleNamespace(is_cuda=cuda), shard=FakeShard(world, transport) if world else None,
                                 neural_sde=SimpleNamespace(use_stopping_time=False))
        tr = Trainer(solver, None, 8, hip_graph=hip_graph, log=logs.append)
        return tr.hip_graph, tr.capture_graphs, logs

    assert flags(True, 0)[:2] == (True, True)
    assert flags(True, 1, transport="group")[:2] == (True, True)    # one rank: nobody to disagree with
    assert flags(True, 8)[:2] == (True, True)                       # own communicators: the default captures
    for mode in (True, "sharded", "force"):
        for transport in ("group", "staged"):
            hg, cap, logs = flags(mode, 8, transport=transport)
            assert (hg, cap) == (True, False) and any("sharded run" in m and transport in m for m in logs)   # said, never silent
    assert flags("nocapture", 0)[:2] == (True, False)
    assert flags("nocapture", 8)[:2] == (True, False)
    assert flags(True, 8, cuda=False)[:2] == (False, False)


def test_shard_transport_follows_the_process_group():
    """socmx/dist.py: an un-initialised process (rank / world given by hand) and a gloo group on CPU tensors use the group's own
    calls; only backend nccl brings up the package's communicators (GPU tests), only gloo + a CUDA device stages through the host."""
    from socmx.dist import Shard
    sh = Shard(rank=1, world_size=4)
    assert sh.transport == "group" and not sh.capturable and sh.local_rows(10) == (3, 3)
    assert Shard(rank=0, world_size=1).capturable
