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
    out = solver.loss(B, algorithm="SOCM", use_warm_start=False)
    out[0].backward()
    params = list(sde.nabla_V.parameters()) + list(sde.M.sigmoid_layers.parameters()) + [sde.gamma]
    (obj,) = solver.shard.allreduce_gradients(params, extra=[out[0].detach()])
    if rank == 0:
        ret["objective"] = float(obj)
        ret["w_mean"], ret["w_std"] = float(out[5]), float(out[6])
        ret["grads"] = [p.grad.numpy().copy() for p in params]
        ret["rows"] = (Bl, row0)
    dist.destroy_process_group()


@pytest.mark.parametrize("name", ["tiny_ou_linear_d5_B20", "tiny_double_well_d10"])
def test_sharded_step_equals_single_process(name):
    world = 2
    port = 29500 + (os.getpid() % 2000)
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
