"""One rank of tests/test_gpu_dist.py: `python gpu_dist_child.py RANK WORLD PORT MODE NAME OUT`.

Every rank uses cuda:0 (a one-GPU box) and a gloo process group; `Shard(device="cuda:0")` then stages its collectives through
host memory (socmx/dist.py, transport "staged"), so that the SHARDED HIP code path -- the autograd-free body whose backward
kernel writes into the flat all-reduce buffer, rollouts with row0 > 0, uneven splits -- runs at world sizes RCCL itself cannot
be given on one device.  World size 1 runs the same code unsharded (the comparison run of the Philox modes)."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [ROOT, os.path.join(ROOT, "soc-matching_amd"), HERE]


def _np(t):
    return t.detach().to("cpu", torch.float32).numpy()


def main():
    rank, world, port, mode, name, out_path = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5], sys.argv[6]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", port
    # SOCMX_TEST_NCCL=1 (boxes with one GPU per rank): rank r on cuda:r over RCCL -- the shard's own communicators, the captured iteration
    real = os.environ.get("SOCMX_TEST_NCCL") == "1"
    dev = torch.device("cuda", rank if real else 0)
    torch.cuda.set_device(dev)
    sharded = world > 1
    if sharded:
        if real:
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    from test_host_cpu import build_sde
    from SOC_matching.method import SOC_Solver
    from socmx import rollout as R
    from socmx.dist import Shard
    from socmx.rollout import PhiloxKey
    from socmx.train import Trainer, make_optimizer
    sde, aux = build_sde(name, str(dev))
    z = aux["z"]
    solver = SOC_Solver(sde, aux["x0"], None, T=aux["T"], num_steps=aux["K"], lmbd=aux["lmbd"], d=aux["d"], sigma=sde.sigma)
    # (SOCMX_TEST_PHILOX_B: another global batch for the Philox mode -- 64 over two ranks = whole 16-row tiles on every rank, where
    #  the rollout SAVES the control network's activations for the sharded body's backward)
    B = int(os.environ.get("SOCMX_TEST_PHILOX_B", "50")) if mode == "philox" else aux["B"]
    Bl, row0 = B, 0
    if sharded:
        solver.shard = Shard() if real else Shard(device=dev)
        assert solver.shard.transport == ("rccl" if real else "staged") and solver.shard.capturable == real, solver.shard.transport_note
        Bl, row0 = solver.shard.local_rows(B)
    res = dict(rank=rank, rows=[Bl, row0])
    arrays = {}
    bodies = dict(manual=0, eager=0)

    def count_bodies(tr):
        real_manual, real_eager = tr._body_manual, tr._eager_step
        tr._body_manual = lambda *a, **k: (bodies.__setitem__("manual", bodies["manual"] + 1), real_manual(*a, **k))[1]
        tr._eager_step = lambda *a, **k: (bodies.__setitem__("eager", bodies["eager"] + 1), real_eager(*a, **k))[1]

    if mode in ("train", "train_eager"):
        # Trainer.step on a reference-generated training fixture, batch rows [row0, row0 + Bl) of every iteration's noise.
        # "train": the default schedule of a sharded GPU run (backend.hip_graph True -> the autograd-free body, eager over this
        # transport); "train_eager": hip_graph False -> the autograd iteration with the flat gradient all-reduce
        lr_V, lr_M, eps, norm0 = [float(v) for v in z["meta_f"][5:9]] if len(z["meta_f"]) >= 9 else (1e-4, 1e-3, 1e-4, 1.0)
        opt = make_optimizer(solver, nabla_V_lr=lr_V, M_lr=lr_M, adam_eps=eps)
        logs = []
        # (one process: the same body, not captured either -- every iteration hands over a fresh noise tensor; over RCCL the iteration
        #  IS captured: the noise of every iteration is copied into one static buffer, as the graph tests do)
        graph = False if mode == "train_eager" else (True if sharded else "nocapture")
        static = torch.empty_like(aux["noise"][0][:, row0:row0 + Bl].contiguous()) if (real and aux["noise"].dim() == 4) else None
        tr = Trainer(solver, opt, B, normalization_const=norm0, sync_timing=False, hip_graph=graph, log=logs.append)
        count_bodies(tr)
        rec = dict(loss=[], weight_mean=[], weight_std=[], norm=[])
        n_it = int(z["train_iters"]) if "train_iters" in z.files else 3
        noise = aux["noise"] if aux["noise"].dim() == 4 else aux["noise"].unsqueeze(0).expand(n_it, -1, -1, -1)
        c0 = solver.shard.collectives if sharded else 0
        for it in range(n_it):
            if static is not None:
                static.copy_(noise[it][:, row0:row0 + Bl])
                solver.noise_in = static
            else:
                solver.noise_in = noise[it][:, row0:row0 + Bl].contiguous()
            o = tr.step()
            for k, v in (("loss", o["loss"]), ("weight_mean", o["weight_mean"]), ("weight_std", o["weight_std"]),
                         ("norm", tr.normalization_const)):
                rec[k].append(float(v))
        res["collectives_in_steps"] = (solver.shard.collectives - c0) if sharded else 0
        tr.join()
        torch.cuda.synchronize()
        res["collectives_after_join"] = (solver.shard.collectives - c0) if sharded else 0
        res["captured"] = len([k for k in tr._graphs if isinstance(k, tuple) and k and k[0] == "manual"])
        res.update(rec=rec, iters=n_it, bodies=bodies, capture_graphs=bool(tr.capture_graphs), hip_graph=bool(tr.hip_graph),
                   logs=logs, manual_ok=bool(tr._manual_ok({})))
        arrays.update({"V." + k: _np(v) for k, v in sde.nabla_V.state_dict().items()})
        arrays.update({"M." + k: _np(v) for k, v in sde.M.state_dict().items()})
        arrays["gamma"] = _np(sde.gamma)
    elif mode == "philox":
        # no injected noise: Philox keyed by the GLOBAL row.  (a) one rollout at (seed, offset): this rank's rows must be the
        # bits of the same rows of a one-process launch; (b) Trainer iterations on the device-resident key
        x0 = aux["x0"].repeat(Bl, 1)
        roll = R.stochastic_trajectories(sde, x0, aux["ts"], aux["lmbd"], seed=1234, offset=7, row0=row0)
        arrays["states"], arrays["noises"], arrays["lpd"] = _np(roll[0]), _np(roll[1]), _np(roll[4])
        solver.philox_key = PhiloxKey(dev, seed=99, offset=11)
        opt = make_optimizer(solver, M_lr=1e-3)
        tr = Trainer(solver, opt, B, normalization_const=0.05, sync_timing=False, hip_graph="nocapture")
        count_bodies(tr)
        rec = []
        for _ in range(4):
            o = tr.step()
            rec.append([float(o["loss"]), float(o["weight_mean"]), float(o["weight_std"]), float(tr.normalization_const)])
        tr.join()
        torch.cuda.synchronize()
        res.update(rec=rec, bodies=bodies, key=solver.philox_key.key.cpu().tolist(),
                   saved=bool((tr._dev or {}).get("saved") is not None))
        arrays.update({"V." + k: _np(v) for k, v in sde.nabla_V.state_dict().items()})
        arrays.update({"M." + k: _np(v) for k, v in sde.M.state_dict().items()})
        arrays["gamma"] = _np(sde.gamma)
    elif mode == "loss":
        # SOC_Solver.loss called directly on a shard (weight statistics by all_gather, the stopping-time normaliser by a scalar
        # all-reduce BEFORE the backward), then the flat gradient all-reduce
        solver.noise_in = aux["noise"][:, row0:row0 + Bl].contiguous()
        c0 = solver.shard.collectives if sharded else 0
        o = solver.loss(B, algorithm="SOCM", use_warm_start=False, use_stopping_time=aux["stopping"])
        res["collectives_in_loss"] = (solver.shard.collectives - c0) if sharded else 0
        o[0].backward()
        params = list(sde.nabla_V.parameters()) + list(sde.M.sigmoid_layers.parameters()) + [sde.gamma]
        if aux["stopping"]:
            params.append(sde.gamma2)
        obj = o[0].detach()
        if sharded:
            (obj,) = solver.shard.allreduce_gradients(params, extra=[obj])
        res.update(objective=float(obj), w_mean=float(o[5]), w_std=float(o[6]))
        for i, p in enumerate(params):
            arrays[f"g{i}"] = _np(p.grad)
    elif mode == "ctrl":
        # method.py:185-221 / utils.py:131-231 over a shard: the rows of the evaluation burst are split over the ranks, mean and
        # standard error pooled
        n = 3 * 40
        g = torch.Generator().manual_seed(5)
        noise = torch.randn(aux["K"], n, aux["d"], generator=g).to(dev)
        mean, err, traj = solver.control_objective(40, total_n_samples=n, noise_in=noise)
        res.update(mean=float(mean), err=float(err), traj_rows=int(traj.shape[1]))
    else:
        raise SystemExit(f"unknown mode {mode}")
    np.savez(out_path + f".rank{rank}.npz", **arrays)
    with open(out_path + f".rank{rank}.json", "w") as f:
        json.dump(res, f)
    if sharded:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
