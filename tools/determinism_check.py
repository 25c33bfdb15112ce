#!/usr/bin/env python3
"""Run-to-run determinism of the hot path on one GPU: the same rollout (same Philox key) and the same first Trainer iteration, repeated
with other work in between (dirty LDS / caches), compared BITWISE.  python tools/determinism_check.py [fixture]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "soc-matching_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from test_host_cpu import build_sde
from SOC_matching.method import SOC_Solver
from socmx import rollout as R
from socmx.rollout import PhiloxKey
from socmx.train import Trainer, make_optimizer

name = sys.argv[1] if len(sys.argv) > 1 else "tiny_double_well_d10"
dev = "cuda:0"


def dirty():
    # something else on the chip: big matmuls + an LDS-heavy kernel of ours (another fixture's rollout)
    a = torch.randn(4096, 4096, device=dev)
    (a @ a).sum().item()
    sde, aux = build_sde("cfg3_double_well_d10_K200", dev)
    R.hip_trajectories(sde, aux["x0"].repeat(aux["B"], 1), aux["ts"], aux["lmbd"], seed=123)
    torch.cuda.synchronize()


def one_rollout():
    sde, aux = build_sde(name, dev)
    out = R.hip_trajectories(sde, aux["x0"].repeat(aux["B"], 1), aux["ts"], aux["lmbd"], seed=7, offset=0, want_nabla_v=True)
    out2 = R.hip_trajectories(sde, aux["x0"].repeat(aux["B"], 1), aux["ts"], aux["lmbd"], noise_in=aux["noise"])
    torch.cuda.synchronize()
    return [o.detach().cpu().numpy().copy() for o in list(out) + list(out2)]


def one_iteration(hip_graph, n=3, sharded=False):
    sde, aux = build_sde(name, dev)
    solver = SOC_Solver(sde, aux["x0"], None, T=aux["T"], num_steps=aux["K"], lmbd=aux["lmbd"], d=aux["d"], sigma=sde.sigma)
    if sharded:
        from socmx.dist import Shard
        solver.shard = Shard()
    solver.philox_key = PhiloxKey(torch.device("cuda", 0), seed=9, offset=0)
    tr = Trainer(solver, make_optimizer(solver, M_lr=1e-3), aux["B"], normalization_const=0.03, sync_timing=False, hip_graph=hip_graph)
    rec = []
    for _ in range(n):
        info = tr.step()
        rec.append([float(info["loss"]), float(info["weight_mean"]), float(info["weight_std"])])
    tr.join(); torch.cuda.synchronize()
    return np.array(rec), [v.detach().cpu().numpy().copy() for v in sde.state_dict().values()]


bad = 0
ref_roll = one_rollout()
for rep in range(6):
    dirty()
    got = one_rollout()
    for i, (a, b) in enumerate(zip(ref_roll, got)):
        if not np.array_equal(a, b, equal_nan=True):
            bad += 1
            print(f"rollout output {i} differs in run {rep}: max |diff| {np.abs(a - b).max():.3e} at {np.argwhere(a != b)[:4].tolist()}")
SHARD = os.environ.get("DET_SHARD") == "1"
if SHARD:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29517")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
for mode in (False, True):
    ref_rec, ref_par = one_iteration(mode, sharded=SHARD)
    for rep in range(4):
        dirty()
        rec, par = one_iteration(mode, sharded=SHARD)
        if not np.array_equal(ref_rec[:, 1:], rec[:, 1:]):
            bad += 1
            print(f"hip_graph={mode}: WEIGHT STATISTICS differ in run {rep}:\n{ref_rec}\n{rec}")
        elif not np.array_equal(ref_rec, rec):
            bad += 1
            print(f"hip_graph={mode}: records differ in run {rep}:\n{ref_rec}\n{rec}")
        dp = max(float(np.abs(a - b).max()) for a, b in zip(ref_par, par))
        if dp != 0.0:
            bad += 1
            print(f"hip_graph={mode}: parameters differ in run {rep}: max |diff| {dp:.3e}")
print("determinism:", "OK" if bad == 0 else f"{bad} differences")
if SHARD:
    dist.barrier(); dist.destroy_process_group()
