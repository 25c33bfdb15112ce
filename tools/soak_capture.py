#!/usr/bin/env python3
"""Soak of the captured SHARDED iteration at world size 1 with the real RCCL backend: N cycles of (new solver + Trainer with a
shard -> 2 eager warm-up iterations -> capture with the ncclAllReduce launches of the shard's own communicators inside -> replays),
with eager process-group collectives issued right in front of every capture -- their Work objects are what torch's
ProcessGroupNCCL watchdog thread is still polling while the capture runs (the arrangement that took the process down in about one
of five full GPU test runs in rounds 4-5, when the capture stream came from torch's pool and the collectives were
ProcessGroupNCCL calls).  python tools/soak_capture.py [cycles] [fixture]   (SOCMX_RCCL=0: the same over torch's process group)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "soc-matching_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch, torch.distributed as dist
from test_host_cpu import build_sde
from SOC_matching.method import SOC_Solver
from socmx.dist import Shard
from socmx.rollout import PhiloxKey
from socmx.train import Trainer, make_optimizer

cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 200
name = sys.argv[2] if len(sys.argv) > 2 else "tiny_double_well_d10"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29519")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
shard = Shard()
print(f"transport {shard.transport} ({shard.transport_note}); fixture {name}; {cycles} cycles", flush=True)
t0 = time.time()
captured = replays = 0
first = None
for c in range(cycles):
    sde, aux = build_sde(name, "cuda:0")
    solver = SOC_Solver(sde, aux["x0"], None, T=aux["T"], num_steps=aux["K"], lmbd=aux["lmbd"], d=aux["d"], sigma=sde.sigma)
    solver.shard = shard
    solver.philox_key = PhiloxKey(dev, seed=9, offset=0)
    logs = []
    tr = Trainer(solver, make_optimizer(solver, M_lr=1e-3), aux["B"], normalization_const=0.03, sync_timing=False, hip_graph=True,
                 log=logs.append)
    rec = []
    for it in range(7):
        if it == 2:
            # eager process-group work immediately in front of the capture (NOT waited for on the host)
            for _ in range(3):
                dist.all_reduce(torch.ones(1024, device=dev))
            dist.barrier()
        rec.append(float(tr.step()["loss"]))
    tr.join()
    torch.cuda.synchronize()
    got = [k for k in tr._graphs if isinstance(k, tuple) and k and k[0] == "manual"]
    assert tr.hip_graph and len(got) == 1 and not logs, (c, logs, list(tr._graphs))
    captured += 1
    replays += 4
    if first is None:
        first = rec
    assert np.allclose(rec, first, rtol=1e-5, atol=0), (c, rec, first)   # same key, same weights: every cycle trains the same seven iterations
    if (c + 1) % 50 == 0:
        print(f"  {c + 1} cycles, {time.time() - t0:.0f} s", flush=True)
calls = sum(k.calls for k in shard._comms.values())
print(f"soak ok: {captured} captures, {replays} replays, {calls} RCCL calls enqueued or captured through the own communicators, "
      f"{time.time() - t0:.0f} s; losses of a cycle {['%.5f' % v for v in first]}")
dist.barrier()
shard.close()
dist.destroy_process_group()
