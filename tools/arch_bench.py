#!/usr/bin/env python3
"""Non-default arch.hdims: rollout / control-network-backward / iteration timings with the variant library (constexpr kernels
for this architecture) and with the default library's descriptor-driven kernels.
    python tools/arch_bench.py 128 64 32 [d K B]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "soc-matching_amd"), os.path.join(ROOT, "tests")]
import torch
from socmx import _lib, nets, rollout as R
from SOC_matching.experiment_settings.double_well import DoubleWell
from SOC_matching.method import SOC_Solver
from socmx.train import Trainer, make_optimizer

hd = [int(v) for v in sys.argv[1:4]]
d, K, B = ([int(v) for v in sys.argv[4:7]] + [10, 200, 128])[:3] if len(sys.argv) > 4 else (10, 200, 128)
dev = "cuda:0"


def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for mode in ("variant", "generic"):
    torch.manual_seed(0)
    kappa, nu = torch.ones(d, device=dev), torch.ones(d, device=dev)
    sde = DoubleWell(device=dev, dim=d, hdims=hd, hdims_M=[128, 128], lmbd=1.0, kappa=kappa, nu=nu,
                     sigma=torch.eye(d, device=dev), gamma=6.0, scaling_factor_nabla_V=1.0, scaling_factor_M=0.1)
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        sde.initialize_models()
    net = sde.nabla_V
    if mode == "generic":
        net.hip_lib = lambda: _lib.lib()
    elif net.hip_lib() is _lib.lib() and tuple(hd) != (256, 128, 64):
        print("no variant library for", hd, "(make -C soc-matching_amd/csrc VARIANT=...)")
        continue
    ts = torch.linspace(0, 1, K + 1, device=dev)
    x0 = torch.zeros(B, d, device=dev)
    t_roll = timeit(lambda: R.stochastic_trajectories(sde, x0, ts, 1.0, seed=0))
    N = (K + 1) * B
    x, gout = torch.randn(N, d, device=dev), torch.randn(N, d, device=dev)
    t_bwd = timeit(lambda: nets.unet_backward_hip(net, x, ts, B, gout)) if nets.unet_backward_supported(net, N) else float("nan")
    solver = SOC_Solver(sde, torch.zeros(d, device=dev), None, T=1.0, num_steps=K, lmbd=1.0, d=d, sigma=sde.sigma)
    tr = Trainer(solver, make_optimizer(solver, M_lr=1e-3), B, sync_timing=False, hip_graph=True)
    for _ in range(4): tr.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): tr.step()
    torch.cuda.synchronize()
    t_it = 1e3 * (time.perf_counter() - t0) / 10
    print(f"hdims {hd} d={d} K={K} B={B} [{mode:7s}]: rollout {t_roll:.3f} ms   control-network backward {t_bwd:.3f} ms   "
          f"iteration (hipGraph) {t_it:.3f} ms")
