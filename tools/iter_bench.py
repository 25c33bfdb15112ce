#!/usr/bin/env python3
"""Iteration timing for one config in one mode (for rocprofv3 runs): python tools/iter_bench.py cfg2|cfg3|cfg5r|ouq20 eager|graph [n]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "soc-matching_amd")]
import torch, bench
from socmx.train import Trainer, make_optimizer
CFG = {"cfg2": ("OU_quadratic_easy", 2, 50, 2.0, 128), "cfg3": ("double_well", 10, 200, 6.0, 128),
       "cfg5r": ("OU_linear", 64, 400, 2.0, 512), "ouq20": ("OU_quadratic_easy", 20, 50, 2.0, 128),
       "cfg4r": ("double_well", 10, 200, 6.0, 1024), "md": ("molecular_dynamics", 1, 150, 2.0, 64),
       "cfg3b256": ("double_well", 10, 200, 6.0, 256), "cfg3b192": ("double_well", 10, 200, 6.0, 192)}
name, mode = sys.argv[1], sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 20
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
setting, d, K, gamma, B = CFG[name]
if name == "md":      # README: molecular_dynamics d=1, num_steps=150, batch 64, stopping times, arch.hdims_M=[64,64]
    from socmx.config import load_config
    from socmx.settings import define_variables
    from SOC_matching.method import SOC_Solver
    import contextlib, io
    cfg = load_config(["method.setting=molecular_dynamics", "method.d=1", "method.num_steps=150", "method.gamma=2.0",
                       "method.gamma2=2.0", "method.gamma3=2.0", "method.scaling_factor_M=0.1", "optim.M_lr=1e-3",
                       "optim.batch_size=64", "method.use_stopping_time=True", "arch.hdims_M=[64,64]"])
    cfg.method.device = str(dev)
    torch.manual_seed(0)
    ts = torch.linspace(0, cfg.method.T, 151).to(dev)
    with contextlib.redirect_stdout(io.StringIO()):
        x0, sigma, _, sde, _ = define_variables(cfg, ts)
    solver = SOC_Solver(sde, x0, None, T=cfg.method.T, num_steps=150, lmbd=1.0, d=1, sigma=sigma)
else:
    cfg, ts, x0, sde, solver = bench.build(dev, setting, d, K, gamma, B)
opt = make_optimizer(solver, nabla_V_lr=1e-4, M_lr=1e-3, adam_eps=1e-4)
tr = Trainer(solver, opt, batch_size=B, normalization_const=1.0, sync_timing=False, gemm_select=True, hip_graph=(mode == "graph"))
for _ in range(4):
    tr.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    tr.step()
torch.cuda.synchronize()
print(f"{name} {mode}: {1e3 * (time.perf_counter() - t0) / n:.3f} ms/iteration")
