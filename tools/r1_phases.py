#!/usr/bin/env python3
"""Per-phase cycles of the one-row rollout kernel (needs a library built with -DSOCMX_R1_PROF:
   make -C soc-matching_amd/csrc PROF=1 and SOCMX_LIB=soc-matching_amd/socmx/libsocmx_prof.so).  Prints, per wave of workgroup 0, the average cycles per step
   between the kernel's marks (s_memtime: 100 MHz ticks on this part are converted with the measured kernel time)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "soc-matching_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch, contextlib, io
from socmx.config import load_config
from socmx.settings import define_variables
from socmx import rollout
dev = torch.device("cuda:0")
# python tools/r1_phases.py [B [setting d K]]     e.g. 128 OU_quadratic_easy 20 50
setting, d, K, B, gamma = ("double_well", 10, 200, int(sys.argv[1]) if len(sys.argv) > 1 else 128, 6.0)
if len(sys.argv) > 4:
    setting, d, K = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
cfg = load_config([f"method.setting={setting}", f"method.d={d}", f"method.num_steps={K}", f"method.gamma={gamma}",
                   "method.scaling_factor_M=0.1", "optim.M_lr=1e-3", f"optim.batch_size={B}"])
cfg.method.device = "cuda:0"
torch.manual_seed(0)
ts = torch.linspace(0, 1.0, K + 1).to(dev)
with contextlib.redirect_stdout(io.StringIO()):
    x0, sigma, opt_sde, sde, _ = define_variables(cfg, ts)
state0 = x0.repeat(B, 1)
for _ in range(3):
    rollout.stochastic_trajectories(sde, state0, ts, 1.0, seed=0)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    rollout.stochastic_trajectories(sde, state0, ts, 1.0, seed=0)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
cyc = torch.zeros(8 * 16 + 64 * 64, dtype=torch.int64, device=dev)
rollout.hip_trajectories(sde, state0, ts, 1.0, seed=0, phase_cycles=cyc)
torch.cuda.synchronize()
c = cyc[:128].double().cpu().numpy().reshape(8, 16) / K
names = ["-", "S1+S2", "bar", "(S2)", "(bar)", "S3", "bar", "S4end", "S5p", "bar", "serial/noise", "bar", "out", "sde", "S4pairs135", "S4pairs246"]
tot = c.sum(1)
print(f"rollout {ms:.3f} ms = {ms*1e3/K:.3f} us/step; ticks per step per wave (sum {tot.mean():.1f}) -> one tick = {ms*1e6/K/tot.mean():.2f} ns")
scale = ms * 1e6 / K / tot.mean() * 2.4   # cycles at 2.4 GHz per tick
for w in range(8):
    print(f"wave {w}: " + " ".join(f"{n}={c[w, i] * scale:.0f}" for i, n in enumerate(names) if i < 16 and c[w, i] > 0))
