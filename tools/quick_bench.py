#!/usr/bin/env python3
"""Developer micro-benchmark: rollout / loss timings per config + a parity spot check."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "soc-matching_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch, contextlib, io
from socmx.config import load_config
from socmx.settings import define_variables
from socmx import rollout
from SOC_matching.method import SOC_Solver
from socmx.train import Trainer, make_optimizer

dev = torch.device("cuda:0")
CONFIGS = {
    "cfg2": ("OU_quadratic_easy", 2, 50, 128, 2.0),
    "cfg3": ("double_well", 10, 200, 128, 6.0),
    "cfg4r": ("double_well", 10, 200, 1024, 6.0),
    "cfg5r": ("OU_linear", 64, 400, 512, 2.0),
    "ouq20": ("OU_quadratic_easy", 20, 50, 128, 2.0),            # soc.yaml defaults (d = 20)
    "ouq20b": ("OU_quadratic_hard", 20, 50, 2048, 2.0),
    "burst": ("double_well", 10, 200, 65536, 6.0),
    "oul10": ("OU_linear", 10, 100, 64, 2.0),                    # README "Linear OU": dense sigma, d = 10
    "ouh20": ("OU_quadratic_hard", 20, 50, 128, 2.0),            # README "Quadratic OU hard" (method.d from soc.yaml: 20)
    "md": ("molecular_dynamics", 1, 150, 64, 2.0),
}
which = [a for a in sys.argv[1:] if not a.startswith("--")] or ["cfg3"]

def timeit(fn, n, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

# parity spot check on cfg3 fixture
from test_host_cpu import build_sde
from SOC_matching import utils
sde, aux = build_sde("cfg3_double_well_d10_K200", "cuda:0")
r = utils.stochastic_trajectories(sde, aux["x0"].repeat(aux["B"], 1), aux["ts"], aux["lmbd"], noise_in=aux["noise"])
err = float(np.abs(r[0].cpu().numpy() - aux["z"]["roll_states"]).max())
print(f"parity cfg3 fixture: max|states diff| = {err:.2e}", "OK" if err < 1e-4 else "FAIL")

for name in which:
    setting, d, K, B, gamma = CONFIGS[name]
    cfg = load_config([f"method.setting={setting}", f"method.d={d}", f"method.num_steps={K}", f"method.gamma={gamma}",
                       "method.scaling_factor_M=0.1", "optim.M_lr=1e-3", f"optim.batch_size={B}"])
    cfg.method.device = "cuda:0"
    torch.manual_seed(0)
    ts = torch.linspace(0, 1.0, K + 1).to(dev)
    with contextlib.redirect_stdout(io.StringIO()):
        x0, sigma, opt_sde, sde, _ = define_variables(cfg, ts)
    state0 = x0.repeat(B, 1)
    ms = timeit(lambda: rollout.stochastic_trajectories(sde, state0, ts, 1.0, seed=0), 10 if B < 10000 else 3)
    print(f"{name}: rollout {ms:.3f} ms  ({ms*1e3/K:.2f} us/step, {B*K/ms*1e3/1e6:.2f} M traj-steps/s)")
    if "--phases" in sys.argv:
        cyc = torch.zeros((B + 15) // 16, 64, dtype=torch.int64, device=dev)
        rollout.hip_trajectories(sde, state0, ts, 1.0, seed=0, phase_cycles=cyc)
        torch.cuda.synchronize()
        used = cyc[cyc.sum(1) > 0]          # (the two-tile burst kernel has half as many workgroups as the table has rows)
        c = used.double().mean(0).cpu().numpy() / K
        names = ["x0build", "down_0", "down_1", "down_2", "up2+res2", "up1+res1", "up0+res0", "ctrl+noise", "EM", "cost+wb"]
        print(f"{name}: cycles/step per phase: " + ", ".join(f"{n}={v:.0f}" for n, v in zip(names, c)) + f"  total={c[:10].sum():.0f}")
        # general SDE step only (wave 0's view): 10 = MFMA products of the Euler-Maruyama phase (8 = its epilogue +
        # barrier), 11 = x'Px partials / noise finish, 12 = u'u, u'eps and the owners' update, 14 = write-back
        # (9 = the barrier that ends the step + the owners' x'Px sum)
        print("   SDE-step slots 10 (EM products) 11 (x'Px, noise) 12 (u'u, u'eps) 14 (write-back) 15 (last barrier):",
              " ".join(f"{c[i]:.0f}" for i in (10, 11, 12, 14, 15)))
        for si in range(6):
            sub = c[16 + si * 8: 16 + si * 8 + 7]
            print(f"   stage {si+1} ({names[si+1]}): desc={sub[5]:.0f} prewait={sub[6]:.0f} issue={sub[0]:.0f} gemm1={sub[1]:.0f} gemm2={sub[2]:.0f} store+prefetch={sub[3]:.0f} barrier={sub[4]:.0f}")
    if name != "burst" and "--loss" in sys.argv or name in ("cfg2", "cfg3"):
        solver = SOC_Solver(sde, x0, None, T=1.0, num_steps=K, lmbd=1.0, d=d, sigma=sigma)
        opt = make_optimizer(solver, M_lr=1e-3)
        tr = Trainer(solver, opt, B, sync_timing=False)
        ms_it = timeit(lambda: tr.step(), 10)
        print(f"{name}: SOCM iteration {ms_it:.3f} ms ({1e3/ms_it:.1f} it/s)")
