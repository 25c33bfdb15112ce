#!/usr/bin/env python3
"""Developer fuzz of the small-batch (4-row) rollout kernels: random settings / d / B / K at the default widths against the
device-agnostic eager path on the same injected noise.   python tools/fuzz_r4.py [n_cases] [seed]"""
import os, sys, contextlib, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "soc-matching_amd")]
import numpy as np, torch
from socmx.config import load_config
from socmx.settings import define_variables
from socmx import rollout as R

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
DEV = "cuda:0"
SETTINGS = ["OU_quadratic_easy", "OU_quadratic_hard", "OU_linear", "double_well", "molecular_dynamics"]
bad = 0
for case in range(n_cases):
    setting = SETTINGS[rng.integers(len(SETTINGS))]
    d = int(rng.choice([1, 2, 3, 5, 8, 11, 15, 16, 17, 20, 24, 31, 33, 48, 64]))
    if setting == "molecular_dynamics":
        d = min(d, 31)
    B = int(rng.choice([1, 2, 3, 4, 5, 7, 16, 31, 64, 100, 128, 255, 256]))
    K = int(rng.integers(1, 14)) if setting.startswith("OU") else int(rng.integers(20, 60))
    extra = ["method.use_stopping_time=True", "method.T=2.0", "method.lmbd=2.0"] if setting == "molecular_dynamics" else []
    cfg = load_config([f"method.setting={setting}", f"method.d={d}", f"method.num_steps={K}", "arch.hdims=[256,128,64]"] + extra)
    cfg.method.device = DEV
    torch.manual_seed(case)
    T = float(cfg.method.T)
    ts = torch.linspace(0, T, K + 1).to(DEV)
    with contextlib.redirect_stdout(io.StringIO()):
        x0, sigma, opt_sde, sde, _ = define_variables(cfg, ts)
    lmbd = float(cfg.method.lmbd)
    noise = torch.randn(K, B, d, generator=torch.Generator().manual_seed(1000 + case)).to(DEV)
    state0 = x0.repeat(B, 1) + 0.05 * torch.randn(B, d, generator=torch.Generator().manual_seed(case)).to(DEV)
    try:
        got = R.hip_trajectories(sde, state0, ts, lmbd, noise_in=noise)
        with torch.no_grad():
            want = R.eager_trajectories(sde, state0, ts, lmbd, noise_in=noise)
        if not all(torch.isfinite(t).all() for t in want):
            print(f"skip {setting:20s} d={d:2d} B={B:3d} K={K:2d}  (the eager path itself overflows on this draw)", flush=True)
            continue
        # a stopping decision is a sign test on fp32 values -- and the re-interpolation (utils.py:49-75) puts x_0 on the
        # boundary up to rounding, so `-x_0 > 0` right behind it can go either way (observed: x_0 == 0.0 exactly in one of the
        # two paths at every differing decision, ~1 % of rows, the same rows for both tile shapes): such rows are counted, not
        # compared
        same = (got[2] == want[2]).all(0)                       # (B,) rows with identical stop indicators on the whole grid
        flipped = int((~same).sum())
        worst = 0.0
        for a, b in zip(got, want):
            a, b = a.double(), b.double()
            rows = same if a.dim() == 1 else same.reshape(1, -1, *([1] * (a.dim() - 2))).expand_as(a)
            err = ((a - b).abs() / (1e-4 + 2e-4 * b.abs()))
            err = torch.where(rows, err, torch.zeros_like(err))
            worst = max(worst, float(err.max()))
        ok = worst <= 1.0 and flipped <= max(2, B // 20) and all(torch.isfinite(t).all() for t in got)
        if flipped:
            # how close to the boundary x_0 = 0 was the state at the first step where the decisions differ?
            near = []
            for b_ in torch.nonzero(~same).flatten().tolist():
                k_ = int(torch.nonzero(got[2][:, b_] != want[2][:, b_]).flatten()[0])
                near.append(min(abs(float(want[0][k_, b_, 0])), abs(float(got[0][k_, b_, 0]))))
            worst = f"{worst} ({flipped} of {B} rows decide a stop differently; |x_0| there: " + \
                    ", ".join(f"{v:.1e}" for v in near) + ")"
    except Exception as e:                                     # (e.g. SOCMX_E_LDS for a shape that does not fit: reported)
        ok, worst = False, repr(e)[:80]
    bad += not ok
    print(f"{'ok  ' if ok else 'FAIL'} {setting:20s} d={d:2d} B={B:3d} K={K:2d}  worst tolerance ratio {worst}", flush=True)
print(f"{bad} of {n_cases} cases failed")
sys.exit(1 if bad else 0)
