#!/usr/bin/env python3
"""Developer check of the packed-fma one-row rollout (csrc/socmx_rollout1p.hip): every default-width, sigma = I, d <= 15
fixture through it (SOCMX_TILE_ROWS=1), against the reference's 8-tuple in the fixture; then its time at configs[2] beside the
v_fmac_f32_dpp form's (SOCMX_R1_FORM=dpp, in a child process: the switch is read once).
    python3 tools/r1p_check.py [time-only]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "soc-matching_amd"), os.path.join(ROOT, "tests")]
os.environ.setdefault("SOCMX_TILE_ROWS", "1")
import numpy as np
import torch

from test_host_cpu import build_sde, GOLDEN
from SOC_matching import utils

DEV = "cuda:0"


def check(name):
    sde, aux = build_sde(name, DEV)
    if aux["d"] > 15 or list(sde.nabla_V.hdims) != [256, 128, 64]:
        return None
    z, B = aux["z"], aux["B"]
    r = utils.stochastic_trajectories(sde, aux["x0"].repeat(B, 1), aux["ts"], aux["lmbd"], noise_in=aux["noise"])
    torch.cuda.synchronize()
    names = ["states", "noises", "stop_indicators", "fractional_timesteps", "lpd", "lps", "ltw", "controls"]
    worst = {}
    for n, v in zip(names, r):
        want = z["roll_" + n]
        worst[n] = float(np.abs(v.cpu().numpy() - want).max())
    return worst


def timing():
    import bench
    cfg, ts, x0, sde, solver = bench.build(torch.device(DEV))
    from socmx import rollout
    state0 = x0.repeat(128, 1)
    for i in range(5):
        rollout.stochastic_trajectories(sde, state0, ts, 1.0, seed=0, offset=i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for i in range(50):
        rollout.stochastic_trajectories(sde, state0, ts, 1.0, seed=0, offset=10 + i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 50


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "time-only":
        print(f"{timing():.4f}")
        sys.exit(0)
    names = sorted(f[:-4] for f in os.listdir(GOLDEN) if f.endswith(".npz") and not f.startswith(("train_", "gt_", "dwpde")))
    bad = 0
    for name in names:
        try:
            w = check(name)
        except Exception as e:  # noqa: BLE001
            print(f"{name}: skipped ({type(e).__name__}: {str(e)[:80]})")
            continue
        if w is None:
            continue
        ok = all(v < 2e-4 for v in w.values())
        bad += not ok
        print(f"{'ok ' if ok else 'BAD'} {name}: " + " ".join(f"{k}={v:.2e}" for k, v in w.items()))
    ms = timing()
    env = dict(os.environ, SOCMX_R1_FORM="dpp")
    old = subprocess.run([sys.executable, __file__, "time-only"], env=env, capture_output=True, text=True)
    print(f"configs[2] rollout (double_well d=10 K=200 B=128): packed-fma form {ms:.4f} ms; v_fmac_f32_dpp form {old.stdout.strip()} ms")
    sys.exit(1 if bad else 0)
