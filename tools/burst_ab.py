#!/usr/bin/env python3
"""Developer A/B of the evaluation-burst rollout (65,536 rows in one launch): python3 tools/burst_ab.py
Runs itself once per SOCMX_BURST_ROWS value (the switch is read once per process), times the launch and compares the
per-row costs of a 4,112-row launch (257 tiles, ragged tail) between the forms."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")

if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path[:0] = [ROOT, os.path.join(ROOT, "soc-matching_amd"), os.path.join(ROOT, "tests")]
    import torch, contextlib, io
    from socmx.config import load_config
    from socmx.settings import define_variables
    from socmx import rollout
    tag = sys.argv[2]
    dev = torch.device("cuda:0")
    for setting, d, K in (("double_well", 10, 200), ("molecular_dynamics", 1, 150), ("OU_quadratic_easy", 2, 50)):
        cfg = load_config([f"method.setting={setting}", f"method.d={d}", f"method.num_steps={K}"])
        cfg.method.device = "cuda:0"
        torch.manual_seed(0)
        ts = torch.linspace(0, 1.0, K + 1).to(dev)
        with contextlib.redirect_stdout(io.StringIO()):
            x0, sigma, opt_sde, sde, _ = define_variables(cfg, ts)
        small = rollout.hip_trajectories(sde, x0.reshape(1, -1).expand(4107, -1).contiguous(), ts, 1.0, seed=5, offset=2)
        torch.save([t.cpu() if t is not None else None for t in small], os.path.join(OUT, f"burst_{setting}_{tag}.pt"))
        for costs_only in (False, True):
            big = x0.reshape(1, -1).expand(65536, -1).contiguous()
            run = lambda i: rollout.hip_trajectories(sde, big, ts, 1.0, seed=1, offset=i, costs_only=costs_only)
            run(0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for i in range(3):
                run(1 + i)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 3
            print(f"[{tag}] {setting} d={d} K={K} 65536 rows costs_only={costs_only}: {ms:.2f} ms, {65536 * K / ms / 1e3:.1f} M traj-steps/s")
            del big
            torch.cuda.empty_cache()
    sys.exit(0)

import torch
os.makedirs(OUT, exist_ok=True)
tags = sys.argv[1:] or ["16", "32"]
for tag in tags:
    env = dict(os.environ, SOCMX_BURST_ROWS=tag)
    subprocess.run([sys.executable, __file__, "child", tag], env=env, check=True)
if len(tags) == 2:
    for setting in ("double_well", "molecular_dynamics", "OU_quadratic_easy"):
        a = torch.load(os.path.join(OUT, f"burst_{setting}_{tags[0]}.pt"))
        b = torch.load(os.path.join(OUT, f"burst_{setting}_{tags[1]}.pt"))
        worst = 0.0
        same = True
        for x, y in zip(a, b):
            if x is None:
                continue
            same = same and torch.equal(x, y)
            worst = max(worst, float((x - y).abs().max() / (1.0 + y.abs().max())))
        print(f"{setting}: bit-identical = {same}")
        os.remove(os.path.join(OUT, f"burst_{setting}_{tags[0]}.pt"))
        os.remove(os.path.join(OUT, f"burst_{setting}_{tags[1]}.pt"))
        print(f"{setting}: {tags[0]}-row vs {tags[1]}-row workgroups, worst relative difference over the 8-tuple {worst:.2e}")
