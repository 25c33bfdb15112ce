#!/usr/bin/env python3
"""Evaluation-burst throughput (65,536 rows, costs-only) per setting: python3 tools/burst_settings.py
Which kernel a row of the table takes follows from the shape: sigma = I or dense at d <= 15 -> the two-tile kernel
(csrc/socmx_rollout32.hip); everything else -> the 16-row kernel's general SDE step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "soc-matching_amd"), os.path.join(ROOT, "tests")]
import torch, contextlib, io
from socmx.config import load_config
from socmx.settings import define_variables
from socmx import rollout
dev = torch.device("cuda:0")
for setting, d, K in (("double_well", 10, 200), ("molecular_dynamics", 1, 150), ("OU_quadratic_easy", 2, 50), ("OU_linear", 10, 100),
                      ("OU_quadratic_easy", 20, 50), ("OU_quadratic_hard", 20, 50), ("OU_linear", 64, 50)):
    over = [f"method.setting={setting}", f"method.d={d}", f"method.num_steps={K}"]
    if setting == "molecular_dynamics":
        over += ["method.use_stopping_time=True"]
    cfg = load_config(over)
    cfg.method.device = "cuda:0"
    torch.manual_seed(0)
    ts = torch.linspace(0, 1.0, K + 1).to(dev)
    with contextlib.redirect_stdout(io.StringIO()):
        x0, sigma, opt_sde, sde, _ = define_variables(cfg, ts)
    big = x0.reshape(1, -1).expand(65536, -1).contiguous()
    run = lambda i: rollout.hip_trajectories(sde, big, ts, 1.0, seed=1, offset=i, costs_only=True)
    run(0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for i in range(3): run(1 + i)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    dims = [(d + 1, 256), (256, 128), (128, 64), (64, 64), (64, 128), (128, 128), (128, 256), (256, 256), (256, d), (d + 1, d)]
    fl = 2 * sum(a * b for a, b in dims)
    # what the kernels execute: with a 16-wide (d <= 16) or >= 64-wide output block the skip res_1 (256 x 256) is folded through
    # up_0 into a d x 256 product (include/socmx.h at socmx_unet_packed_floats)
    outp = (d + 15) // 16 * 16
    fx = fl - 2 * (256 * 256 - d * 256) if (outp == 16 or outp >= 64) else fl
    print(f"{setting:20s} d={d:2d} K={K:3d}  65,536 rows costs-only: {ms:6.2f} ms  {65536 * K / ms / 1e3:6.1f} M trajectory-steps/s  "
          f"executed {fx * 65536 * K / ms / 1e9:6.1f} TFLOP/s (network only) = {fx * 65536 * K / ms / 1e9 / 157.3:.2f} of the fp32 MFMA peak "
          f"(the reference network's flops: {fl * 65536 * K / ms / 1e9:6.1f} TFLOP/s)")
