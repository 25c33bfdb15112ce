#!/usr/bin/env python3
"""Training entry point with the reference's interface (reference main.py:33-481).

    python main.py method.setting=double_well method.d=10 method.num_steps=200 method.gamma=6.0 \
                   method.scaling_factor_M=0.1 optim.M_lr=1e-3 optim.batch_size=128 method.num_iterations=80000
    torchrun --standalone --nproc-per-node 8 main.py ... optim.batch_size=1024      # batch sharded over 8 GPUs

Uses Hydra (`@hydra.main(config_path="configs", config_name="soc")`) when it is installed, otherwise
the same `configs/soc.yaml` through PyYAML with Hydra-style `a.b=value` overrides.  Differences from the
reference: no warm start, no `multiagent_8`; without Hydra a multirun (`-m` with comma lists, the README's
command lines) runs its jobs one after the other; the NVML print at iteration 0 is replaced by a ROCm-safe memory report; with
WORLD_SIZE > 1 the global batch is sharded across ranks (socmx.dist) and only rank 0 prints and saves.
"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

import torch

from SOC_matching.utils import (control_objective, get_file_name, get_folder_name, normalization_constant,
                                save_results)
from SOC_matching.method import SOC_Solver
from SOC_matching.experiment_settings.settings import define_variables
from socmx.train import Trainer, compute_EMA, make_optimizer



def _graph_mode(v):
    """backend.hip_graph as Trainer takes it: True / False, or the strings "force" (also capture the losses measured faster on the
    eager iteration: SOCM_const_M, SOCM_exp, SOCM_adjoint).  ("sharded", rounds 4-5's opt-in for capturing a multi-rank iteration,
    is accepted and means True: a sharded run captures by default now -- its all-reduces are launches of the shard's own RCCL
    communicators, socmx/rccl.py.)"""
    return v if isinstance(v, str) and v in ("force", "nocapture") else bool(v)

def run(cfg):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    shard = None
    if cfg.method.use_gpu:
        cfg.method.device = "cuda:" + str(local_rank if world > 1 else cfg.method.device_number)
        torch.cuda.set_device(cfg.method.device)
    else:
        cfg.method.device = "cpu"
    if world > 1:
        import torch.distributed as dist
        from socmx.dist import Shard
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if not dist.is_initialized():             # (a multirun's second job finds the group of the first)
            dist.init_process_group("nccl" if cfg.method.use_gpu else "gloo")
        shard = Shard()
    if cfg.method.use_gpu and bool((cfg.get("backend", {}) or {}).get("specialize_arch", True)):
        # arch.hdims other than [256,128,64]: build (once, rank 0) / load the kernel variant with these widths as constants
        from socmx import _lib as _socmx_lib
        _socmx_lib.set_specialize(True)
        if world > 1:
            # one build per NODE (local rank 0; the others wait, then only load).  A failed build warns and falls back to
            # the descriptor-driven kernels inside variant(): no rank raises, so nobody is left waiting at the barrier
            if local_rank == 0:
                _socmx_lib.variant(list(cfg.arch.hdims))
            dist.barrier()
            if local_rank != 0:
                _socmx_lib.variant(list(cfg.arch.hdims), build=False)
    log = print if rank == 0 else (lambda *a, **k: None)
    log(cfg)
    torch.manual_seed(cfg.method.seed)          # every rank builds the same problem and the same initial weights
    algorithm = cfg.method.algorithm
    ts = torch.linspace(0, cfg.method.T, cfg.method.num_steps + 1).to(cfg.method.device)
    folder_name = get_folder_name(cfg)

    x0, sigma, optimal_sde, neural_sde, u_warm_start = define_variables(cfg, ts)
    ground_truth_control = optimal_sde.u if optimal_sde is not None else None
    B = cfg.optim.batch_size
    state0 = x0.repeat(B, 1)

    log("Estimating normalization constant and control L2 error for initial control...")
    n_norm = int(getattr(cfg.method, "n_batches_normalization", 512))
    normalization_const, normalization_const_std_error, norm_sqd_diff_mean = normalization_constant(
        neural_sde, state0, ts, cfg, n_batches_normalization=n_norm, ground_truth_control=ground_truth_control)
    log(f"Normalization_constant (mean and std. error): {normalization_const:5.8E} {normalization_const_std_error:5.8E}")
    if ground_truth_control is not None:
        log(f"Control L2 error for initial control: {norm_sqd_diff_mean / normalization_const}")
        mean, err = control_objective(optimal_sde, x0, ts, cfg.method.lmbd, B,
                                      total_n_samples=cfg.method.n_samples_control)
        log(f"Optimal control loss mean: {mean:5.10f}, Optimal control loss std. error: {err:5.10f}")

    solver = SOC_Solver(neural_sde, x0, ground_truth_control, T=cfg.method.T, num_steps=cfg.method.num_steps,
                        lmbd=cfg.method.lmbd, d=cfg.method.d, sigma=sigma)
    if algorithm == "SOCM_exp":                                          # main.py:166-171
        solver.gamma = torch.nn.Parameter(torch.tensor([cfg.method.gamma]).to(cfg.method.device))
    else:
        solver.gamma = cfg.method.gamma
    solver.shard = shard
    optimizer = make_optimizer(solver, nabla_V_lr=cfg.optim.nabla_V_lr, M_lr=cfg.optim.M_lr, adam_eps=cfg.optim.adam_eps,
                               algorithm=algorithm, y0_lr=cfg.optim.y0_lr)
    optimizer.zero_grad()
    backend = cfg.get("backend", {}) or {}
    trainer = Trainer(solver, optimizer, B, normalization_const=normalization_const, algorithm=algorithm,
                      gemm_select=bool(backend.get("gemm_select", False)),
                      tune_new_shapes=bool(backend.get("tune_new_shapes", False)),
                      hip_graph=_graph_mode(backend.get("hip_graph", True)), log=log,
                      history_rows=int(cfg.method.num_iterations) + 1,
                      save_activations=bool(backend.get("save_activations", True)))

    solver.algorithm = algorithm
    info = solver.training_info = {k: [] for k in (
        "time_per_iteration", "EMA_time_per_iteration", "loss", "EMA_loss", "norm_sqd_diff", "EMA_norm_sqd_diff",
        "weight_mean", "EMA_weight_mean", "weight_std", "EMA_weight_std", "grad_norm_sqd", "EMA_grad_norm_sqd",
        "sqd_norm_EMA_grad", "control_objective_mean", "control_objective_std_err", "control_objective_itr",
        "trajectories")}
    info["iteration_mode"] = []      # (not in the reference: which schedule each iteration took -- Trainer.step's `mode`)
    info["cfg"] = cfg
    compute_L2_error = ground_truth_control is not None
    every = cfg.method.compute_control_objective_every
    ema = {}
    n_it = cfg.method.num_iterations
    for itr in range(n_it):
        checkpoint = itr == 0 or itr % every == every - 1 or itr == n_it - 1
        norm_before = trainer.normalization_const
        step = trainer.step(compute_L2_error=compute_L2_error, optimal_control=ground_truth_control,
                            compute_control_objective=checkpoint, total_n_samples=cfg.method.n_samples_control,
                            verbose=(itr == 0 and rank == 0))
        out = step["out"]
        info["iteration_mode"].append(step.get("mode"))
        vals = dict(time_per_iteration=step["time_per_iteration"], loss=step["loss"], weight_mean=step["weight_mean"],
                    weight_std=step["weight_std"])
        if compute_L2_error:
            vals["norm_sqd_diff"] = (out[1] / step.get("norm_before", norm_before)).detach()
        for k in ("grad_norm_sqd", "EMA_grad_norm_sqd", "sqd_norm_EMA_grad"):      # main.py:408-410
            if k in step:
                info[k].append(step[k].detach())
        for k, v in vals.items():
            ema[k] = v if itr == 0 else compute_EMA(v, ema[k], EMA_coeff=0.01, itr=itr)
            info[k].append(v)
            info["EMA_" + k].append(ema[k])
        if itr % 10 == 0 or itr == n_it - 1:
            extra = f" {vals['norm_sqd_diff'].item():5.5f} {ema['norm_sqd_diff'].item():5.6f}" if compute_L2_error else ""
            log(f"{itr} - {vals['time_per_iteration']:5.3f}s/it (EMA {ema['time_per_iteration']:5.3f}s/it): "
                f"{vals['loss'].item():5.5f} {ema['loss'].item():5.5f}{extra} "
                f"{ema['weight_mean'].item():5.6E} {ema['weight_std'].item():5.6E}")
            if algorithm == "moment":
                log(f"soc_solver.y0: {solver.y0.item()}")
            elif algorithm == "SOCM_exp":
                log(f"soc_solver.gamma: {solver.gamma.item()}")
            elif algorithm == "SOCM":
                log(f"soc_solver.neural_sde.M.gamma: {neural_sde.M.gamma.item()}")
        if algorithm == "moment" and itr == 5000:                        # main.py:437-441
            optimizer.param_groups[-1]["lr"] = 1e-4
        if checkpoint:
            log(f"Control loss mean: {out[2]:5.5f}, Control loss std. error: {out[3]:5.5f}")
            info["control_objective_mean"].append(out[2].detach())
            info["control_objective_std_err"].append(out[3].detach())
            info["control_objective_itr"].append(itr + 1)
            info["trajectories"].append(out[4])
            solver.num_iterations = itr + 1
            if rank == 0:
                trainer.join()                               # second-stream M update lands before we read it
                keep, solver.shard = solver.shard, None      # process groups do not pickle
                save_results(solver, folder_name, get_file_name(folder_name, num_iterations=itr + 1))
                save_results(solver, folder_name, get_file_name(folder_name, num_iterations=itr + 1, last=True))
                solver.shard = keep
    # (the reference prints seconds with three decimals, main.py:421: a millisecond iteration reads "0.001s/it"; one more line
    #  with the resolution this backend needs -- steady-state iterations only: no checkpoint bursts, no eager warm-up / capture)
    steady = [t for i, t in enumerate(info["time_per_iteration"])
              if i >= min(10, n_it // 2) and not (i == 0 or i % every == every - 1 or i == n_it - 1)]
    if steady:
        steady.sort()
        modes = {}
        for i, m in enumerate(info["iteration_mode"]):
            if i >= min(10, n_it // 2) and not (i == 0 or i % every == every - 1 or i == n_it - 1):
                modes[m] = modes.get(m, 0) + 1
        log(f"time_per_iteration: median {1e3 * steady[len(steady) // 2]:.3f} ms over {len(steady)} steady-state iterations "
            f"(modes: {', '.join(f'{k} x{v}' for k, v in sorted(modes.items(), key=lambda kv: -kv[1]))})")
    if shard is not None and world > 1:
        trainer.join()
        torch.cuda.synchronize()
        import torch.distributed as dist
        dist.barrier()
        shard.close()                # (the shard's own RCCL communicators; the process group stays up for a multirun's next job)
    return solver


def _main_fallback(argv=None):
    """No Hydra in this image: the same configs/soc.yaml through PyYAML with Hydra-style overrides.  `-m` / `--multirun` with
    comma lists runs the combinations ONE AFTER THE OTHER in ./outputs/multiruns/<job number> (soc.yaml's hydra.sweep.dir /
    subdir; the reference fans them out through the submitit launcher), so the README's command lines work verbatim."""
    from socmx.config import expand_multirun, load_config
    argv = sys.argv[1:] if argv is None else argv
    jobs, multirun = expand_multirun(argv)
    root = os.getcwd()
    solver = None
    for num, overrides in enumerate(jobs):
        cfg = load_config(overrides)
        out_dir = os.path.join(root, "outputs", "multiruns", str(num)) if multirun else os.path.join(root, "outputs", "runs")
        os.makedirs(out_dir, exist_ok=True)
        os.chdir(out_dir)                        # hydra.job.chdir: True (soc.yaml:45-52)
        if multirun:
            print(f"[multirun] job {num} of {len(jobs)}: {' '.join(overrides)}", flush=True)
        with open("cmd.sh", "w") as f:           # main.py:60-63
            f.write("#!/bin/bash\n\n" + " \\\n".join([f"python {sys.argv[0]}"] + ["\t" + x for x in overrides]) + "\n")
        try:
            solver = run(cfg)
        finally:
            os.chdir(root)
    return solver


if __name__ == "__main__":
    try:
        try:
            import hydra  # noqa: F401
            have_hydra = True
        except ImportError:
            have_hydra = False
        if have_hydra:
            import hydra

            @hydra.main(version_base=None, config_path="configs", config_name="soc")
            def _hydra_main(cfg):
                run(cfg)

            _hydra_main()
        else:
            _main_fallback()
    except Exception:
        import traceback

        print(traceback.format_exc())
        sys.exit(1)
