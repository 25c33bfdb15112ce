"""Reference-named helpers of SOC_matching/utils.py that sit on or next to the hot path.

stochastic_trajectories (utils.py:17-128) is the fused kernel / eager rollout of `socmx.rollout`;
control_objective (131-163) and normalization_constant (166-231) are the evaluation bursts that call
it `total_n_samples // batch_size` times; the run-naming / pickle helpers (261-386) and compute_EMA
(389-396) keep the reference's on-disk and telemetry conventions.
"""
import os
import pickle

import numpy as np
import torch

from socmx.rollout import stochastic_trajectories, burst_eligible  # noqa: F401
from socmx.train import compute_EMA  # noqa: F401
from socmx.ground_truth import riccati as solution_Ricatti_grid  # noqa: F401


def control_objective(sde, x0, ts, lmbd, batch_size, total_n_samples=65536, verbose=False):
    """Mean and standard error of the control cost -lmbd (lpd + ltw) over n_batches rollouts."""
    n_batches = int(total_n_samples // batch_size)
    if burst_eligible(sde, x0):
        # fused-kernel path: the n_batches rollouts are independent, so they run as ONE launch of
        # n_batches*batch_size rows (4096 workgroups at the defaults: the one regime where the chip is full)
        n = n_batches * batch_size
        out = stochastic_trajectories(sde, x0.reshape(1, -1).expand(n, -1), ts.to(x0), lmbd, verbose=verbose)
        costs = -lmbd * (out[4] + out[6])
        return torch.mean(costs), torch.std(costs) / np.sqrt(n - 1)
    costs = []
    for k in range(n_batches):
        out = stochastic_trajectories(sde, x0.repeat(batch_size, 1), ts.to(x0), lmbd, verbose=verbose)
        costs.append(-lmbd * (out[4] + out[6]))
        if k % 32 == 31:
            print(f"Batch {k+1}/{n_batches} done")
    costs = torch.cat(costs, 0)
    return torch.mean(costs), torch.std(costs) / np.sqrt(n_batches * batch_size - 1)


def normalization_constant(sde, x0, ts, cfg, n_batches_normalization=512, ground_truth_control=None):
    """E[w] over n_batches rollouts of the initial control (+ weighted L2 error vs a ground truth)."""
    if burst_eligible(sde, x0):
        # one launch for all batches (x0 here is the (B,d) repeated initial state, main.py:115)
        B = x0.shape[0]
        n = B * n_batches_normalization
        states, _, _, _, lpd, lps, ltw, controls = stochastic_trajectories(
            sde, x0.repeat(n_batches_normalization, 1), ts.to(x0), cfg.method.lmbd)
        lw = lpd + lps + ltw
        w = torch.exp(lw)
        err = None
        if ground_truth_control is not None:
            gt = ground_truth_control(ts, states, t_is_tensor=True)[:-1].detach()
            # mean over batches of sum(.)/(K*B)  ==  sum over all rows / (K*B*n_batches)
            err = torch.sum((gt - controls) ** 2 * w.reshape(1, -1, 1)) / (gt.shape[0] * n)
        print(f"Average and std. dev. of log_weights for all batches: {torch.mean(lw)} {torch.std(lw)}")
        return torch.mean(w), torch.std(w) / np.sqrt(n - 1), err
    logw, w_all = [], []
    err = 0 if ground_truth_control is not None else None
    for k in range(n_batches_normalization):
        states, _, _, _, lpd, lps, ltw, controls = stochastic_trajectories(sde, x0, ts.to(x0), cfg.method.lmbd)
        lw = lpd + lps + ltw
        w = torch.exp(lw)
        logw.append(lw)
        w_all.append(w)
        if ground_truth_control is not None:
            gt = ground_truth_control(ts, states, t_is_tensor=True)[:-1].detach()
            err = err + torch.sum((gt - controls) ** 2 * w.reshape(1, -1, 1) / (gt.shape[0] * gt.shape[1]))
        if k % 32 == 31:
            print(f"Batch {k+1}/{n_batches_normalization} done")
    if ground_truth_control is not None:
        err = err / n_batches_normalization
    logw, w_all = torch.stack(logw, dim=1), torch.stack(w_all, dim=1)
    print(f"Average and std. dev. of log_weights for all batches: {torch.mean(logw)} {torch.std(logw)}")
    n = w_all.shape[0] * w_all.shape[1]
    return torch.mean(w_all), torch.std(w_all) / np.sqrt(n - 1), err


def solution_Ricatti(R_inverse, A, P, Q, t):
    """utils.py:234-248 signature (R_inverse = sigma sigma^T is recomputed from sigma in socmx.ground_truth)."""
    F, out = Q, [Q]
    for t0, t1 in zip(t[:-1], t[1:]):
        F = F - (t1 - t0) * (-(A.T @ F) - F @ A + 2 * F @ R_inverse @ F - P)
        out.append(F)
    out.reverse()
    return torch.stack(out)


def optimal_control_LQ(sigma, A, P, Q, t):
    return -2 * torch.einsum("ij,bjk->bik", sigma.T, solution_Ricatti(sigma @ sigma.T, A, P, Q, t))


def exponential_t_A(t, A):
    return torch.matrix_exp(t.reshape(-1, 1, 1) * A.unsqueeze(0))


# ---- run naming and pickled checkpoints (utils.py:261-386) ---------------------------------------

def get_folder_name(cfg):
    m, o = cfg.method, cfg.optim
    parts = [m.algorithm, m.setting, m.lmbd, m.T, m.num_steps, m.use_warm_start, m.seed, o.batch_size, o.M_lr,
             o.nabla_V_lr]
    return "_".join(str(p) for p in parts)


def get_file_name(folder_name, num_iterations=0, last=False):
    if last:
        return folder_name + "/last.pkl"
    print(f"folder_name: {folder_name}")
    return folder_name + "/" + str(num_iterations) + ".pkl"


def save_results(results, folder_name, file_name):
    os.makedirs(folder_name, exist_ok=True)
    with open(file_name, "wb") as f:
        pickle.dump(results, f)


def retrieve_results(file_name):
    with open(file_name, "rb") as f:
        return pickle.load(f)
