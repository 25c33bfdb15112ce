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

from socmx.rollout import stochastic_trajectories, burst_eligible, burst_log_weights  # noqa: F401
from socmx.train import compute_EMA  # noqa: F401
from socmx.ground_truth import riccati as solution_Ricatti_grid  # noqa: F401


def _mean_and_std_err(x, shard=None):
    """(mean, std / sqrt(n - 1)) of the samples x -- pooled over the ranks of `shard` (each holds its own rows)."""
    if shard is None:
        n = x.numel()
        return torch.mean(x), torch.std(x) / np.sqrt(n - 1)
    mean_l = x.mean()
    stats = shard.combine_weight_stats(torch.stack([x.sum(), ((x - mean_l) ** 2).sum(),
                                                    torch.tensor(float(x.numel()), device=x.device)]))
    n = stats[2]
    return stats[0] / n, torch.sqrt(stats[1] / (n - 1)) / torch.sqrt(n - 1)


def control_objective(sde, x0, ts, lmbd, batch_size, total_n_samples=65536, verbose=False, *, noise_in=None,
                      shard=None, chunk_rows=16384):
    """Mean and standard error of the control cost -lmbd (lpd + ltw) over n_batches rollouts (utils.py:131-163).
    `noise_in` (K, n_batches*batch_size, d) injects the noise (parity runs); `shard` splits the rows over ranks."""
    n_batches = int(total_n_samples // batch_size)
    if burst_eligible(sde, x0):
        # fused-kernel path: the n_batches rollouts are independent rows -> costs-only launches of <= chunk_rows rows
        # (4096 workgroups at the defaults: the one regime where the chip is full)
        n = n_batches * batch_size
        n_loc, row0 = (n, 0) if shard is None else shard.local_rows(n)
        nz = None if noise_in is None else noise_in[:, row0:row0 + n_loc]
        lpd, _, ltw = burst_log_weights(sde, x0, ts.to(x0), lmbd, n_loc, noise_in=nz, row0=row0, chunk_rows=chunk_rows)
        return _mean_and_std_err(-lmbd * (lpd + ltw), shard)
    costs = []
    for k in range(n_batches):
        nz = None if noise_in is None else noise_in[:, k * batch_size:(k + 1) * batch_size]
        out = stochastic_trajectories(sde, x0.repeat(batch_size, 1), ts.to(x0), lmbd, verbose=verbose, noise_in=nz)
        costs.append(-lmbd * (out[4] + out[6]))
        if k % 32 == 31:
            print(f"Batch {k+1}/{n_batches} done")
    costs = torch.cat(costs, 0)
    return torch.mean(costs), torch.std(costs) / np.sqrt(n_batches * batch_size - 1)


def normalization_constant(sde, x0, ts, cfg, n_batches_normalization=512, ground_truth_control=None, *,
                           chunk_rows=16384):
    """E[w] over n_batches rollouts of the initial control (+ weighted L2 error vs a ground truth): utils.py:166-231."""
    if burst_eligible(sde, x0):
        # x0 here is the (B,d) repeated initial state (main.py:115); all batches are rows of chunked launches
        B = x0.shape[0]
        n = B * n_batches_normalization
        lmbd = cfg.method.lmbd
        if ground_truth_control is None:
            lw = sum(burst_log_weights(sde, x0[0], ts.to(x0), lmbd, n, chunk_rows=chunk_rows))
            w = torch.exp(lw)
            err = None
        else:
            # the L2 error needs states and controls: full outputs, chunk by chunk (peak memory = one chunk)
            from socmx import rollout as R
            seed, offset = torch.initial_seed(), R._philox_calls
            R._philox_calls += 1
            lws, err = [], 0.0
            for r0 in range(0, n, chunk_rows):
                r1 = min(n, r0 + chunk_rows)
                states, _, _, _, lpd, lps, ltw, controls = stochastic_trajectories(
                    sde, x0[0].reshape(1, -1).expand(r1 - r0, -1), ts.to(x0), lmbd, seed=seed, offset=offset, row0=r0)
                lw_c = lpd + lps + ltw
                gt = ground_truth_control(ts, states, t_is_tensor=True)[:-1].detach()
                # mean over batches of sum(.)/(K*B)  ==  sum over all rows / (K*B*n_batches)
                err = err + torch.sum((gt - controls) ** 2 * torch.exp(lw_c).reshape(1, -1, 1)) / (gt.shape[0] * n)
                lws.append(lw_c)
            lw = torch.cat(lws)
            w = torch.exp(lw)
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
