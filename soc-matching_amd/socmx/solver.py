"""`SOC_Solver`: reference SOC_matching/method.py:146-906, SOCM branch.

Same constructor, same `.loss(...)` keyword surface and 8-tuple return
`(objective, norm_sqd_diff, ctrl_loss_mean, ctrl_loss_std_err, trajectory,
mean(w), std(w), stop_indicators)` (method.py:223-236, 897-906), same
`.control_objective` (method.py:185-221).  On CUDA tensors `algorithm="SOCM"` runs on the HIP
loss kernels, with stopping times (molecular_dynamics, per-sample
TwoBoundarySigmoidMLP) on csrc/socmx_stopping.hip, and seven of the other eight losses of
method.py:264-478, 722-856 (row f4, `socmx.baselines`) on csrc/socmx_baselines.hip -- all on the
fused rollout's buffers, with nabla_V's values from the rollout kernel and its parameter gradients
from socmx_unet_backward_f32 (`rel_entropy` differentiates through the eager rollout instead).
CPU tensors take torch restatements of the same formulas.

Data parallelism: when `self.shard` is set (see socmx.dist), `batch_size` is the
GLOBAL batch; this rank simulates rows [row0, row0+B_local) and divides by the
global (K+1)*B so that summing gradients over ranks reproduces the single-GPU
gradient; mean/std of w are combined across ranks (by the Trainer inside its one flat
all-reduce; by an all_gather when `.loss()` is called directly).
"""
import numpy as np
import torch
import torch.nn as nn

from . import baselines
from . import loss as L
from . import rollout as R
from .nets import SigmoidMLP

ALGORITHMS = ("SOCM", "SOCM_const_M", "SOCM_exp", "SOCM_adjoint", "cross_entropy", "log-variance", "variance",
              "moment", "rel_entropy")


class SOC_Solver(nn.Module):
    noise_type = "diagonal"
    sde_type = "ito"

    def __init__(self, neural_sde, x0, ut, T=1.0, num_steps=100, lmbd=1.0, d=2, sigma=None):
        super().__init__()
        self.dim = neural_sde.dim
        self.neural_sde = neural_sde
        self.x0 = x0
        self.ut = ut
        self.T = T
        self.ts = torch.linspace(0, T, num_steps + 1).to(x0.device)
        self.num_steps = num_steps
        self.dt = T / num_steps
        self.lmbd = lmbd
        self.d = d
        self.y0 = torch.nn.Parameter(torch.randn(1, device=x0.device))  # RNG parity with method.py:172
        self.sigma = sigma if sigma is not None else torch.eye(d)
        self.shard = None        # socmx.dist.Shard or None
        self.noise_in = None     # test hook: (K, B_local, d) noise injected into the next loss() call

    # ---- method.py:175-183 -------------------------------------------------------------------
    def control(self, t0, x0):
        x0 = x0.reshape(-1, self.dim)
        tx = torch.cat([t0.reshape(-1, 1).expand(x0.shape[0], 1), x0], dim=-1)
        return -(self.neural_sde.nabla_V(tx) @ self.sigma)

    # ---- method.py:185-221 -------------------------------------------------------------------
    def _pair_grid(self, ts, K):
        """(t, s, i, j, s-t) of the (t_i <= s_j) pairs; the time grid is fixed, so this is built once per device."""
        key = (K, ts.device, ts.dtype)
        cache = self.__dict__.setdefault("_pair_grid_cache", {})
        if key not in cache:
            t_vec, s_vec, ii, jj = L.pair_times(ts, self.T, K)
            cache[key] = (t_vec, s_vec, ii, jj, (s_vec - t_vec).contiguous())
        return cache[key]

    def __getstate__(self):   # HIP streams and derived caches are per-process: rebuilt on demand after unpickling
        st = self.__dict__.copy()
        st.pop("_side_streams", None)
        st.pop("_pair_grid_cache", None)
        st.pop("_pending_M", None)
        st.pop("_stop_grid", None)
        return st

    def _side_stream(self, device, which=0):
        if not getattr(self, "overlap_M", True):
            return None
        streams = self.__dict__.setdefault("_side_streams", {})
        if (device, which) not in streams:
            # (a stream of this package's own, never one of torch's pooled ones: it joins hipGraph captures -- socmx/streams.py)
            from .streams import private_stream
            streams[(device, which)] = private_stream(device, f"solver{which}")
        return streams[(device, which)]

    def control_objective(self, batch_size, total_n_samples=65536, noise_in=None):
        n_batches = int(total_n_samples // batch_size)
        x0row = self.x0.reshape(1, -1)
        if R.burst_eligible(self.neural_sde, x0row):
            # the first batch as an ordinary launch (its states are the returned `trajectory`, as in the reference),
            # every other row through costs-only launches; rows are split over the ranks of a sharded run
            from SOC_matching.utils import _mean_and_std_err
            n = n_batches * batch_size
            seed, offset = torch.initial_seed(), R._philox_calls
            R._philox_calls += 1
            n_loc, row0 = (n, 0) if self.shard is None else self.shard.local_rows(n)
            nz = None if noise_in is None else noise_in[:, row0:row0 + n_loc]
            b0 = min(batch_size, n_loc)
            first = R.stochastic_trajectories(self.neural_sde, x0row.expand(b0, -1), self.ts, self.lmbd, seed=seed,
                                              offset=offset, row0=row0, noise_in=None if nz is None else nz[:, :b0])
            lpd, ltw = [first[4]], [first[6]]
            if n_loc > b0:
                rest = R.burst_log_weights(self.neural_sde, x0row, self.ts, self.lmbd, n_loc - b0, seed=seed,
                                           offset=offset, row0=row0 + b0, noise_in=None if nz is None else nz[:, b0:])
                lpd.append(rest[0])
                ltw.append(rest[2])
            losses = -self.lmbd * (torch.cat(lpd) + torch.cat(ltw))
            mean, err = _mean_and_std_err(losses, self.shard)
            return mean, err, first[0]
        losses, trajectory = [], None
        for k in range(n_batches):
            state0 = self.x0.repeat(batch_size, 1)
            nz = None if noise_in is None else noise_in[:, k * batch_size:(k + 1) * batch_size]
            out = R.stochastic_trajectories(self.neural_sde, state0, self.ts.to(state0), self.lmbd, noise_in=nz)
            losses.append(-self.lmbd * (out[4] + out[6]))
            if k == 0:
                trajectory = out[0]
            if k % 32 == 31:
                print(f"Batch {k+1}/{n_batches} done")
        losses = torch.cat(losses, 0)
        n = n_batches * batch_size
        return torch.mean(losses), torch.std(losses) / np.sqrt(n - 1), trajectory

    # ---- stopping-time SOCM (molecular_dynamics): method.py:484-507, 523-530, 548-564, 597-613, 633-660,
    #      692-715.  Per-sample M(t,s;tau) from TwoBoundarySigmoidMLP; same restated contraction as the plain SOCM
    #      loss with dt -> fractional time steps, never forming the reference's (Kp,Kp,B,d,d) tensors.
    def _socm_stopping_objective(self, pb, ts, t_vec, s_vec, ii, jj, states, noises, controls, stop_indicators,
                                 frac, nabla_V, weight):
        sde = self.neural_sde
        K = self.num_steps
        Kp, B, d = states.shape
        tau = ((sde.Phi(states) > 0).to(torch.int).sum(dim=0) - 1) / (Kp - 1)            # method.py:525-530
        tau_vec = tau.unsqueeze(0).expand(t_vec.shape[0], B)
        if states.is_cuda and d > 16 and getattr(self, "fused_stopping", True):
            from .nets import warn_library_fallback
            warn_library_fallback("the stopping-time SOCM target", f"d={d} > 16: the (Np, B, d, d) pair matrices are "
                                  "materialised by torch")
        if states.is_cuda and d <= 16 and getattr(self, "fused_stopping", True):
            # HIP path: the two network evaluations and their s-tangents from one launch of the pair-grid-network kernel
            # (n_in = 3), the gates (models.py:341-392), their s-derivatives and the (Np,B,d,d) matrices formed per
            # (pair, sample) inside socmx_socm_stopping_target_*_f32; nan_to_num(dM/ds) (method.py:553-555) entry-wise there
            M = sde.M
            if M.hip_supported(t_vec.shape[0]):
                N0, N1, dN0, dN1 = M.nets_with_ds_hip(t_vec, s_vec, self.__dict__.setdefault("_stop_grid", {}))
            else:
                N0, N1, dN0, dN1 = M.nets_with_ds(t_vec, s_vec)
            ops = L.socm_operands_hip(pb, ts, self.lmbd, states, noises, controls, frac=frac)
            target = L.stopping_target_hip(M.gamma, M.gamma2, M.gamma3, N0, N1, dN0, dN1, t_vec, s_vec, tau, ops, K, M.T)
            return L.masked_residual_hip(pb, K, target, nabla_V, weight, stop_indicators) / self._stop_normaliser(stop_indicators)
        # dM/ds as a forward-mode tangent (the reference: functorch.jacrev over the batch-summed output)
        M_all, dM_all = torch.func.jvp(lambda s_: sde.M(t_vec, s_, tau_vec), (s_vec,), (torch.ones_like(s_vec),))
        dM_all = torch.nan_to_num(dM_all)                                                   # method.py:553-555
        v, q, gT = L.socm_operands(pb, ts, self.lmbd, states, noises, controls, frac=frac)
        last = jj == K
        jq = torch.clamp(jj, max=K - 1)
        qx = torch.where(last.reshape(-1, 1, 1), gT.unsqueeze(0), q[jq])                    # (Np,B,d)
        vx = torch.where(last.reshape(-1, 1, 1), torch.zeros_like(gT).unsqueeze(0), v[jq])
        contrib = torch.einsum("pmkl,pml->pmk", M_all, qx) - torch.einsum("pmkl,pml->pmk", dM_all, vx)
        target = torch.zeros(Kp, B, d, device=states.device, dtype=contrib.dtype).index_add(0, ii, contrib)
        r = stop_indicators.unsqueeze(2) * ((nabla_V - target) @ pb.sigma)
        return torch.sum(r * r * weight.reshape(1, -1, 1)) / self._stop_normaliser(stop_indicators)

    def _stop_normaliser(self, stop_indicators):
        """sum of the stop indicators (method.py:713-715) -- over the GLOBAL batch when this rank holds a shard (one scalar
        all-reduce: the normaliser scales every gradient, so it is needed before the backward pass)."""
        den = torch.sum(stop_indicators.to(torch.float32))
        if self.shard is not None:
            den = self.shard.allreduce_flat_(den.reshape(1)).reshape(())
        return den

    # ---- method.py:223-906 -------------------------------------------------------------------
    def loss(self, batch_size, compute_L2_error=False, optimal_control=None, compute_control_objective=False,
             algorithm="SOCM_const_M", add_weights=False, total_n_samples=65536, verbose=False,
             u_warm_start=None, use_warm_start=True, use_stopping_time=False):
        if algorithm not in ALGORITHMS:
            raise NotImplementedError(f"algorithm={algorithm!r}: expected one of {ALGORITHMS}")
        if use_stopping_time and algorithm in ("SOCM_const_M", "SOCM_exp", "SOCM_adjoint"):
            raise NotImplementedError(f"{algorithm} has no stopping-time form in the reference either")
        if self.shard is not None and algorithm != "SOCM":
            raise NotImplementedError("batch sharding is implemented for algorithm='SOCM' only")
        if u_warm_start and use_warm_start:
            raise NotImplementedError("warm start is out of scope (SURVEY.md component 9)")
        sde = self.neural_sde
        pb = sde.problem
        shard = self.shard
        B_global = batch_size
        B, row0 = (shard.local_rows(batch_size) if shard is not None else (batch_size, 0))
        K = self.num_steps
        Kp = K + 1
        d = self.dim

        state0 = self.x0.repeat(B, 1)
        ts = self.ts.to(state0)
        noise_in, self.noise_in = self.noise_in, None
        detach = algorithm != "rel_entropy"     # rel_entropy differentiates THROUGH the rollout (method.py:240)
        # The pair-grid network (M) does not depend on the trajectories: with B=128 the rollout occupies 8 of the
        # 256 CUs, so its forward is issued on a second HIP stream and runs beside the rollout kernel.
        fused_M = (algorithm == "SOCM" and not use_stopping_time and state0.is_cuda and type(sde.M) is SigmoidMLP)
        # (short pair grids are launch-bound on the host: the extra stream bookkeeping costs more than it hides)
        side = self._side_stream(state0.device) if fused_M and Kp * (Kp + 1) // 2 >= 4096 else None
        if side is not None:
            t_vec, s_vec, ii, jj, delta = self._pair_grid(ts, K)    # built on this stream the first time: before fork
            fork = torch.cuda.Event()
            fork.record()                       # after the previous optimizer step, before the rollout launch
        # Every loss that does not differentiate through the rollout: the rollout kernel hands over nabla_V at all (K+1) B
        # trajectory rows (it evaluates the network there anyway) and socmx_unet_backward_f32 produces the parameter
        # gradients -- no library forward/backward
        from . import nets as _nets
        fused_V = (detach and R._eligible_for_hip(sde, state0, detach) and getattr(self, "fused_nabla_V", True)
                   and _nets.unet_backward_supported(sde.nabla_V, Kp * B))
        # ... and, where the one-row rollout kernel can (d <= 15, B <= 256, default widths, whole 16-row tiles), it SAVES the network's
        # activations and ReLU signs for that backward (socmx_rollout_ex_f32: act_workspace / act_records): no forward re-computation in it.
        # The autograd-free SOCM body of the Trainer does the same with buffers of its own (socmx/train.py); this is the autograd body --
        # the eight other losses, the eager iteration.  solver.save_activations = False keeps the re-computing backward.
        saved = None
        if fused_V and getattr(self, "save_activations", True) and R.saves_activations(sde, state0, B, K, detach):
            from . import _lib
            ws_n = _lib.C.c_int64(0)
            _lib.check(sde.nabla_V.hip_lib().socmx_unet_backward_sizes(self.dim, _lib.i3(sde.nabla_V.hdims), Kp * B, _lib.C.byref(ws_n), None),
                       "socmx_unet_backward_sizes")
            saved = (torch.empty(ws_n.value, dtype=torch.float32, device=state0.device),
                     torch.empty(Kp * B, 32, dtype=torch.int32, device=state0.device))
        rolled = R.stochastic_trajectories(sde, state0, ts, self.lmbd, detach=detach, noise_in=noise_in, row0=row0,
                                           key=getattr(self, "philox_key", None), want_nabla_v=fused_V,
                                           shares_chip=side is not None,     # (the pair-grid network runs beside it)
                                           act_export=saved)
        (states, noises, stop_indicators, fractional_timesteps, lpd, lps, ltw, controls) = rolled[:8]
        if side is not None:
            with torch.cuda.stream(side):
                side.wait_event(fork)
                net, dnet = sde.M.forward_with_ds(t_vec, s_vec, raw=True)
            main = torch.cuda.current_stream(state0.device)
            net.record_stream(main)
            dnet.record_stream(main)
        if algorithm == "rel_entropy":          # method.py:264-270
            objective = torch.mean(-self.lmbd * (lpd + ltw))
            weight = torch.exp(lpd + lps + ltw).detach()
            norm_sqd_diff = None
            if compute_L2_error:
                tc = optimal_control(self.ts, states, t_is_tensor=True)[:-1].detach()
                norm_sqd_diff = torch.sum((tc - controls.detach()) ** 2 * weight.reshape(1, -1, 1)
                                          / (tc.shape[0] * tc.shape[1]))
            cm = ce = traj = None
            if compute_control_objective:
                cm, ce, traj = self.control_objective(batch_size, total_n_samples=total_n_samples)
            return (objective, norm_sqd_diff, cm, ce, traj, torch.mean(weight), torch.std(weight), stop_indicators)

        weight, stats = L.weights_and_stats(lpd, lps, ltw)
        if shard is not None and getattr(self, "defer_weight_stats", False):
            # Trainer carries the statistics in its ONE flat all-reduce (socmx.dist): leave this shard's (n, mean, M2) slot for
            # it; the mean / std returned below are this shard's own until Trainer replaces them
            from .dist import weight_stat_slots
            self._local_w_sums = weight_stat_slots(weight, shard.rank, shard.world_size)
        elif shard is not None:
            stats = shard.combine_weight_stats(stats)
        w_mean, w_std = L.mean_std_from_stats(stats)

        # nabla_V on all Kp*B trajectory rows (method.py:272-278): library GEMMs + autograd
        if fused_V:
            nabla_V = _nets.unet_on_trajectory(sde.nabla_V, rolled[8], states, ts, saved=saved)
        else:
            tx = torch.cat([ts.reshape(-1, 1, 1).expand(Kp, B, 1), states], dim=-1).reshape(-1, d + 1)
            nabla_V = sde.nabla_V(tx).reshape(Kp, B, d)

        frac = fractional_timesteps if use_stopping_time else None
        if algorithm == "SOCM":
            # M and dM/ds on the pair grid (method.py:510-515, 533-573)
            t_vec, s_vec, ii, jj, delta = self._pair_grid(ts, K)
            if use_stopping_time:
                objective = self._socm_stopping_objective(pb, ts, t_vec, s_vec, ii, jj, states, noises, controls,
                                                          stop_indicators, fractional_timesteps, nabla_V, weight)
            else:
                inv_norm = 1.0 / (Kp * B_global)
                if fused_M:
                    # the exp(-gamma (s-t)) blend and its d/ds are formed inside the HIP contraction
                    if side is None:
                        net, dnet = sde.M.forward_with_ds(t_vec, s_vec, raw=True)
                    else:
                        torch.cuda.current_stream(state0.device).wait_stream(side)
                        if getattr(self, "defer_M_backward", False):
                            # Trainer finishes the M-network's backward + Adam update on the second stream, beside
                            # the NEXT rollout (which needs only nabla_V): cut the graph at (net, dnet)
                            cut = (net.detach().requires_grad_(True), dnet.detach().requires_grad_(True))
                            self._pending_M = (net, dnet) + cut
                            net, dnet = cut
                    objective = L.socm_objective_net(pb, ts, self.lmbd, K, states, noises, controls, net, dnet,
                                                     sde.M.gamma, delta, nabla_V, weight, inv_norm)
                else:
                    M_all, dM_all = sde.M.forward_with_ds(t_vec, s_vec)
                    objective = L.socm_objective(pb, ts, self.lmbd, K, states, noises, controls, M_all, dM_all,
                                                 nabla_V, weight, inv_norm)
        elif algorithm == "SOCM_const_M":
            objective = baselines.socm_const_m(pb, ts, self.lmbd, states, noises, controls, nabla_V, weight)
        elif algorithm == "SOCM_exp":
            objective = baselines.socm_exp(pb, ts, self.T, self.lmbd, self.gamma, states, noises, controls, nabla_V,
                                           weight)
        elif algorithm == "SOCM_adjoint":
            objective = baselines.socm_adjoint(pb, ts, self.dt, states, nabla_V, weight)
        elif algorithm == "cross_entropy":
            objective = baselines.cross_entropy(pb, ts, self.lmbd, states, noises, controls, nabla_V, weight, frac=frac)
        else:  # variance, log-variance, moment
            objective = baselines.variance_family(
                algorithm, pb, ts, self.lmbd, states, noises, controls, nabla_V, weight, self.y0,
                add_weights=add_weights, frac=frac, stop_indicators=stop_indicators if use_stopping_time else None)

        if compute_L2_error:
            target_control = optimal_control(self.ts, states, t_is_tensor=True)
            learned_control = -(nabla_V @ self.sigma)
            # (a shard holds B_local rows: dividing by the GLOBAL batch makes the sum over ranks the reference's value;
            #  Trainer adds the shards up inside its flat all-reduce)
            norm_sqd_diff = torch.sum((target_control - learned_control) ** 2 * weight.reshape(1, -1, 1)
                                      / (target_control.shape[0] * B_global))
        else:
            norm_sqd_diff = None

        if compute_control_objective:
            ctrl_loss_mean, ctrl_loss_std_err, trajectory = self.control_objective(
                batch_size, total_n_samples=total_n_samples)
        else:
            ctrl_loss_mean = ctrl_loss_std_err = trajectory = None

        if verbose and state0.is_cuda:
            # the reference prints NVML numbers for device 0 here (method.py:886-895); ROCm-safe equivalent
            free, total = torch.cuda.mem_get_info(state0.device)
            print("Total memory:", total / 1048576, "MiB")
            print("Free memory:", free / 1048576, "MiB")
            print("Used memory:", (total - free) / 1048576, "MiB")

        return (objective, norm_sqd_diff, ctrl_loss_mean, ctrl_loss_std_err, trajectory, w_mean, w_std,
                stop_indicators)
