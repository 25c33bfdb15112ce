"""One SOCM training iteration with the reference's bookkeeping (main.py:280-359).

`time_per_iteration` -- the quantity behind the "SOCM iters/sec" metric -- is measured exactly like
the reference: wall clock from before `.loss()` to after `optimizer.step(); zero_grad()`
(main.py:280, 351-352).  The normalisation constant is the EMA of the mean importance weight
(main.py:313-322, 354-359; utils.py:389-396).
"""
import time

import numpy as np
import os
import torch


KEYED_OFFSET_BASE = 1 << 31      # first Philox offset of the device-resident key (hipGraph mode)


def compute_EMA(value, EMA_value, EMA_coeff=0.01, itr=0):
    warm = int(np.floor(1 / EMA_coeff))
    if itr == 0:
        return value
    if itr <= warm:
        return (value + itr * EMA_value) / (itr + 1)
    return EMA_coeff * value + (1 - EMA_coeff) * EMA_value


def make_optimizer(solver, nabla_V_lr=1e-4, M_lr=1e-2, adam_eps=1e-4, algorithm="SOCM", y0_lr=1e-2):
    """Adam with the per-algorithm parameter groups of main.py:174-238."""
    sde = solver.neural_sde
    if algorithm == "SOCM":
        groups = [{"params": sde.nabla_V.parameters()},
                  {"params": sde.M.sigmoid_layers.parameters(), "lr": M_lr},
                  {"params": sde.gamma, "lr": M_lr}]
        if getattr(sde, "use_stopping_time", False):
            groups.append({"params": sde.gamma2, "lr": M_lr})   # gamma3 is NOT optimised in the reference (main.py:190-212)
    elif algorithm == "moment":
        groups = [{"params": sde.parameters()}, {"params": solver.y0, "lr": y0_lr}]
    elif algorithm == "SOCM_exp":
        groups = [{"params": sde.parameters()}, {"params": solver.gamma, "lr": M_lr}]
    else:
        groups = [{"params": solver.parameters()}]
    # one fused multi-tensor kernel per group on the GPU (same update rule; ~20 launches fewer per iteration)
    on_gpu = next(sde.parameters()).is_cuda
    return torch.optim.Adam(groups, lr=nabla_V_lr, eps=adam_eps, fused=True if on_gpu else None)


class Trainer:
    def __init__(self, solver, optimizer, batch_size, normalization_const=1.0, algorithm="SOCM",
                 ema_weight_mean_coeff=0.002, sync_timing=True, gemm_select=False, overlap_M_backward=True,
                 grad_telemetry=True, tune_new_shapes=False, hip_graph=False, graph_warmup=2, fused_adam=True, log=None,
                 history_rows=1 << 16, save_activations=True):
        self.solver, self.optimizer = solver, optimizer
        # the rollout saves the control network's activations / ReLU signs for the iteration's backward where its kernel can
        # (socmx_rollout_ex_f32: act_workspace; the autograd-free SOCM body): False keeps the re-computing backward
        self.save_activations = bool(save_activations)
        solver.save_activations = self.save_activations     # (the autograd body's rollout: socmx/solver.py)
        self.batch_size = batch_size
        self.normalization_const = normalization_const
        self.algorithm = algorithm
        self.coeff = ema_weight_mean_coeff
        self.itr = 0
        self.sync_timing = sync_timing
        self.grad_telemetry = grad_telemetry     # main.py:325-345 bookkeeping (part of the reference's timed iteration)
        self._ema_grad = None
        self._ema_grad_norm_sqd = None
        # Library-GEMM selection (PyTorch TunableOp) is process-wide state: opt-in (main.py / bench.py pass the
        # `backend.gemm_select` config key), and by default only the shipped selections are loaded -- timing
        # candidate solutions for new shapes (`tune_new_shapes`) happens inside the first iterations and may pick
        # different solutions on different ranks.
        if gemm_select and solver.x0.is_cuda:
            from . import gemm_select as _gs
            _gs.enable(tune_new_shapes=tune_new_shapes)
        # hipGraph mode (SOCM with or without stopping times; sharded runs: plain SOCM on the hand-written kernels, with the
        # RCCL all-reduces captured inside the graph): the whole iteration -- rollout, loss, backward, collectives, Adam, EMA
        # normaliser, gradient telemetry -- is captured once and replayed; see _graph_step
        self._log = log if log is not None else (lambda *a, **k: None)
        # (SOCM replays the autograd-free body below; the other eight losses of the README sweeps replay the captured autograd
        #  body -- HIP rollout, fused loss kernels, control-network backward, Adam, EMA normaliser -- as one graph as well)
        # Measured at configs[2] (tools/alg_bench.py, ms per iteration eager / replayed): SOCM 1.68 / 0.96, rel_entropy 251 / 120
        # (it differentiates through ~16k eager rollout launches), log-variance 1.23 / 1.13, moment 1.20 / 1.12, variance
        # 1.29 / 1.13, cross_entropy 1.09 / 1.08 -- and SOCM_const_M 1.04 / 1.13, SOCM_adjoint 1.09 / 1.18: those (and SOCM_exp,
        # the same kernels) are GPU-bound already and keep the eager iteration unless asked with hip_graph="force".
        eager_is_faster = algorithm in ("SOCM_const_M", "SOCM_exp", "SOCM_adjoint") and hip_graph != "force"
        self.hip_graph = bool(hip_graph and solver.x0.is_cuda and not eager_is_faster)
        if hip_graph and not self.hip_graph:
            # (never silent: `backend.hip_graph` is on by default, so a run that cannot use it says why)
            self._log("backend.hip_graph: running the eager two-stream iteration instead -- " +
                      (f"measured faster for algorithm {algorithm!r}" if solver.x0.is_cuda else "the solver is not on a GPU"))
        # A sharded run (world size > 1) captures its iteration WITH the collectives inside when the shard's transport is the
        # package's own RCCL communicators (socmx/dist.py "rccl": a collective is one launch on the caller's stream) -- the ranks
        # agree on the capture's success before any of them replays (Shard.agree), so that none falls back alone and waits in a
        # different sequence of collectives.  Over torch's process group (SOCMX_RCCL=0, a communicator that did not come up) or
        # the host-staged test transport nothing is captured: the same autograd-free body runs EAGERLY (one flat all-reduce on
        # the main stream, the pair-grid network's small one beside the next rollout; no gradient copies).
        shard = getattr(solver, "shard", None)
        self._multi_rank = shard is not None and shard.world_size > 1
        capturable = not self._multi_rank or bool(getattr(shard, "capturable", False))
        # (hip_graph="nocapture": that eager form of the body at any world size -- bench.py times it beside the replayed graph)
        self.capture_graphs = self.hip_graph and hip_graph != "nocapture" and capturable
        if self.hip_graph and not self.capture_graphs and hip_graph != "nocapture":
            self._log("backend.hip_graph: sharded run over transport " + repr(getattr(shard, "transport", "?")) + " -- the "
                      "autograd-free iteration runs eagerly (collectives of torch's process group are never captured); the "
                      "package's own RCCL communicators (backend nccl, SOCMX_RCCL unset) capture it")
        self.graph_warmup = int(graph_warmup)
        self.fused_adam = bool(fused_adam)    # hipGraph body: control-network Adam + telemetry as one launch (socmx_adam_step_f32)
        # hipGraph mode: the scalars of iteration n (loss, weight statistics, telemetry, normaliser, L2 error) are ALSO written to row n
        # of a device array by the kernel that forms them; `step` hands out views of that row instead of cloning the replayed graph's
        # static output on the iteration's stream -- a copy and two launch gaps (~10 us) between Adam's end and the next rollout.
        # Rows are written once and never reused (main.py keeps every iteration's values, main.py:398-413): beyond `history_rows`
        # iterations the clone comes back.  main.py passes method.num_iterations.
        self.history_rows = int(history_rows)
        self._graphs = {}
        self._dev = None
        self._m_pending = False          # hipGraph mode: the pair-grid network's update of the last iteration is outstanding
        if self.hip_graph:
            overlap_M_backward = False     # inside a graph the pair-grid network's backward forks and joins within the iteration
        # SOCM on one GPU: the pair-grid network's backward and its Adam groups run on the solver's second stream
        # and overlap with the next iteration's rollout (same arithmetic, same order of updates per parameter)
        # (single-GPU eager runs only: a sharded eager iteration issues ONE collective, after the complete backward)
        self.defer_M = (overlap_M_backward and algorithm == "SOCM" and solver.x0.is_cuda
                        and not getattr(solver.neural_sde, "use_stopping_time", False))
        self._group_split = None

    def _split_groups(self):
        """(groups of the control network & co, groups of the pair-grid network + gamma) of the optimiser's CURRENT
        param_groups -- `optimizer.load_state_dict` replaces the group dicts, so they are never cached by object: the split
        is recomputed whenever the list's dicts change identity."""
        groups = self.optimizer.param_groups
        ident = tuple(id(g) for g in groups)
        if self._group_split is None or self._group_split[0] != ident:
            sde = self.solver.neural_sde
            ids_M = {id(p) for p in sde.M.parameters()} | {id(sde.gamma)}
            side = [g for g in groups if all(id(p) in ids_M for p in g["params"])]
            main = [g for g in groups if not all(id(p) in ids_M for p in g["params"])]
            self._group_split = (ident, main, side)
        return self._group_split[1], self._group_split[2]

    @property
    def _groups_main(self):
        return self._split_groups()[0]

    @property
    def _groups_side(self):
        return self._split_groups()[1]

    @torch.no_grad()
    def _grad_telemetry(self):
        """main.py:325-345, inside the timed region like there: squared norm of the nabla_V gradient, its EMA, and the
        squared norm of the EMA of the gradient -- as multi-tensor ops (about ten launches instead of ~160)."""
        grads = [p.grad for p in self.solver.neural_sde.nabla_V.parameters() if p.grad is not None]
        if not grads:
            return {}
        sq = lambda ts: torch.stack(torch._foreach_norm(ts)).square().sum()
        grad_norm_sqd = sq(grads)
        itr, c = self.itr, 0.01                              # EMA_coeff of main.py:102
        if itr == 0 or self._ema_grad is None:
            self._ema_grad = grads                           # like the reference: the first EMA IS the first gradient
            self._ema_grad_norm_sqd = grad_norm_sqd
        else:
            warm = int(np.floor(1 / c))
            a, b = (itr / (itr + 1), 1.0 / (itr + 1)) if itr <= warm else (1 - c, c)     # compute_EMA, utils.py:389-396
            torch._foreach_mul_(self._ema_grad, a)
            torch._foreach_add_(self._ema_grad, grads, alpha=b)
            self._ema_grad_norm_sqd = compute_EMA(grad_norm_sqd, self._ema_grad_norm_sqd, EMA_coeff=c, itr=itr)
        return dict(grad_norm_sqd=grad_norm_sqd, EMA_grad_norm_sqd=self._ema_grad_norm_sqd,
                    sqd_norm_EMA_grad=sq(self._ema_grad))

    def join(self):
        """Make the current stream wait for the second stream's pending M-network update.  Needed only by code that
        reads M / gamma outside `step()` / `solver.loss()` while `sync_timing=False` (with the default
        `sync_timing=True` every step ends with a device synchronisation, like the reference's timing does)."""
        dev = self.solver.x0.device
        if self.hip_graph:
            self._flush_M()
        if self.defer_M and dev.type == "cuda":
            side = self.solver._side_stream(dev)
            if side is not None:
                torch.cuda.current_stream(dev).wait_stream(side)

    def _step_groups(self, groups):
        opt = self.optimizer
        saved = opt.param_groups
        opt.param_groups = groups
        try:
            opt.step()
        finally:
            opt.param_groups = saved

    def _finish_M_on_side_stream(self, pending, dev):
        """Backward of the pair-grid network from the gradients at the cut, then its Adam groups, on the second
        stream.  The next forward of that network is issued on the same stream, hence ordered after the update."""
        solver = self.solver
        side = solver._side_stream(dev)
        main = torch.cuda.current_stream(dev)
        net, dnet, net_cut, dnet_cut = pending
        done = torch.cuda.Event()
        done.record(main)
        telemetry = {}
        with torch.cuda.stream(side), torch.no_grad():
            side.wait_event(done)
            if self.grad_telemetry:       # reads the nabla_V gradients only: off the main stream's critical path
                for p_ in solver.neural_sde.nabla_V.parameters():
                    if p_.grad is not None:
                        p_.grad.record_stream(side)
                telemetry = self._grad_telemetry()
            grads = [net_cut.grad, dnet_cut.grad]
            for g in grads:
                g.record_stream(side)
            with torch.enable_grad():
                torch.autograd.backward([net, dnet], grads)
            for grp in self._groups_side:
                for p in grp["params"]:
                    if p.grad is not None:
                        p.grad.record_stream(side)       # gamma's gradient was produced on the main stream
            self._step_groups(self._groups_side)
        return telemetry

    # ---- hipGraph replay of the iteration (SOCM, with or without stopping times) --------------------------------------
    # Everything the iteration needs between launches lives on the device: the Philox key (socmx.rollout.PhiloxKey,
    # advanced by a one-thread node behind the rollout), the iteration counter and the EMA normaliser (the
    # reference's host-side compute_EMA, utils.py:389-396, written with torch.where on a device counter), Adam's step
    # counters (capturable=True).  A replay is then ONE host call; the reference's per-iteration semantics
    # (main.py:280-359) are unchanged -- test_gpu_graph.py replays reference-generated training fixtures through it.
    def _graph_state(self):
        if self._dev is None:
            import gc
            gc.collect()           # drop unreachable autograd graphs of earlier eager iterations (see _eager_step's note)
            from .rollout import PhiloxKey
            dev = self.solver.x0.device
            f = lambda v: torch.tensor(float(v), dtype=torch.float32, device=dev)
            nc = self.normalization_const
            self._dev = dict(norm=(nc.detach().clone().to(dev, torch.float32).reshape(()) if torch.is_tensor(nc) else f(nc)),
                             itr=f(self.itr), ema_gn=f(0.0),
                             ema_grad=[torch.zeros_like(p) for p in self.solver.neural_sde.nabla_V.parameters()])
            D = self._dev          # (1,) views of the 0-dim state tensors for the C ABI, and the (A, B) pair of the EMA
            D["itr1"], D["norm1"], D["ema_gn1"] = D["itr"].reshape(1), D["norm"].reshape(1), D["ema_gn"].reshape(1)
            D["ab"] = torch.zeros(2, dtype=torch.float32, device=dev)
            D["hist"] = (torch.zeros(self.history_rows, 8, dtype=torch.float32, device=dev) if self.history_rows > 0 else None)
            if getattr(self.solver, "philox_key", None) is None:
                # a stream of its own: the host-side counter (rollout._philox_calls: evaluation bursts, eager rollouts)
                # counts 0, 1, 2, ... under the same seed, the device key counts from 2^31 -- the 32-bit offset word of the
                # Philox counter (include/socmx.h) never coincides, so no burst re-draws noise an iteration trained on
                self.solver.philox_key = PhiloxKey(dev, offset=KEYED_OFFSET_BASE)
            self._make_capturable()
        return self._dev

    @staticmethod
    def _ema_dev(value, ema, coeff, itr):
        """compute_EMA with the iteration counter as a device tensor (same arithmetic, branch by torch.where)."""
        warm = float(int(np.floor(1 / coeff)))
        running = (value + itr * ema) / (itr + 1)
        smooth = coeff * value + (1 - coeff) * ema
        return torch.where(itr == 0, value, torch.where(itr <= warm, running, smooth))

    def _body_dev(self, loss_kwargs):
        """One iteration expressed on device-resident state only (capturable); returns a (7,) tensor
        [loss, weight_mean, weight_std, grad_norm_sqd, EMA_grad_norm_sqd, sqd_norm_EMA_grad, normaliser before]."""
        solver, D = self.solver, self._graph_state()
        out = solver.loss(self.batch_size, algorithm=self.algorithm, use_warm_start=False,
                          use_stopping_time=bool(getattr(solver.neural_sde, "use_stopping_time", False)), **loss_kwargs)
        if self.algorithm in ("SOCM", "SOCM_const_M", "SOCM_exp", "SOCM_adjoint", "cross_entropy"):
            loss = out[0] / D["norm"]                                    # main.py:313-320
        elif self.algorithm == "variance":
            loss = out[0] / D["norm"] ** 2                               # main.py:321-322
        else:
            loss = out[0]
        # main.py:323 -- as torch.autograd.grad: `.backward()` routes every parameter's gradient through its AccumulateGrad
        # node, which is pinned to the stream that was current when the node was first created.  If any earlier autograd
        # graph is still alive (an eager iteration's outputs kept by the caller) that is the DEFAULT stream, and the
        # engine's synchronisation with it inside a capture invalidates the capture (observed: segfault in capture_end).
        params = [p for g in self.optimizer.param_groups for p in g["params"]]
        for p, g in zip(params, torch.autograd.grad(loss, params, allow_unused=True)):
            p.grad = g
        # Scalar bookkeeping on the device by socmx_iteration_scalars_f32 (two one-thread launches) instead of ~25 elementwise
        # torch launches per iteration (the (A, B) coefficients, two compute_EMA evaluations with torch.where, the counter, the
        # stack of the outputs): the README's molecular_dynamics iteration is launch-bound -- 119 launches, 1.21 ms.
        from . import _lib
        Lh, f = _lib.lib(), _lib.ptr
        dev = D["norm"].device
        gn = gne = None
        if self.grad_telemetry:                                          # main.py:325-345
            with torch.no_grad():
                grads = [p.grad for p in solver.neural_sde.nabla_V.parameters()]
                sq = lambda ts: torch.stack(torch._foreach_norm(ts)).square().sum().reshape(1)
                gn = sq(grads)
                with _lib.on_device(dev):
                    _lib.check(Lh.socmx_iteration_scalars_f32(0, f(D["itr1"]), None, None, None, None, None, None, None,
                                                              self.coeff, 0.01, f(D["ab"]), None, _lib.stream_ptr(dev)),
                               "socmx_iteration_scalars_f32")
                torch._foreach_mul_(D["ema_grad"], D["ab"][0])
                torch._foreach_add_(D["ema_grad"], torch._foreach_mul(grads, D["ab"][1]))
                gne = sq(D["ema_grad"])
        with torch.no_grad():
            self.optimizer.step()                                        # main.py:347-349
            self.optimizer.zero_grad(set_to_none=True)
            f32 = lambda t: t.detach().to(torch.float32).reshape(1).contiguous()
            w_mean, w_std, obj = f32(out[5]), f32(out[6]), f32(out[0])
            vals = torch.empty(7, dtype=torch.float32, device=dev)
            with _lib.on_device(dev):                                    # main.py:313-320, 354-359: loss / normaliser, EMAs, counter
                _lib.check(Lh.socmx_iteration_scalars_f32(
                    1, f(D["itr1"]), f(D["norm1"]), f(D["ema_gn1"]) if gn is not None else None, f(w_mean), f(w_std), f(obj),
                    f(gn) if gn is not None else None, f(gne) if gne is not None else None, self.coeff, 0.01, None, f(vals),
                    _lib.stream_ptr(dev)), "socmx_iteration_scalars_f32")
            # (this body -- the eight other losses, stopping-time SOCM -- keeps the cloned output: its loss value may be rescaled below)
            D["hist_written"] = False
            if self.algorithm not in ("SOCM", "SOCM_const_M", "SOCM_exp", "SOCM_adjoint", "cross_entropy"):
                vals[0:1].copy_(loss.detach().reshape(1))                # (another scaling of the objective: main.py:321-322)
            if out[1] is not None:
                vals = torch.cat([vals, out[1].detach().reshape(1).to(torch.float32)])
            return vals

    # ---- the iteration without autograd (plain SOCM, every network on the hand-written kernels) -----------------------
    # main stream: rollout (+ nabla_V values) -> weights -> operands -> contraction forward (objective, G) -> control-network
    # backward from G -> Adam(nabla_V).  Everything that only serves the pair-grid network -- the contraction BACKWARD
    # (d obj / d (net, dnet, gamma)), that network's backward and its Adam groups for iteration n -- runs at the START of
    # iteration n+1 on the second stream, beside that iteration's rollout, followed by the network's forward for iteration
    # n+1: the critical path of an iteration is rollout + contraction forward + control-network backward.  This is
    # expressible inside ONE captured graph because nothing of it lives in an autograd graph: the buffers that cross the
    # iteration boundary (G, q, v, gT, net, dnet, gamma and 1/normaliser of iteration n; g_net, g_dnet, g_gamma; the packed
    # weight image the backward recomputes from) are owned by the Trainer, at fixed addresses -- the second stream consumes
    # iteration n's values before the main stream, behind the join, overwrites them with iteration n+1's.
    def _manual_ok(self, loss_kwargs):
        from . import nets
        solver, sde = self.solver, self.solver.neural_sde
        # (keyword arguments of solver.loss the manual body implements itself: the weighted L2 error against a ground truth)
        extra = {k: v for k, v in loss_kwargs.items()
                 if k not in ("compute_L2_error", "optimal_control", "total_n_samples") and v}
        if loss_kwargs.get("compute_L2_error") and loss_kwargs.get("optimal_control") is None:
            return False
        if self.algorithm != "SOCM" or extra or getattr(sde, "use_stopping_time", False) or type(sde.M) is not nets.SigmoidMLP:
            return False
        if not getattr(solver, "fused_nabla_V", True) or not getattr(sde.M, "fused_pair_net", True):
            return False
        K = solver.num_steps
        return (nets.unet_backward_supported(sde.nabla_V, (K + 1) * self.batch_size)
                and nets.pair_net_supported(sde.M, (K + 1) * (K + 2) // 2))

    def _m_update(self, t_vec, s_vec):
        """Backward of the pair-grid network from the kept gradients + Adam for (M, gamma): iteration n's update."""
        from . import nets
        sde, D = self.solver.neural_sde, self._dev
        layers = [sde.M.sigmoid_layers[i] for i in (0, 2, 4)]
        params = [p for l in layers for p in (l.weight, l.bias)]
        shard = self.solver.shard
        # the contraction backward of iteration n (d obj / d (net, dnet, gamma): it feeds this network only, the control
        # network's gradient came from the forward kernels' G) runs HERE, on the second stream beside rollout n + 1, from the
        # operands iteration n left in the Trainer's buffers
        from . import loss as L
        solver = self.solver
        _, _, part = L.target_bwd_net(solver.dim, solver.num_steps, D["q"].shape[1], D["G"], D, D["gout"], D["net"], D["dnet"],
                                      D["delta"], D["gam"], g_net=D["g_net"], g_dnet=D["g_dnet"])
        side = self._fused_side_table(D, params + [sde.gamma])
        if side is not None:
            torch.sum(part, dim=0, keepdim=True, out=D["m_flat"][side[4]:])      # gamma's gradient: straight into the flat buffer's tail
        else:
            torch.sum(part, dim=0, keepdim=True, out=D["g_gamma"])
        if side is not None:
            # The pair-grid network's gradients and gamma's in ONE flat buffer (the buffer a sharded run all-reduces), then ONE
            # Adam launch over the seven tensors (socmx_adam_step_f32) instead of torch's four multi-tensor launches: this
            # branch runs on the second stream beside the rollout and, since the rollout got shorter than it, IS the iteration's
            # critical path up to the point where the main stream joins it
            from . import _lib
            table, grp, sums, scratch, n = side
            flat = D["m_flat"]
            nets.pair_net_backward(sde.M.dim, sde.M.hdims, [p.shape for p in params], D["packed"], t_vec, s_vec,
                                   D["g_net"], D["g_dnet"], out=flat)
            if shard is not None:
                shard.allreduce_flat_(flat, slot="side")
            b1, b2 = grp["betas"]
            dev = flat.device
            with _lib.on_device(dev):
                _lib.check(_lib.lib().socmx_adam_step_f32(table.data_ptr(), len(params) + 1, flat.numel(), _lib.ptr(flat), None,
                                                         _lib.ptr(D["itr1"]), 0.01, float(grp["lr"]), float(b1), float(b2),
                                                         float(grp["eps"]), _lib.ptr(scratch), _lib.ptr(sums),
                                                         _lib.stream_ptr(dev)), "socmx_adam_step_f32")
            return
        if shard is None:
            grads = nets.pair_net_backward(sde.M.dim, sde.M.hdims, [p.shape for p in params], D["packed"], t_vec, s_vec,
                                           D["g_net"], D["g_dnet"])
            g_gamma = D["g_gamma"].reshape(sde.gamma.shape).clone()
        else:
            # sharded: g_net / g_dnet / g_gamma are this rank's partial sums, and the network's backward is linear in them --
            # its parameter gradients + gamma's travel in ONE small all-reduce (the iteration's second collective, on the
            # second stream beside the rollout, inside the captured graph)
            n = sum(p.numel() for p in params)
            if "m_flat" not in D:
                D["m_flat"] = torch.empty(n + 1, dtype=torch.float32, device=D["g_gamma"].device)
            grads = nets.pair_net_backward(sde.M.dim, sde.M.hdims, [p.shape for p in params], D["packed"], t_vec, s_vec,
                                           D["g_net"], D["g_dnet"], out=D["m_flat"])
            D["m_flat"][n:].copy_(D["g_gamma"])
            shard.allreduce_flat_(D["m_flat"], slot="side")
            g_gamma = D["m_flat"][n:].reshape(sde.gamma.shape).clone()
        for p, g in zip(params, grads):
            p.grad = g
        sde.gamma.grad = g_gamma
        self._step_groups(self._groups_side)
        for p in params + [sde.gamma]:
            p.grad = None
        # (torch's step has created the Adam state by now: build the fused step's table here, outside any capture, so that the
        #  next update -- possibly the captured one -- finds it)
        if self.fused_adam and D is not None and not torch.cuda.is_current_stream_capturing():
            self._fused_side_table(D, params + [sde.gamma])

    def _fused_side_table(self, D, tensors):
        """Device table for socmx_adam_step_f32 over the pair-grid network's six tensors + gamma, or None while the fused step does
        not apply: every side group must be a torch.optim.Adam group with the SAME float lr / betas / eps, no weight decay /
        amsgrad / maximize, covering exactly these tensors (contiguous fp32, gamma a single element), with device-resident fp32
        state that torch's own step created (the eager warm-up iterations) and EQUAL step counts -- checked when the table is
        built.  Rebuilt whenever a data pointer or a hyper-parameter changed."""
        if not self.fused_adam or "itr1" not in D:
            return None
        opt, groups = self.optimizer, self._groups_side
        if type(opt) is not torch.optim.Adam or not groups:
            return None
        g0 = groups[0]
        hyper = lambda g: (g["lr"], tuple(g["betas"]), g["eps"])
        if any(torch.is_tensor(g["lr"]) or g.get("weight_decay", 0) != 0 or g.get("amsgrad", False) or g.get("maximize", False)
               or hyper(g) != hyper(g0) for g in groups):
            return None
        if sorted(id(p) for g in groups for p in g["params"]) != sorted(id(p) for p in tensors):
            return None
        if tensors[-1].numel() != 1:          # (gamma: one scalar behind the network's gradients in the flat buffer)
            return None
        st = opt.state
        for p in tensors:
            q = st.get(p)
            if (not q or not torch.is_tensor(q.get("step")) or q["step"].device != p.device or q["step"].dtype != torch.float32
                    or p.dtype != torch.float32 or not p.is_contiguous()):
                return None
        sig = (tuple(x.data_ptr() for p in tensors for x in (p, st[p]["exp_avg"], st[p]["exp_avg_sq"], st[p]["step"])),
               float(g0["lr"]), tuple(g0["betas"]), float(g0["eps"]))
        if D.get("adam_side_sig") != sig:
            if torch.cuda.is_current_stream_capturing():
                return None              # (the table is a host-to-device copy: built by an eager iteration, see _m_update's tail)
            # adam_step_kernel takes the bias correction from the FIRST tensor's step counter and writes it back to all of them:
            # only valid while the seven counters agree (they do when torch's own step created the state in one go; an
            # optimizer.load_state_dict or a history that stepped the groups apart must keep torch's per-tensor step) -- read
            # here, outside any capture, where a host synchronisation is affordable
            steps = torch.stack([st[p]["step"].reshape(()) for p in tensors]).cpu()
            if not bool((steps == steps[0]).all()):
                D.pop("adam_side", None)
                D["adam_side_sig"] = None
                return None
            rows, off = [], 0
            for p in tensors:
                rows.append([p.data_ptr(), st[p]["exp_avg"].data_ptr(), st[p]["exp_avg_sq"].data_ptr(), st[p]["step"].data_ptr(),
                             p.numel(), off])
                off += p.numel()
            dev = tensors[0].device
            n = off - tensors[-1].numel()
            if "m_flat" not in D or D["m_flat"].numel() != off:
                D["m_flat"] = torch.empty(off, dtype=torch.float32, device=dev)
            D["adam_side"] = (torch.tensor(rows, dtype=torch.int64, device=dev), g0, torch.zeros(2, dtype=torch.float32, device=dev),
                              torch.zeros(4 + 2 * ((off + 1023) // 1024), dtype=torch.float32, device=dev), n)
            D["adam_side_sig"] = sig
        return D["adam_side"]

    def _flush_M(self):
        """Apply the outstanding pair-grid-network update now (before anything outside the replayed graph reads or
        trains M / gamma: checkpoints, eager fallback iterations, the end of training)."""
        if not self._m_pending:
            return
        solver = self.solver
        dev = solver.x0.device
        side = solver._side_stream(dev)
        if side is not None:
            torch.cuda.current_stream(dev).wait_stream(side)
        t_vec, s_vec = solver._pair_grid(solver.ts.to(dev), solver.num_steps)[:2]
        with torch.no_grad():
            self._m_update(t_vec, s_vec)
        self._m_pending = False

    @torch.no_grad()
    def _body_manual(self, loss_kwargs=None):
        from . import loss as L, nets, rollout as R
        solver, D = self.solver, self._graph_state()
        sde, pb = solver.neural_sde, solver.neural_sde.problem
        dev = solver.x0.device
        shard = solver.shard
        B_global, K, d = self.batch_size, solver.num_steps, solver.dim
        B, row0 = (B_global, 0) if shard is None else shard.local_rows(B_global)
        Kp = K + 1
        ts = solver.ts.to(dev)
        t_vec, s_vec, ii, jj, delta = solver._pair_grid(ts, K)
        Np = t_vec.shape[0]
        M = sde.M
        if "packed" not in D:
            from . import _lib
            D["packed"] = torch.empty(_lib.lib().socmx_mnet_packed_floats(d, _lib.i2(M.hdims)), dtype=torch.float32,
                                      device=dev)
            f32 = dict(dtype=torch.float32, device=dev)
            D["g_net"] = torch.zeros(Np, d, d, **f32)
            D["g_dnet"] = torch.zeros(Np, d, d, **f32)
            D["g_gamma"] = torch.zeros(1, **f32)
            # what the DEFERRED contraction backward reads one iteration later (fixed addresses: a replayed graph's second
            # stream consumes the previous replay's values before this replay's main stream overwrites them behind the join)
            D["net"], D["dnet"] = torch.empty(Np, d, d, **f32), torch.empty(Np, d, d, **f32)
            D["q"], D["v"] = torch.empty(K, B, d, **f32), torch.empty(K, B, d, **f32)
            D["gT"] = torch.empty(B, d, **f32)
            D["G"] = torch.empty(Kp, B, d, **f32)
            D["gout"], D["gam"] = torch.ones(1, **f32), torch.ones(1, **f32)
            D["obj"] = torch.zeros(1, **f32)
            D["delta"] = delta
            # the telemetry EMA of the control-network gradient as one flat buffer, in parameters() order (= the order of
            # socmx_unet_backward_f32's output); D["ema_grad"] (shared with the autograd body / the eager mirrors) = its views
            vp = list(sde.nabla_V.parameters())
            D["ema_flat"] = torch.cat([e.reshape(-1) for e in D["ema_grad"]])
            off = 0
            for i, p_ in enumerate(vp):
                D["ema_grad"][i] = D["ema_flat"][off:off + p_.numel()].view_as(p_)
                off += p_.numel()
        layers = [M.sigmoid_layers[i] for i in (0, 2, 4)]
        mparams = [p.detach() for l in layers for p in (l.weight, l.bias)]

        def m_branch(gout_free=None):
            if self._m_pending:
                self._m_update(t_vec, s_vec)
            if gout_free is not None:
                gout_free.record(torch.cuda.current_stream(dev))      # the deferred backward has read the previous d loss / d objective
            net, dnet, _ = nets.pair_net_forward(d, M.hdims, mparams, t_vec, s_vec, packed=D["packed"],
                                                 out=(D["net"], D["dnet"]))
            # two scalars the main stream needs only behind the rollout: gamma (final once the deferred update above ran) and
            # 1 / running normaliser = d loss / d objective (main.py:313-320) -- formed here, off the critical path
            if side is None:
                D["gam"].copy_(sde.gamma.detach().reshape(1))
                torch.reciprocal(D["norm1"], out=D["gout"])
                D["obj"].zero_()       # (the objective's accumulator: cleared here instead of in front of the contraction)
            return net, dnet

        main = torch.cuda.current_stream(dev)
        # (inside a captured graph a second stream costs nothing on the host: the pair-grid network's branch runs beside the rollout at
        #  every grid size -- the eager iteration keeps its 4,096-pair threshold, solver.py.  configs[1]: 0.40 -> 0.33 ms, soc.yaml's
        #  default d = 20: 0.73 -> 0.62 ms)
        side = solver._side_stream(dev)
        # (x0 repeated over the batch: built once, in an eager warm-up iteration -- inside the captured body it was one more launch
        #  in front of every rollout)
        tag = (B, solver.x0.data_ptr(), solver.x0._version)
        if D.get("state0_tag") != tag:
            D["state0"], D["state0_tag"] = solver.x0.repeat(B, 1), tag
        state0 = D["state0"]
        noise_in, solver.noise_in = solver.noise_in, None
        if side is not None:
            fork = torch.cuda.Event()
            fork.record(main)
        else:
            net, dnet = m_branch()
        # The rollout is enqueued BEFORE the second stream's branch: its few workgroups claim whole CUs (exclusive LDS), and
        # behind a chip-filling kernel that keeps refilling every CU with small workgroups they wait for CUs to drain (the
        # d = 64 rollout took 13.0 instead of 7.3 ms behind the deferred contraction backward)
        # The rollout SAVES the control network's activations and ReLU signs for this iteration's backward where it can (the one-row
        # kernel: BASELINE configs[1], [2]): the backward's workspace and the rows' records are the Trainer's, at fixed addresses, written
        # by the rollout and read behind the contraction -- kernel A then runs its five backward stages only (144 -> 86 us at configs[2]).
        # backend.save_activations: False (Trainer(save_activations=False)) keeps the re-computing backward.
        saved = None
        if self.save_activations:
            tag_s = (B, K, d, tuple(sde.nabla_V.hdims))
            if D.get("saved_tag") != tag_s:
                D["saved_tag"], D["saved"] = tag_s, None
                if R.saves_activations(sde, state0, B, K):
                    from . import _lib
                    ws_n = _lib.C.c_int64(0)
                    _lib.check(sde.nabla_V.hip_lib().socmx_unet_backward_sizes(d, _lib.i3(sde.nabla_V.hdims), Kp * B, _lib.C.byref(ws_n),
                                                                               None), "socmx_unet_backward_sizes")
                    D["saved"] = (torch.empty(ws_n.value, dtype=torch.float32, device=dev),
                                  torch.zeros(Kp * B, 32, dtype=torch.int32, device=dev))
            saved = D["saved"]
        (states, noises, stop, frac, lpd, lps, ltw, controls, nabla_v) = R.stochastic_trajectories(
            sde, state0, ts, solver.lmbd, noise_in=noise_in, key=solver.philox_key, want_nabla_v=True, row0=row0,
            shares_chip=True,        # (the second stream's branch -- deferred contraction backward, pair-grid network -- runs beside it)
            act_export=saved)
        packed_bwd = None
        scal3 = None
        join_early = False
        if side is not None:
            gout_free = torch.cuda.Event()
            with torch.cuda.stream(side):
                side.wait_event(fork)
                # (the transposed weight image the control-network backward reads: packed here, beside the rollout, instead of
                #  in front of the backward on the critical path -- the weights do not change in between)
                packed_bwd = sde.nabla_V.packed_bwd()
                net, dnet = m_branch(gout_free)
            # two scalars of the main stream's own chain, enqueued behind the rollout where this stream waits for the second one
            # anyway (on the second stream they were 20 us at the end of what has become the longer branch): 1 / running
            # normaliser = d loss / d objective (main.py:313-320) -- once the deferred backward has read the previous one -- and
            # the objective's cleared accumulator
            # Where the main stream joins the second one.  Small pair matrices (d <= 16: the branch is SHORTER than the rollout since the
            # pair-grid network's kernels went resident): right here, in front of the weights -- the cross-queue hand-shake (~6 us as a
            # separate wait in front of the contraction) then resolves while the rollout is still running.  Large ones (the d = 64 slice:
            # the branch is three times the rollout): only the event, and the join behind the operands, which are formed meanwhile.
            join_early = d <= 16
            if join_early:
                main.wait_stream(side)
            else:
                main.wait_event(gout_free)
            # (gamma is final once the deferred update ran; the three scalars ride in the weights' launch below)
            scal3 = (sde.gamma.detach().reshape(1), D["gam"], D["norm1"], D["gout"], D["obj"])
        # (the weights and the contraction's operands need the rollout only -- and, for the operand buffers the deferred backward
        #  of the previous iteration read, the event above: formed BEFORE the join, while the second stream finishes the
        #  pair-grid network's forward)
        weight, stats = L.weights_and_stats(lpd, lps, ltw, scalars=scal3)
        w_mean, w_std = L.mean_std_from_stats(stats)
        ops = L.socm_operands_hip(pb, ts, solver.lmbd, states, noises, controls, out=D)
        if side is not None:
            if not join_early:
                main.wait_stream(side)
            net.record_stream(main)
            dnet.record_stream(main)
        gam = D["gam"]                                                     # (filled in m_branch)
        obj, G, _ = L.target_fwd_net(pb, K, net, dnet, delta, gam, ops, nabla_v, weight, 1.0 / (Kp * B_global), G=D["G"],
                                     obj=D["obj"])
        gout = D["gout"]                                                   # d loss / d objective, from m_branch
        # (the contraction BACKWARD -- gradients of the pair-grid network and gamma only -- is deferred: _m_update)
        from . import _lib
        Lh, f = _lib.lib(), _lib.ptr
        want_l2 = bool(loss_kwargs and loss_kwargs.get("compute_L2_error"))
        main_flat = None
        if shard is not None:
            # the iteration's flat all-reduce buffer: [control-network gradient | objective | one (n, mean, M2) slot per rank | L2 error]
            ws, ng = _lib.C.c_int64(0), _lib.C.c_int64(0)
            _lib.check(Lh.socmx_unet_backward_sizes(d, _lib.i3(sde.nabla_V.hdims), Kp * B, _lib.C.byref(ws), _lib.C.byref(ng)),
                       "socmx_unet_backward_sizes")
            n_slots = 3 * shard.world_size
            main_flat = torch.zeros(ng.value + 2 + n_slots, dtype=torch.float32, device=dev)
        # (G . d loss / d objective: the device scalar multiplies the gradient tiles as the backward kernel reads them)
        vgrads, vflat = nets.unet_backward_hip(sde.nabla_V, states.reshape(Kp * B, d), ts, B,
                                               G.reshape(Kp * B, d), gout_scale=gout, return_flat=True,
                                               packed=sde.nabla_V._packed,     # (the image this iteration's rollout packed)
                                               out=main_flat, packed_bwd=packed_bwd, saved=saved)
        vparams = list(sde.nabla_V.parameters())
        nsd = None
        if want_l2:
            # method.py:858-873: weighted squared distance between the learned control and the ground truth on this batch
            # (a shard's share: divided by the GLOBAL batch, summed by the all-reduce)
            target_control = loss_kwargs["optimal_control"](solver.ts, states, t_is_tensor=True)
            learned = -(nabla_v @ solver.sigma)
            nsd = torch.sum((target_control - learned) ** 2 * weight.reshape(1, -1, 1)
                            / (target_control.shape[0] * B_global)).reshape(1)
        w_mean, w_std = stats[3:4].contiguous(), stats[4:5].contiguous()
        if shard is not None:
            tail = main_flat[vflat.numel():]
            with _lib.on_device(dev):
                _lib.check(Lh.socmx_shard_stats_f32(0, f(weight), B, shard.rank, shard.world_size, f(obj), f(tail), None,
                                                    _lib.stream_ptr(dev)), "socmx_shard_stats_f32")
            if nsd is not None:
                tail[1 + n_slots:2 + n_slots].copy_(nsd)
            shard.allreduce_flat_(main_flat)           # the iteration's ONE collective on this stream (captured in the graph)
            mean_std = torch.empty(2, dtype=torch.float32, device=dev)
            with _lib.on_device(dev):
                _lib.check(Lh.socmx_shard_stats_f32(1, None, 0, shard.rank, shard.world_size, None, f(tail), f(mean_std),
                                                    _lib.stream_ptr(dev)), "socmx_shard_stats_f32")
            obj, w_mean, w_std = tail[0:1], mean_std[0:1], mean_std[1:2]
            if nsd is not None:
                nsd = tail[1 + n_slots:2 + n_slots]
        gn = gne = None
        adam = self._fused_adam_table(D, vparams, vflat)
        out = torch.empty(7, dtype=torch.float32, device=dev)
        if adam is not None:
            # control-network Adam step + gradient telemetry + the iteration's scalar bookkeeping (loss, EMA normaliser, iteration
            # counter, telemetry EMAs) in ONE launch on the flat gradient (socmx_adam_step_scalars_f32): replaces the optimiser's
            # multi-tensor launches, two dot products, the EMA lerp, the coefficient kernel and ~35 elementwise launches
            table, _, sums = adam
            grp = self._groups_main[0]
            b1, b2 = grp["betas"]
            with _lib.on_device(dev):
                hist = D.get("hist")
                _lib.check(Lh.socmx_adam_step_scalars_hist_f32(
                    table.data_ptr(), len(vparams), vflat.numel(), f(vflat),
                    f(D["ema_flat"]) if self.grad_telemetry else None, f(D["itr1"]), 0.01, float(grp["lr"]), float(b1),
                    float(b2), float(grp["eps"]), f(D["adam_scratch"]), f(sums), f(D["norm1"]),
                    f(D["ema_gn1"]) if self.grad_telemetry else None, f(w_mean), f(w_std), f(obj), self.coeff, f(out),
                    f(hist), 0 if hist is None else hist.shape[0], f(nsd.contiguous()) if nsd is not None else None,
                    _lib.stream_ptr(dev)), "socmx_adam_step_scalars_hist_f32")
                D["hist_written"] = hist is not None
        else:
            for p, g in zip(vparams, vgrads):
                p.grad = g
            if self.grad_telemetry:                                      # main.py:325-345
                # the control-network gradients are views of ONE flat buffer (socmx_unet_backward_f32), and so is their EMA:
                # squared norms are dot products, the EMA  A ema + B g  (A = 1 - B in every branch of compute_EMA) one lerp
                gn = torch.dot(vflat, vflat).reshape(1)
                with _lib.on_device(dev):
                    _lib.check(Lh.socmx_iteration_scalars_f32(0, f(D["itr1"]), None, None, None, None, None, None, None,
                                                              self.coeff, 0.01, f(D["ab"]), None, _lib.stream_ptr(dev)),
                               "socmx_iteration_scalars_f32")
                D["ema_flat"].lerp_(vflat, D["ab"][1])
                gne = torch.dot(D["ema_flat"], D["ema_flat"]).reshape(1)
            self._step_groups(self._groups_main)                          # main.py:347-349 (nabla_V: the next rollout needs it)
            for p in vparams:
                p.grad = None
            # scalar bookkeeping: one one-thread kernel instead of ~35 elementwise launches (socmx_iteration_scalars_f32)
            with _lib.on_device(dev):
                hist = D.get("hist")
                _lib.check(Lh.socmx_iteration_scalars_hist_f32(
                    1, f(D["itr1"]), f(D["norm1"]), f(D["ema_gn1"]) if gn is not None else None, f(w_mean),
                    f(w_std), f(obj), f(gn) if gn is not None else None, f(gne) if gne is not None else None,
                    self.coeff, 0.01, None, f(out), f(hist), 0 if hist is None else hist.shape[0],
                    f(nsd.contiguous()) if nsd is not None else None, _lib.stream_ptr(dev)), "socmx_iteration_scalars_hist_f32")
                D["hist_written"] = hist is not None
        self._m_pending = True
        if nsd is not None:
            out = torch.cat([out, nsd.reshape(1)])
        return out

    def _fused_adam_table(self, D, vparams, vflat):
        """Device table for socmx_adam_step_f32, or None while the fused step does not apply: the control network must be
        exactly one torch.optim.Adam group (no weight decay / amsgrad / maximize, float lr) whose state already exists (the
        first eager iterations create it through torch's own step) with device-resident fp32 step counters."""
        if not self.fused_adam:
            return None
        if "adam_table" in D:
            return D["adam_table"] if self._adam_signature(self._groups_main[0], vparams) == D["adam_sig"] else None
        opt, groups = self.optimizer, self._groups_main
        if type(opt) is not torch.optim.Adam or len(groups) != 1:
            return None
        grp = groups[0]
        if ([id(p) for p in grp["params"]] != [id(p) for p in vparams] or grp.get("weight_decay", 0) != 0
                or grp.get("amsgrad", False) or grp.get("maximize", False) or torch.is_tensor(grp["lr"])
                or len(vparams) > 64):
            return None
        rows, off = [], 0
        for p in vparams:
            st = opt.state.get(p)
            if (not st or not torch.is_tensor(st.get("step")) or st["step"].device != p.device
                    or st["step"].dtype != torch.float32 or p.dtype != torch.float32 or not p.is_contiguous()):
                return None
            rows.append([p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), st["step"].data_ptr(),
                         p.numel(), off])
            off += p.numel()
        if off != vflat.numel():
            return None
        dev = vflat.device
        D["adam_scratch"] = torch.zeros(4 + 2 * ((off + 1023) // 1024), dtype=torch.float32, device=dev)
        sums = torch.zeros(2, dtype=torch.float32, device=dev)
        D["adam_table"] = (torch.tensor(rows, dtype=torch.int64, device=dev), grp, sums)     # socmx_adam_tensor records
        D["adam_sig"] = self._adam_signature(grp, vparams)
        return D["adam_table"]

    def _adam_signature(self, grp, vparams):
        """What the device table and a captured iteration have baked in: every data pointer of the control network's Adam
        state and the hyper-parameters of every group (the pair-grid network's groups are stepped by torch's capturable
        Adam inside the same graph).  `optimizer.load_state_dict` (resume), a re-allocated parameter or state tensor, or an
        lr schedule changes it -- `_graph_step` then drops the table and the captured graphs instead of stepping through
        stale pointers or frozen hyper-parameters."""
        st = self.optimizer.state
        if not all(p in st and "exp_avg" in st[p] for p in vparams):
            return None
        ptrs = tuple(q.data_ptr() for p in vparams for q in (p, st[p]["exp_avg"], st[p]["exp_avg_sq"], st[p]["step"]))
        hyper = tuple((float(g["lr"]) if not torch.is_tensor(g["lr"]) else id(g["lr"]), tuple(g["betas"]), float(g["eps"]),
                       tuple(st[p]["exp_avg"].data_ptr() if p in st and "exp_avg" in st[p] else 0 for p in g["params"]))
                      for g in self.optimizer.param_groups if g is not grp)
        return (ptrs, float(grp["lr"]) if not torch.is_tensor(grp["lr"]) else None, tuple(grp["betas"]), float(grp["eps"]),
                hyper)

    def _make_capturable(self):
        dev = self.solver.x0.device
        for g in self.optimizer.param_groups:            # device-side step counters: required under capture
            g["capturable"] = True
            for p in g["params"]:
                st = self.optimizer.state.get(p)
                if st and torch.is_tensor(st.get("step")) and not st["step"].is_cuda:
                    st["step"] = st["step"].to(dev)

    def _optimizer_signature(self):
        """Everything a captured iteration has baked in from the optimiser, fused table or not: the data pointers of every
        parameter and of its Adam state, and each group's hyper-parameters (torch's capturable Adam freezes lr / betas / eps
        into the captured kernels' arguments)."""
        st = self.optimizer.state
        sig = []
        for g in self.optimizer.param_groups:
            lr = g.get("lr")
            sig.append((float(lr) if not torch.is_tensor(lr) else ("t", lr.data_ptr()), tuple(g.get("betas", ())),
                        float(g.get("eps", 0.0)), float(g.get("weight_decay", 0.0)),
                        tuple((p.data_ptr(),) + tuple(v.data_ptr() for v in st.get(p, {}).values() if torch.is_tensor(v))
                              for p in g["params"])))
        return tuple(sig)

    def _drop_stale_graphs(self):
        """`optimizer.load_state_dict` (resume), an lr change or a re-allocated state tensor after a capture: the captured
        graphs would keep stepping the old tensors with the old hyper-parameters -- drop them (the next iterations warm up
        and capture again).  Checked for every captured body, with or without the fused Adam table."""
        if not any(not (isinstance(k, tuple) and k and k[0] == "warm") for k in self._graphs):
            return
        if getattr(self, "_graphs_sig", None) != self._optimizer_signature():
            self._graphs = {}
            if self._dev is not None:
                for k in ("adam_table", "adam_sig", "adam_scratch", "adam_side", "adam_side_sig"):
                    self._dev.pop(k, None)
            self._make_capturable()

    def _drop_stale_adam_state(self):
        D = self._dev
        if D is None or "adam_table" not in D:
            return
        vparams = list(self.solver.neural_sde.nabla_V.parameters())
        if len(self._groups_main) != 1 or self._adam_signature(self._groups_main[0], vparams) != D["adam_sig"]:
            for k in ("adam_table", "adam_sig", "adam_scratch", "adam_side", "adam_side_sig"):
                D.pop(k, None)
            self._graphs = {}          # warm-up iterations run eagerly again (they rebuild the table), then a new capture
            self._make_capturable()

    def _graph_step(self, loss_kwargs):
        from .streams import private_stream
        solver = self.solver
        dev = solver.x0.device
        if self.sync_timing:
            torch.cuda.synchronize(dev)
        start = time.time()              # (the reference starts its timer at the top of the loop body, main.py:280: the host-side
        manual = self._manual_ok(loss_kwargs)        #  checks below are part of the iteration)
        self._drop_stale_graphs()
        if not manual:
            self._flush_M()
        else:
            self._drop_stale_adam_state()
        body = (lambda: self._body_manual(loss_kwargs)) if manual else (lambda: self._body_dev(loss_kwargs))
        key = tuple(sorted((k, id(v) if callable(v) else v) for k, v in loss_kwargs.items()))
        key = (("manual",) + key) if manual else key
        entry = self._graphs.get(key)
        if entry is None:
            self._graph_state()
            # first calls with this signature: `graph_warmup` iterations run eagerly on a side stream (they are real
            # training iterations), the next one is captured while it runs
            n = self._graphs.setdefault(("warm",) + key, 0)
            # (the manual body needs two eager iterations: the second one is the first to run the pair-grid network's
            #  update, which creates that group's Adam state -- it must exist before a capture)
            if not self.capture_graphs:
                vals = body()           # (transport that cannot be captured / "nocapture": the same body on the current stream)
                mode = "body-eager" if manual else "autograd-body-eager"
            elif n < (max(2, self.graph_warmup) if manual else self.graph_warmup) or (manual and not self._m_pending):
                self._graphs[("warm",) + key] = n + 1
                side = private_stream(dev, "capture")        # (the stream the capture will run on: socmx/streams.py)
                side.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(side):
                    vals = body()
                torch.cuda.current_stream(dev).wait_stream(side)
                vals.record_stream(torch.cuda.current_stream(dev))
                mode = "graph-warmup"
            else:
                g = torch.cuda.CUDAGraph()
                # Every stream that joins this capture -- the capture stream itself, the solver's second stream -- is a stream of
                # this package's own (socmx/streams.py), and a sharded body's collectives are launches of the shard's own RCCL
                # communicators on those streams (socmx/rccl.py): torch's process group, its pooled streams and its watchdog thread
                # -- which polls the events of earlier eager calls with hipEventQuery while this thread captures -- never meet the
                # capture.  (Rounds 4-5 captured ProcessGroupNCCL calls, drew the capture stream from torch's pool and slept 0.35 s
                # in front of the capture to let the watchdog drain; all three are gone.)
                cmode = "thread_local" if solver.shard is not None else "global"
                err = None
                try:
                    with torch.cuda.graph(g, stream=private_stream(dev, "capture"), capture_error_mode=cmode):
                        static_vals = body()
                except Exception as e:                       # noqa: BLE001 -- whatever the capture choked on, training goes on
                    err = e
                if self._multi_rank and not solver.shard.agree(err is None) and err is None:
                    # (a capture records, it does not execute: every rank is here, none has issued a collective of this iteration)
                    err = RuntimeError("another rank could not capture the iteration")
                if err is not None:
                    return self._capture_failed(err, loss_kwargs)
                # (does this body leave its scalars in the history array?  decided while it was captured)
                self._graphs[key] = entry = (g, static_vals, bool(self._dev.get("hist_written")))
                self._graphs_sig = self._optimizer_signature()
                g.replay()                                   # capture does not execute: this replay IS the iteration
                vals = self._replayed_values(entry)
                mode = "graph-capture"
        elif manual and not self._m_pending:
            # a flush (checkpoint, eager fallback) consumed the outstanding update the captured graph starts with: this one
            # iteration runs the same body eagerly (it skips the update), the next one replays again
            vals = body()
            mode = "body-eager (after a flush)"
        else:
            entry[0].replay()
            vals = self._replayed_values(entry)
            mode = "graph-replay"
        if self.sync_timing:
            torch.cuda.synchronize(dev)
        time_per_iteration = time.time() - start
        self.itr += 1
        self.normalization_const = self._dev["norm"]         # (a live view of the device-side normaliser)
        info = dict(loss=vals[0], time_per_iteration=time_per_iteration, weight_mean=vals[1], weight_std=vals[2], mode=mode,
                    norm_before=vals[6], out=(None, vals[7] if vals.numel() > 7 else None, None, None, None, vals[1],
                                              vals[2], None))
        if self.grad_telemetry:
            info.update(grad_norm_sqd=vals[3], EMA_grad_norm_sqd=vals[4], sqd_norm_EMA_grad=vals[5])
        return info

    def _replayed_values(self, entry):
        """The scalars of the iteration just replayed: a VIEW of its row of the history array when the body writes one (no device
        copy on the iteration's stream), else a clone of the graph's static output (the next replay overwrites it)."""
        static_vals, in_hist = entry[1], entry[2]
        hist = self._dev.get("hist")
        if in_hist and hist is not None and self.itr < hist.shape[0]:
            row = hist[self.itr]
            return row if static_vals.numel() > 7 else row[:7]
        return static_vals.clone()

    def _capture_failed(self, err, loss_kwargs):
        """A body that cannot be captured (an operation that synchronises, a library call without capture support on this
        stack, ...) must not end a training run that `backend.hip_graph: True` started by default: say so once, drop graph
        mode for the rest of the run and take this iteration -- nothing of it has executed -- on the eager path, from the
        device-side state the warm-up iterations left (normaliser, telemetry EMAs, outstanding pair-grid-network update)."""
        import warnings
        msg = f"backend.hip_graph: capture failed ({type(err).__name__}: {err}); the run continues on the eager two-stream iteration"
        warnings.warn(msg)
        self._log(msg)
        torch.cuda.synchronize(self.solver.x0.device)
        self._flush_M()
        self._sync_from_device_state()
        if self._dev is not None and self.itr > 0:
            self._ema_grad = [g.clone() for g in self._dev["ema_grad"]]
            self._ema_grad_norm_sqd = self._dev["ema_gn"].clone()
        self.hip_graph = False
        self._graphs.clear()
        self._eager_reason = "capture failed"
        return self._eager_step(**loss_kwargs)

    def _graph_eligible(self, loss_kwargs):
        # (sharded: only the autograd-free body carries its collectives inside the graph; anything else runs eagerly)
        return (self.hip_graph and not loss_kwargs.get("compute_control_objective", False)
                and not loss_kwargs.get("verbose", False)
                and (self.solver.shard is None or self._manual_ok(loss_kwargs)))

    def _sync_from_device_state(self):
        """An eager iteration in hipGraph mode (checkpoint iterations: control-objective bursts, verbose prints) works
        on the host-side mirrors of the device state, and writes them back afterwards."""
        D = self._dev
        if D is not None:
            self.normalization_const = D["norm"].clone()

    def _sync_to_device_state(self):
        D = self._dev
        if D is not None:
            nc = self.normalization_const
            D["norm"].copy_(nc.detach().reshape(()) if torch.is_tensor(nc) else torch.tensor(float(nc)))
            D["itr"].fill_(float(self.itr))
            if self._ema_grad is not None and self._ema_grad is not D["ema_grad"]:
                torch._foreach_copy_(D["ema_grad"], self._ema_grad)
            if self._ema_grad_norm_sqd is not None:
                D["ema_gn"].copy_(self._ema_grad_norm_sqd.detach().reshape(()))

    def step(self, **loss_kwargs):
        """One iteration (main.py:280-359).  The returned dict's `mode` names the schedule THIS iteration took -- "graph-replay",
        "graph-capture", "graph-warmup", "body-eager", "eager (<why>)" -- so that a run whose iterations silently left the
        replayed graph (2.3x the time at configs[2]) shows it in its telemetry (main.py keeps it in training_info)."""
        if self._graph_eligible(loss_kwargs):
            return self._graph_step(loss_kwargs)
        self._eager_why = (getattr(self, "_eager_reason", None) or ("hip_graph off" if not self.hip_graph else
                           "control-objective burst" if loss_kwargs.get("compute_control_objective") else
                           "verbose" if loss_kwargs.get("verbose") else "sharded run outside the autograd-free body"))
        if self.hip_graph:
            self._graph_state()
            self._flush_M()
            self._sync_from_device_state()
            if self._dev is not None and self.itr > 0:
                self._ema_grad = [g.clone() for g in self._dev["ema_grad"]]
                self._ema_grad_norm_sqd = self._dev["ema_gn"].clone()
            try:
                return self._eager_step(**loss_kwargs)
            finally:
                self._sync_to_device_state()
        return self._eager_step(**loss_kwargs)

    def _eager_step(self, **loss_kwargs):
        solver = self.solver
        dev = solver.x0.device
        if self.sync_timing and dev.type == "cuda":
            torch.cuda.synchronize(dev)
        start = time.time()
        shard = solver.shard
        # a sharded eager iteration keeps the pair-grid network's backward inside loss.backward(): every gradient exists
        # before the iteration's ONE collective
        solver.defer_M_backward = self.defer_M and shard is None   # only for this call: direct users of .loss() get the full graph
        solver.defer_weight_stats = shard is not None
        try:
            out = solver.loss(self.batch_size, algorithm=self.algorithm, use_warm_start=False,
                              use_stopping_time=bool(getattr(solver.neural_sde, "use_stopping_time", False)),
                              **loss_kwargs)
        finally:
            solver.defer_M_backward = False
            solver.defer_weight_stats = False
        objective, weight_mean = out[0], out[5]
        if self.algorithm in ("SOCM", "SOCM_const_M", "SOCM_exp", "SOCM_adjoint", "cross_entropy"):
            loss = objective / self.normalization_const                  # main.py:313-320
        elif self.algorithm == "variance":
            loss = objective / self.normalization_const ** 2             # main.py:321-322
        else:
            loss = objective
        loss.backward()                                                  # main.py:323
        if shard is not None:
            # ONE flat all-reduce per iteration: every gradient + the loss value + the ranks' (n, mean, M2) slots + (when computed)
            # this rank's share of the weighted L2 error, which solver.loss already divided by the GLOBAL (K+1) B
            from .dist import mean_std_from_slots
            extra = [loss.detach(), solver.__dict__.pop("_local_w_sums")] + ([out[1].detach()] if out[1] is not None else [])
            reduced = shard.allreduce_gradients([p for g in self.optimizer.param_groups for p in g["params"]], extra=extra)
            loss_val = reduced[0].reshape(())
            weight_mean, weight_std = mean_std_from_slots(reduced[1])
            out = (out[0], reduced[2].reshape(()) if out[1] is not None else None) + tuple(out[2:5]) + (weight_mean, weight_std) \
                + tuple(out[7:])
        else:
            loss_val = loss.detach()
        pending = solver.__dict__.pop("_pending_M", None)
        telemetry = {}
        if self.grad_telemetry and pending is None:
            telemetry = self._grad_telemetry()
        with torch.no_grad():
            if pending is not None:
                self._step_groups(self._groups_main)                     # nabla_V: needed by the next rollout
                telemetry = self._finish_M_on_side_stream(pending, dev)  # (gradient telemetry rides on that stream)
            else:
                self.optimizer.step()                                    # main.py:347-349
            self.optimizer.zero_grad()
            if self.sync_timing and dev.type == "cuda":
                torch.cuda.synchronize(dev)
            time_per_iteration = time.time() - start                    # main.py:351-352
            self.normalization_const = compute_EMA(weight_mean.detach(), self.normalization_const,
                                                   EMA_coeff=self.coeff, itr=self.itr)   # main.py:354-359
        self.itr += 1
        # the loss outputs leave DETACHED: the backward pass is done, and a caller that kept the objective's autograd graph
        # alive would also keep the parameters' AccumulateGrad nodes alive -- pinned to the stream of this eager
        # iteration, which breaks a later hipGraph capture (see _body_dev)
        out = tuple(o.detach() if torch.is_tensor(o) else o for o in out)
        return dict(loss=loss_val, time_per_iteration=time_per_iteration, weight_mean=weight_mean.detach(),
                    weight_std=out[6], out=out, mode=f"eager ({getattr(self, '_eager_why', None) or 'hip_graph off'})", **telemetry)
