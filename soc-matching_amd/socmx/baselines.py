"""The reference's other losses on the rollout buffers: SOCM_const_M, SOCM_exp, SOCM_adjoint
(method.py:289-478, 722-749), cross_entropy (751-785), variance / log-variance / moment (787-856).

All of them are reductions over the same `(states, noises, controls, nabla_V, w)` the fused rollout produces; they are
written in the restated operands of `socmx.loss`:
    v_j = -( sqrt(lmbd) sqrt(dt_j) S^-T eps_j + dt_j S^-T u_j ),   q_j = dt_j nabla_f(X_j) + nabla_b(X_j)^T v_j
so that e.g. the constant-M target is just a reverse running sum of q.
Two executions: CUDA tensors -> one launch per family member of csrc/socmx_baselines.hip (matching family: target scan or the
K-step costate recursion, then the SOCM residual kernel; Girsanov family: integrand kernel + its backward), wrapped in
autograd Functions; CPU tensors -> the plain torch forms below (BASELINE config 0, gloo tests).  Checked against
reference-generated fixtures in tests/test_host_cpu.py and tests/test_gpu_parity.py.
"""
import math

import torch

from . import _lib
from . import loss as L


# --------------------------------------------------------------------------------------
# GPU: one launch per family member (csrc/socmx_baselines.hip) + the residual kernels of the SOCM loss
# --------------------------------------------------------------------------------------

class _MatchingHip(torch.autograd.Function):
    """objective = sum w |sigma^T (nabla_V - target)|^2 / ((K+1) B) with the target of SOCM_const_M (kind 0), SOCM_exp (1: also
    d target / d gamma) or SOCM_adjoint (2: the K-step costate recursion in one kernel): socmx_matching_target_f32 +
    socmx_socm_residual_f32."""

    @staticmethod
    def forward(ctx, nabla_V, gamma, kind, pb, ts, T, dt, ops, states, weight):
        Lh, f = _lib.lib(), _lib.ptr
        Kp, B, d = nabla_V.shape
        K = Kp - 1
        dev = nabla_V.device
        c = lambda t: t.detach().to(torch.float32).contiguous()
        nv, w, tsc = c(nabla_V), c(weight), c(ts)
        target = torch.empty(Kp, B, d, dtype=torch.float32, device=dev)
        dtarget = torch.empty_like(target) if kind == 1 else None
        gam = c(gamma).reshape(1) if kind == 1 else None
        G = torch.empty_like(target)
        obj = torch.zeros(1, dtype=torch.float32, device=dev)
        with _lib.on_device(dev):
            _lib.check(Lh.socmx_matching_target_f32(kind, pb.c_struct(), K, B, f(tsc), float(T), float(dt), f(gam),
                                                    f(ops["q"]), f(ops["v"]), f(ops["gT"]), f(c(states)), f(target),
                                                    f(dtarget), _lib.stream_ptr(dev)), "socmx_matching_target_f32")
            _lib.check(Lh.socmx_socm_residual_f32(pb.c_struct(), K, B, f(target), f(nv), f(w), 1.0 / (Kp * B), f(G), f(obj),
                                                  f(_lib.objective_workspace(dev, K, B)), _lib.stream_ptr(dev)),
                       "socmx_socm_residual_f32")
        ctx.save_for_backward(G, *([dtarget] if kind == 1 else []))
        ctx.gamma_shape = gamma.shape if kind == 1 else None
        return obj[0]

    @staticmethod
    def backward(ctx, gout):
        G, *rest = ctx.saved_tensors
        g_gamma = None
        if rest and ctx.needs_input_grad[1]:
            g_gamma = (-(G * rest[0]).sum() * gout).reshape(ctx.gamma_shape)     # d obj / d target = -G
        return G * gout, g_gamma, None, None, None, None, None, None, None, None


class _GirsanovHip(torch.autograd.Function):
    """total_m = sum_i c[i,m] of method.py:751-829 (socmx_girsanov_fwd_f32) as a function of nabla_V on the trajectory;
    backward: socmx_girsanov_bwd_f32."""

    @staticmethod
    def forward(ctx, nabla_V, pb, ts, lmbd, with_f, noises, controls, states, frac, stop):
        Lh, f = _lib.lib(), _lib.ptr
        Kp, B, d = nabla_V.shape
        K = Kp - 1
        dev = nabla_V.device
        c = lambda t: None if t is None else t.detach().to(torch.float32).contiguous()
        nv, tsc, noises, controls, states, frac, stop = map(c, (nabla_V, ts, noises, controls, states, frac, stop))
        cim = torch.empty(K, B, dtype=torch.float32, device=dev)
        with _lib.on_device(dev):
            _lib.check(Lh.socmx_girsanov_fwd_f32(pb.c_struct(), K, B, float(lmbd), int(bool(with_f)), f(tsc), f(nv), f(noises),
                                                 f(controls), f(states), f(frac), f(stop), f(cim), _lib.stream_ptr(dev)),
                       "socmx_girsanov_fwd_f32")
        ctx.save_for_backward(nv, tsc, noises, controls, *[t for t in (frac, stop) if t is not None])
        ctx.meta = (pb, float(lmbd), frac is not None, stop is not None)
        return cim.sum(0)

    @staticmethod
    def backward(ctx, gtotal):
        Lh, f = _lib.lib(), _lib.ptr
        nv, tsc, noises, controls, *rest = ctx.saved_tensors
        pb, lmbd, has_frac, has_stop = ctx.meta
        frac = rest[0] if has_frac else None
        stop = rest[-1] if has_stop else None
        Kp, B, d = nv.shape
        G = torch.empty_like(nv)
        gt = gtotal.detach().to(torch.float32).contiguous()
        with _lib.on_device(nv.device):
            _lib.check(Lh.socmx_girsanov_bwd_f32(pb.c_struct(), Kp - 1, B, lmbd, f(tsc), f(nv), f(noises), f(controls), f(frac),
                                                 f(stop), f(gt), f(G), _lib.stream_ptr(nv.device)), "socmx_girsanov_bwd_f32")
        return G, None, None, None, None, None, None, None, None, None


def _on_gpu(t):
    return t.is_cuda and t.dtype == torch.float32


def _rev_cumsum(x):
    """out[i] = sum_{j >= i} x[j] along dim 0."""
    return torch.flip(torch.cumsum(torch.flip(x, [0]), 0), [0])


def _matching_objective(pb, nabla_V, target, weight):
    r = (nabla_V - target) @ pb.sigma
    return torch.sum(r * r * weight.reshape(1, -1, 1)) / (nabla_V.shape[0] * nabla_V.shape[1])


def socm_const_m(pb, ts, lmbd, states, noises, controls, nabla_V, weight):
    """M = I: target[i] = sum_{j>=i}^{K-1} q_j + nabla_g(X_K)   (method.py:289-369)."""
    if _on_gpu(nabla_V):
        ops = L.socm_operands_hip(pb, ts, lmbd, states, noises, controls)
        return _MatchingHip.apply(nabla_V, None, 0, pb, ts, 0.0, 0.0, ops, states, weight)
    v, q, gT = L.socm_operands(pb, ts, lmbd, states, noises, controls)
    tail = torch.cat([_rev_cumsum(q), torch.zeros_like(q[:1])], 0)
    return _matching_objective(pb, nabla_V, tail + gT.unsqueeze(0), weight)


def socm_exp(pb, ts, T, lmbd, gamma, states, noises, controls, nabla_V, weight):
    """M(t,s) = e^{-gamma (s-t)} I  (method.py:371-478):
    target[i] = e^{gamma t_i} sum_{j>=i}^{K-1} e^{-gamma t_j} (q_j + gamma v_j) + e^{-gamma (T - t_i)} nabla_g(X_K)."""
    if _on_gpu(nabla_V) and torch.is_tensor(gamma):
        ops = L.socm_operands_hip(pb, ts, lmbd, states, noises, controls)
        return _MatchingHip.apply(nabla_V, gamma, 1, pb, ts, T, 0.0, ops, states, weight)
    v, q, gT = L.socm_operands(pb, ts, lmbd, states, noises, controls)
    decay = torch.exp(-gamma * ts)
    inner = decay[:-1].reshape(-1, 1, 1) * (q + gamma * v)
    tail = torch.cat([_rev_cumsum(inner), torch.zeros_like(inner[:1])], 0) / decay.reshape(-1, 1, 1)
    terminal = torch.exp(-gamma * (T - ts)).reshape(-1, 1, 1) * gT.unsqueeze(0)
    return _matching_objective(pb, nabla_V, tail + terminal, weight)


def socm_adjoint(pb, ts, dt, states, nabla_V, weight):
    """Adjoint (costate) target, trapezoidal in time with the constant step T/K (method.py:722-749)."""
    if _on_gpu(nabla_V) and pb.d <= 64:
        gT = pb.nabla_g(states[-1]).to(torch.float32).contiguous()
        return _MatchingHip.apply(nabla_V, None, 2, pb, ts, 0.0, dt, dict(q=None, v=None, gT=gT), states, weight)
    nf = pb.nabla_f(ts, states)
    a = pb.nabla_g(states[-1])
    out = [a]
    for k in range(states.shape[0] - 2, -1, -1):
        # ((nabla_b_k + nabla_b_{k+1}) / 2)^T-contracted a, without the dense Jacobians
        jb = 0.5 * (pb.nabla_b_T_apply(states[k], a) + pb.nabla_b_T_apply(states[k + 1], a))
        a = a + dt * (0.5 * (nf[k] + nf[k + 1]) + jb)
        out.append(a)
    out.reverse()
    return _matching_objective(pb, nabla_V, torch.stack(out), weight)


def _girsanov_terms(pb, ts, lmbd, learned, noises, controls, frac, stop_indicators, with_f, states):
    """sum_k integrand dt and sum_k stochastic sqrt(dt) of method.py:751-829."""
    dts = (ts[1:] - ts[:-1]).reshape(-1, 1) if frac is None else frac
    det = -(1 / lmbd) * (learned[:-1] * controls).sum(2) + (1 / (2 * lmbd)) * (learned ** 2).sum(2)[:-1]
    if with_f:
        det = det - (1 / lmbd) * pb.f(ts[0], states)[:-1]
    sto = -math.sqrt(1 / lmbd) * (learned[:-1] * noises).sum(2)
    if stop_indicators is not None:
        det, sto = det * stop_indicators[:-1], sto * stop_indicators[:-1]
    return (det * dts).sum(0), (sto * torch.sqrt(dts)).sum(0)


def cross_entropy(pb, ts, lmbd, states, noises, controls, nabla_V, weight, frac=None):
    if _on_gpu(nabla_V):
        total = _GirsanovHip.apply(nabla_V, pb, ts, lmbd, False, noises, controls, states, frac, None)
        return torch.mean(total * weight)
    learned = -(nabla_V @ pb.sigma)
    det, sto = _girsanov_terms(pb, ts, lmbd, learned, noises, controls, frac, None, False, states)
    return torch.mean((det + sto) * weight)


def variance_family(algorithm, pb, ts, lmbd, states, noises, controls, nabla_V, weight, y0, add_weights=False,
                    frac=None, stop_indicators=None):
    if _on_gpu(nabla_V):
        total = _GirsanovHip.apply(nabla_V, pb, ts, lmbd, True, noises, controls, states, frac, stop_indicators)
    else:
        learned = -(nabla_V @ pb.sigma)
        det, sto = _girsanov_terms(pb, ts, lmbd, learned, noises, controls, frac, stop_indicators, True, states)
        total = det + sto
    total = total - (1 / lmbd) * pb.g(states[-1])
    if algorithm == "variance":
        total = torch.exp(total)
    elif algorithm == "moment":
        total = total + y0
    w2 = weight if add_weights else torch.ones_like(weight)
    if algorithm == "moment":
        return torch.mean(total ** 2 * w2)
    n = total.shape[0]
    return n / (n - 1) * (torch.mean(total ** 2 * w2) - torch.mean(total * w2) ** 2)
