"""The reference's other losses on the rollout buffers: SOCM_const_M, SOCM_exp, SOCM_adjoint
(method.py:289-478, 722-749), cross_entropy (751-785), variance / log-variance / moment (787-856).

All of them are cheap reductions over the same `(states, noises, controls, nabla_V, w)` the fused rollout and the
library-GEMM network evaluation produce; they are written in the restated operands of `socmx.loss`:
    v_j = -( sqrt(lmbd) sqrt(dt_j) S^-T eps_j + dt_j S^-T u_j ),   q_j = dt_j nabla_f(X_j) + nabla_b(X_j)^T v_j
so that e.g. the constant-M target is just a reverse cumulative sum of q.  Checked against reference-generated
fixtures in tests/test_host_cpu.py.
"""
import math

import torch

from . import loss as L


def _rev_cumsum(x):
    """out[i] = sum_{j >= i} x[j] along dim 0."""
    return torch.flip(torch.cumsum(torch.flip(x, [0]), 0), [0])


def _matching_objective(pb, nabla_V, target, weight):
    r = (nabla_V - target) @ pb.sigma
    return torch.sum(r * r * weight.reshape(1, -1, 1)) / (nabla_V.shape[0] * nabla_V.shape[1])


def socm_const_m(pb, ts, lmbd, states, noises, controls, nabla_V, weight):
    """M = I: target[i] = sum_{j>=i}^{K-1} q_j + nabla_g(X_K)   (method.py:289-369)."""
    v, q, gT = L.socm_operands(pb, ts, lmbd, states, noises, controls)
    tail = torch.cat([_rev_cumsum(q), torch.zeros_like(q[:1])], 0)
    return _matching_objective(pb, nabla_V, tail + gT.unsqueeze(0), weight)


def socm_exp(pb, ts, T, lmbd, gamma, states, noises, controls, nabla_V, weight):
    """M(t,s) = e^{-gamma (s-t)} I  (method.py:371-478):
    target[i] = e^{gamma t_i} sum_{j>=i}^{K-1} e^{-gamma t_j} (q_j + gamma v_j) + e^{-gamma (T - t_i)} nabla_g(X_K)."""
    v, q, gT = L.socm_operands(pb, ts, lmbd, states, noises, controls)
    decay = torch.exp(-gamma * ts)
    inner = decay[:-1].reshape(-1, 1, 1) * (q + gamma * v)
    tail = torch.cat([_rev_cumsum(inner), torch.zeros_like(inner[:1])], 0) / decay.reshape(-1, 1, 1)
    terminal = torch.exp(-gamma * (T - ts)).reshape(-1, 1, 1) * gT.unsqueeze(0)
    return _matching_objective(pb, nabla_V, tail + terminal, weight)


def socm_adjoint(pb, ts, dt, states, nabla_V, weight):
    """Adjoint (costate) target, trapezoidal in time with the constant step T/K (method.py:722-749)."""
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
    learned = -(nabla_V @ pb.sigma)
    det, sto = _girsanov_terms(pb, ts, lmbd, learned, noises, controls, frac, None, False, states)
    return torch.mean((det + sto) * weight)


def variance_family(algorithm, pb, ts, lmbd, states, noises, controls, nabla_V, weight, y0, add_weights=False,
                    frac=None, stop_indicators=None):
    learned = -(nabla_V @ pb.sigma)
    det, sto = _girsanov_terms(pb, ts, lmbd, learned, noises, controls, frac, stop_indicators, True, states)
    total = det + sto - (1 / lmbd) * pb.g(states[-1])
    if algorithm == "variance":
        total = torch.exp(total)
    elif algorithm == "moment":
        total = total + y0
    w2 = weight if add_weights else torch.ones_like(weight)
    if algorithm == "moment":
        return torch.mean(total ** 2 * w2)
    n = total.shape[0]
    return n / (n - 1) * (torch.mean(total ** 2 * w2) - torch.mean(total * w2) ** 2)
