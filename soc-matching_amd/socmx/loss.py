"""SOCM matching loss (reference SOC_matching/method.py:480-720), restated.

With  v_jm = -( sqrt(lmbd) sqrt(dt_j) S^-T eps_jm + dt_j S^-T u_jm ),
      q_jm = dt_j nabla_f(X_jm) + nabla_b(X_jm)^T v_jm,
the reference's least-squares target (method.py:591-690) is

  target[i,m] = sum_{j=i}^{K-1} ( M_ij q_jm - dM_ij v_jm ) + M_iK nabla_g(X_Km)

and  objective = sum_{i,m} w_m | sigma^T (nabla_V(t_i,X_im) - target[i,m]) |^2 / ((K+1) B)
(method.py:702-720).  This needs O(Np d^2 + Kp B d) memory instead of the
reference's (Kp,Kp,B,d,d) intermediates (method.py:614-618) and is algebraically
identical (checked against the reference-generated fixtures in tests/).

Two executions of the same math:
  * CUDA tensors -> HIP kernels through `libsocmx.so` (csrc/socmx_loss.hip) wrapped in
    `torch.autograd.Function`s; missing library raises.
  * CPU tensors  -> plain torch (BASELINE config 0 plumbing; gloo tests).
On the GPU neither network is a library GEMM: nabla_V's values on the Kp*B trajectory rows
come from the rollout kernel and its gradients from socmx_unet_backward_f32, the M network on
the Np pairs runs on socmx_mnet_* (socmx/nets.py); the CPU path uses torch autograd.
"""
import math

import torch

from . import _lib


# --------------------------------------------------------------------------------------
# pair grid
# --------------------------------------------------------------------------------------

def pair_index(K, device):
    """(i, j) index vectors of the Np=(K+1)(K+2)/2 pairs t_i <= s_j, i-major (method.py:533-547)."""
    ii, jj = torch.triu_indices(K + 1, K + 1, device=device)
    return ii, jj


def pair_times(ts, T, K):
    """The reference builds s by `linspace(t_k, T, K+1-k)` per row (method.py:535-537); those values
    differ from ts[j] by <= 1 ulp.  We reproduce the linspace values exactly in closed form:
    s = t_k + step*(j-k) for the lower half, T - step*(n-1-(j-k)) for the upper half (torch.linspace
    semantics), which keeps CPU parity with the fixtures bit-tight."""
    dev = ts.device
    ii, jj = pair_index(K, dev)
    t = ts[ii]
    n = (K + 1 - ii).to(ts.dtype)                      # points in row i
    r = (jj - ii).to(ts.dtype)                          # position inside the row
    step = (T - t) / torch.clamp(n - 1, min=1)
    lower = t + step * r
    upper = T - step * (n - 1 - r)
    s = torch.where(r < torch.floor(n / 2), lower, upper)
    s = torch.where(n == 1, t, s)
    return t, s, ii, jj


# --------------------------------------------------------------------------------------
# operands
# --------------------------------------------------------------------------------------

def socm_operands(pb, ts, lmbd, states, noises, controls, frac=None):
    """v, q (K,B,d) and gT (B,d) in torch (any device)."""
    sit = pb.sigma_inv_t()
    dts = (ts[1:] - ts[:-1]).reshape(-1, 1, 1) if frac is None else frac.unsqueeze(-1)
    vt = noises @ sit.T
    ut = controls @ sit.T
    v = -(math.sqrt(lmbd) * torch.sqrt(dts) * vt + dts * ut)
    xs = states[:-1]
    q = dts * pb.nabla_f(ts[:-1], xs) + pb.nabla_b_T_apply(xs, v)
    gT = pb.nabla_g(states[-1])
    return v, q, gT


# --------------------------------------------------------------------------------------
# target + residual
# --------------------------------------------------------------------------------------

def target_residual_torch(pb, K, M_all, dM_all, q, v, gT, nablaV, w, inv_norm):
    """Dense-scatter formulation in plain torch: one (Kp d x K d) GEMM per operand."""
    d = pb.d
    dev = M_all.device
    Kp = K + 1
    ii, jj = pair_index(K, dev)
    Md = torch.zeros(Kp, Kp, d, d, device=dev, dtype=M_all.dtype).index_put((ii, jj), M_all)
    dMd = torch.zeros(Kp, Kp, d, d, device=dev, dtype=M_all.dtype).index_put((ii, jj), dM_all)
    B = q.shape[1]
    A1 = Md[:, :K].permute(0, 2, 1, 3).reshape(Kp * d, K * d)
    A2 = dMd[:, :K].permute(0, 2, 1, 3).reshape(Kp * d, K * d)
    qm = q.permute(0, 2, 1).reshape(K * d, B)
    vm = v.permute(0, 2, 1).reshape(K * d, B)
    tgt = (A1 @ qm - A2 @ vm).reshape(Kp, d, B).permute(0, 2, 1)
    tgt = tgt + torch.einsum("ikl,ml->imk", Md[:, K], gT)
    r = (nablaV - tgt) @ pb.sigma
    objective = (r * r * w.reshape(1, -1, 1)).sum() * inv_norm
    return objective, tgt


def socm_operands_hip(pb, ts, lmbd, states, noises, controls, frac=None, out=None):
    """socmx_socm_prep_f32: v, q (K,B,d) and gT (B,d), batch-major (read by the forward and the backward kernels).
    `out`: a dict of caller-owned v / q / gT buffers (hipGraph mode keeps them for the deferred contraction backward)."""
    L = _lib.lib()
    K, B, d = noises.shape
    dev = states.device
    f32 = dict(dtype=torch.float32, device=dev)
    if out is not None:
        v, q, gT = out["v"], out["q"], out["gT"]
    else:
        v, q = torch.empty(K, B, d, **f32), torch.empty(K, B, d, **f32)
        gT = torch.empty(B, d, **f32)
    tsc = ts.detach().to(**f32).contiguous()
    with _lib.on_device(dev):
        _lib.check(L.socmx_socm_prep_f32(
            pb.c_struct(), _lib.ptr(tsc), K, B, float(lmbd), _lib.ptr(states.contiguous()),
            _lib.ptr(noises.contiguous()), _lib.ptr(controls.contiguous()),
            _lib.ptr(frac.contiguous()) if frac is not None else None,
            _lib.ptr(v), _lib.ptr(q), _lib.ptr(gT), None, None, None,
            _lib.stream_ptr(dev)), "socmx_socm_prep_f32")
    return dict(v=v, q=q, gT=gT)


class _TargetResidualHip(torch.autograd.Function):
    """objective = socmx_socm_target_fwd_f32(...); grads by socmx_socm_target_bwd_f32."""

    @staticmethod
    def forward(ctx, M_all, dM_all, nablaV, w, ops, pb, K, inv_norm, want_target):
        L = _lib.lib()
        dev = M_all.device
        B, d = ops["gT"].shape
        c = lambda t: t.detach().to(torch.float32).contiguous()
        M_all, dM_all, nablaV, w = map(c, (M_all, dM_all, nablaV, w))
        G = torch.empty_like(nablaV)
        target = torch.empty_like(nablaV)   # the contraction's output; the residual kernel reads it back
        obj = torch.zeros(1, dtype=torch.float32, device=dev)
        with _lib.on_device(dev):
            _lib.check(L.socmx_socm_target_fwd_f32(
                pb.c_struct(), K, B, _lib.ptr(M_all), _lib.ptr(dM_all), _lib.ptr(ops["q"]), _lib.ptr(ops["v"]),
                _lib.ptr(ops["gT"]), _lib.ptr(nablaV), _lib.ptr(w), float(inv_norm), _lib.ptr(target),
                _lib.ptr(G), _lib.ptr(obj), _lib.ptr(_lib.objective_workspace(dev, K, B)), _lib.stream_ptr(dev)),
                "socmx_socm_target_fwd_f32")
        ctx.save_for_backward(G, ops["q"], ops["v"], ops["gT"])
        ctx.dims = (d, K, B, M_all.shape[0])
        if want_target:
            ctx.mark_non_differentiable(target)
            return obj[0], target
        return obj[0]

    @staticmethod
    def backward(ctx, gout, *unused):
        L = _lib.lib()
        G, q, v, gT = ctx.saved_tensors
        d, K, B, Np = ctx.dims
        gM = gdM = gV = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            gM = torch.empty(Np, d, d, dtype=torch.float32, device=G.device)
            gdM = torch.empty(Np, d, d, dtype=torch.float32, device=G.device)
            with _lib.on_device(G.device):
                _lib.check(L.socmx_socm_target_bwd_f32(
                    d, K, B, _lib.ptr(G), _lib.ptr(q), _lib.ptr(v), _lib.ptr(gT), _lib.ptr(gM), _lib.ptr(gdM),
                    _lib.stream_ptr(G.device)), "socmx_socm_target_bwd_f32")
            gM = gM * gout
            gdM = gdM * gout
        if ctx.needs_input_grad[2]:
            gV = G * gout
        return gM, gdM, gV, None, None, None, None, None, None


def target_fwd_net(pb, K, net, dnet, delta, gam, ops, nablaV, w, inv_norm, G=None, obj=None):
    """socmx_socm_target_fwd_net_f32 on plain tensors: (objective (1,), G = d obj / d nablaV, target); `G` may be a
    caller-owned buffer, `obj` a caller-owned (1,) accumulator the caller has ZEROED (the kernels add into it)."""
    L = _lib.lib()
    dev = net.device
    B, d = ops["gT"].shape
    if G is None:
        G = torch.empty_like(nablaV)
    target = torch.empty_like(nablaV)
    if obj is None:
        obj = torch.zeros(1, dtype=torch.float32, device=dev)
    with _lib.on_device(dev):
        _lib.check(L.socmx_socm_target_fwd_net_f32(
            pb.c_struct(), K, B, _lib.ptr(net), _lib.ptr(dnet), _lib.ptr(delta), _lib.ptr(gam), _lib.ptr(ops["q"]),
            _lib.ptr(ops["v"]), _lib.ptr(ops["gT"]), _lib.ptr(nablaV), _lib.ptr(w), float(inv_norm),
            _lib.ptr(target), _lib.ptr(G), _lib.ptr(obj), _lib.ptr(_lib.objective_workspace(dev, K, B)),
            _lib.stream_ptr(dev)), "socmx_socm_target_fwd_net_f32")
    return obj, G, target


def target_bwd_net(d, K, B, G, ops, gout, net, dnet, delta, gam, g_net=None, g_dnet=None):
    """socmx_socm_target_bwd_net_f32 on plain tensors: (g_net, g_dnet, partial sums of d obj / d gamma); `gout` (1,) is
    the upstream gradient on the device; g_net / g_dnet may be caller-owned buffers (hipGraph mode keeps them)."""
    L = _lib.lib()
    dev = G.device
    Np = net.shape[0]
    nb = (d + 15) // 16
    if g_net is None:
        g_net = torch.empty(Np, d, d, dtype=torch.float32, device=dev)
        g_dnet = torch.empty(Np, d, d, dtype=torch.float32, device=dev)
    part = torch.empty(Np * nb * nb, dtype=torch.float32, device=dev)
    with _lib.on_device(dev):
        _lib.check(L.socmx_socm_target_bwd_net_f32(
            d, K, B, _lib.ptr(G), _lib.ptr(ops["q"]), _lib.ptr(ops["v"]), _lib.ptr(ops["gT"]), _lib.ptr(gout),
            _lib.ptr(net), _lib.ptr(dnet), _lib.ptr(delta), _lib.ptr(gam), _lib.ptr(g_net), _lib.ptr(g_dnet),
            _lib.ptr(part), _lib.stream_ptr(dev)), "socmx_socm_target_bwd_net_f32")
    return g_net, g_dnet, part


class _TargetResidualNetHip(torch.autograd.Function):
    """Same objective with M = e I + (1-e) net, dM = gamma e (net - I) + (1-e) dnet formed inside the kernels
    (socmx_socm_target_{fwd,bwd}_net_f32): M and dM/ds never exist in HBM."""

    @staticmethod
    def forward(ctx, net, dnet, gamma, nablaV, w, delta, ops, pb, K, inv_norm):
        B, d = ops["gT"].shape
        c = lambda t: t.detach().to(torch.float32).contiguous()
        net, dnet, nablaV, w, delta = map(c, (net, dnet, nablaV, w, delta))
        gam = gamma.detach().to(torch.float32).reshape(1).contiguous()
        obj, G, _ = target_fwd_net(pb, K, net, dnet, delta, gam, ops, nablaV, w, inv_norm)
        ctx.save_for_backward(G, ops["q"], ops["v"], ops["gT"], net, dnet, delta, gam)
        ctx.dims = (d, K, B, net.shape[0])
        ctx.gamma_shape = gamma.shape
        return obj[0]

    @staticmethod
    def backward(ctx, gout):
        G, q, v, gT, net, dnet, delta, gam = ctx.saved_tensors
        d, K, B, Np = ctx.dims
        g_net = g_dnet = g_gamma = gV = None
        gout = gout.detach().to(torch.float32).reshape(1).contiguous()
        if any(ctx.needs_input_grad[:3]):
            g_net, g_dnet, part = target_bwd_net(d, K, B, G, dict(q=q, v=v, gT=gT), gout, net, dnet, delta, gam)
            if ctx.needs_input_grad[2]:
                g_gamma = part.sum().reshape(ctx.gamma_shape)
        if ctx.needs_input_grad[3]:
            gV = G * gout
        return g_net, g_dnet, g_gamma, gV, None, None, None, None, None, None


class _StoppingTargetHip(torch.autograd.Function):
    """target (Kp,B,d) of the stopping-time SOCM loss from the two network evaluations, the pair grid, the samples' stopping
    times and (gamma, gamma2, gamma3) (socmx_socm_stopping_target_{fwd,bwd}_f32): the gates of models.py:341-392, their
    s-derivatives and -- backward -- their derivatives in the three gammas are formed per (pair, sample) inside the kernels."""

    @staticmethod
    def forward(ctx, gamma, gamma2, gamma3, N0, N1, dN0, dN1, t_vec, s_vec, tau, ops, K, T_model):
        L = _lib.lib()
        B, d = ops["gT"].shape
        c = lambda t: t.detach().to(torch.float32).contiguous()
        N0, N1, dN0, dN1, t_vec, s_vec, tau = map(c, (N0, N1, dN0, dN1, t_vec, s_vec, tau))
        gam = torch.cat([c(gamma).reshape(1), c(gamma2).reshape(1), c(gamma3).reshape(1)])
        target = torch.empty(K + 1, B, d, dtype=torch.float32, device=N0.device)
        with _lib.on_device(N0.device):
            _lib.check(L.socmx_socm_stopping_target_fwd_f32(
                d, K, B, _lib.ptr(t_vec), _lib.ptr(s_vec), _lib.ptr(tau), _lib.ptr(gam), float(T_model), _lib.ptr(N0),
                _lib.ptr(N1), _lib.ptr(dN0), _lib.ptr(dN1), _lib.ptr(ops["q"]), _lib.ptr(ops["v"]), _lib.ptr(ops["gT"]),
                _lib.ptr(target), _lib.stream_ptr(N0.device)), "socmx_socm_stopping_target_fwd_f32")
        ctx.save_for_backward(gam, N0, N1, dN0, dN1, t_vec, s_vec, tau, ops["q"], ops["v"], ops["gT"])
        ctx.meta = (K, float(T_model), gamma.shape, gamma2.shape, gamma3.shape)
        return target

    @staticmethod
    def backward(ctx, gtarget):
        L = _lib.lib()
        gam, N0, N1, dN0, dN1, t_vec, s_vec, tau, q, v, gT = ctx.saved_tensors
        K, T_model, sh, sh2, sh3 = ctx.meta
        B, d = gT.shape
        gtarget = gtarget.detach().to(torch.float32).contiguous()
        gN0, gN1, gdN0, gdN1 = (torch.empty_like(N0) for _ in range(4))
        part = torch.empty(N0.shape[0], 3, dtype=torch.float32, device=N0.device)
        with _lib.on_device(N0.device):
            _lib.check(L.socmx_socm_stopping_target_bwd_f32(
                d, K, B, _lib.ptr(t_vec), _lib.ptr(s_vec), _lib.ptr(tau), _lib.ptr(gam), T_model, _lib.ptr(N0), _lib.ptr(N1),
                _lib.ptr(dN0), _lib.ptr(dN1), _lib.ptr(q), _lib.ptr(v), _lib.ptr(gT), _lib.ptr(gtarget), _lib.ptr(gN0),
                _lib.ptr(gN1), _lib.ptr(gdN0), _lib.ptr(gdN1), _lib.ptr(part), _lib.stream_ptr(N0.device)),
                "socmx_socm_stopping_target_bwd_f32")
        gg = part.sum(0)
        return (gg[0].reshape(sh), gg[1].reshape(sh2), gg[2].reshape(sh3), gN0, gN1, gdN0, gdN1, None, None, None, None, None,
                None)


class _ResidualHip(torch.autograd.Function):
    """objective = inv_norm sum_{i,m} w_m |sigma^T (nabla_V - target)[i,m]|^2 for ANY target (socmx_socm_residual_f32: the
    SOCM loss's residual kernels behind their own entry point); d obj / d nabla_V = G, d obj / d target = -G."""

    @staticmethod
    def forward(ctx, target, nabla_V, weight, pb, K, inv_norm):
        Lh, f = _lib.lib(), _lib.ptr
        Kp, B, d = nabla_V.shape
        dev = nabla_V.device
        c = lambda t: t.detach().to(torch.float32).contiguous()
        tg, nv, w = c(target), c(nabla_V), c(weight)
        G = torch.empty_like(nv)
        obj = torch.zeros(1, dtype=torch.float32, device=dev)
        with _lib.on_device(dev):
            _lib.check(Lh.socmx_socm_residual_f32(pb.c_struct(), K, B, f(tg), f(nv), f(w), float(inv_norm), f(G), f(obj),
                                                  f(_lib.objective_workspace(dev, K, B)), _lib.stream_ptr(dev)),
                       "socmx_socm_residual_f32")
        ctx.save_for_backward(G)
        return obj[0]

    @staticmethod
    def backward(ctx, gout):
        (G,) = ctx.saved_tensors
        g = G * gout
        return (-g if ctx.needs_input_grad[0] else None), (g if ctx.needs_input_grad[1] else None), None, None, None, None


def masked_residual_hip(pb, K, target, nabla_V, weight, stop_indicators):
    """sum_{i,m} w_m |stop[i,m] sigma^T (nabla_V - target)[i,m]|^2 (method.py:692-712, the numerator of the stopping-time
    objective): the 0 / 1 mask commutes with sigma^T, so it is applied to both operands and the plain residual kernel does
    the rest -- instead of a (d, d) library GEMM per direction and ~20 elementwise launches with their autograd."""
    sm = stop_indicators.unsqueeze(2)
    return _ResidualHip.apply(sm * target, sm * nabla_V, weight, pb, K, 1.0)


def stopping_target_hip(gamma, gamma2, gamma3, N0, N1, dN0, dN1, t_vec, s_vec, tau, ops, K, T_model):
    return _StoppingTargetHip.apply(gamma, gamma2, gamma3, N0, N1, dN0, dN1, t_vec, s_vec, tau, ops, K, T_model)


def socm_objective_net(pb, ts, lmbd, K, states, noises, controls, net, dnet, gamma, delta, nablaV, w, inv_norm):
    """SOCM objective from the raw SigmoidMLP outputs (GPU only; raises if libsocmx.so is missing)."""
    ops = socm_operands_hip(pb, ts, lmbd, states, noises, controls)
    return _TargetResidualNetHip.apply(net, dnet, gamma, nablaV, w, delta, ops, pb, K, inv_norm)


def socm_objective(pb, ts, lmbd, K, states, noises, controls, M_all, dM_all, nablaV, w, inv_norm,
                   want_target=False):
    """The SOCM objective from the rollout buffers and the network outputs.
    CUDA tensors: HIP kernels (raises if libsocmx.so is missing).  CPU tensors: plain torch."""
    if M_all.is_cuda:
        ops = socm_operands_hip(pb, ts, lmbd, states, noises, controls)
        return _TargetResidualHip.apply(M_all, dM_all, nablaV, w, ops, pb, K, inv_norm, want_target)
    v, q, gT = socm_operands(pb, ts, lmbd, states, noises, controls)
    obj, tgt = target_residual_torch(pb, K, M_all, dM_all, q, v, gT, nablaV, w, inv_norm)
    return (obj, tgt.detach()) if want_target else obj


# --------------------------------------------------------------------------------------
# importance weights
# --------------------------------------------------------------------------------------

def weights_and_stats(lpd, lps, ltw, scalars=None):
    """w = exp(lpd+lps+ltw) (method.py:258-262) and stats = (sum w, sum (w-mean_local)^2, n[, mean, std]): mean
    and unbiased std (method.py:903-904) follow from the first three, and shards combine with Chan's formula.
    `scalars` (GPU only): (gamma, gam_out, norm, gout_out, obj_zero) one-element tensors or None each -- the same launch copies
    gamma, forms 1 / norm and clears the objective's accumulator (socmx_weights_stats_scalars_f32)."""
    if lpd.is_cuda:
        L = _lib.lib()
        B = lpd.shape[0]
        w = torch.empty_like(lpd)
        stats = torch.empty(5, dtype=torch.float32, device=lpd.device)
        with _lib.on_device(lpd.device):
            if scalars is not None:
                _lib.check(L.socmx_weights_stats_scalars_f32(_lib.ptr(lpd), _lib.ptr(lps), _lib.ptr(ltw), B, _lib.ptr(w),
                                                             _lib.ptr(stats), *[_lib.ptr(x) for x in scalars],
                                                             _lib.stream_ptr(lpd.device)),
                           "socmx_weights_stats_scalars_f32")
            else:
                _lib.check(L.socmx_weights_stats_f32(_lib.ptr(lpd), _lib.ptr(lps), _lib.ptr(ltw), B, _lib.ptr(w),
                                                     _lib.ptr(stats), _lib.stream_ptr(lpd.device)),
                           "socmx_weights_stats_f32")
        return w, stats
    w = torch.exp(lpd + lps + ltw)
    return w, torch.stack([w.sum(), ((w - w.mean()) ** 2).sum(), torch.tensor(float(w.shape[0]))])


def combine_stats(all_stats):
    """(G,3) per-shard stats -> (3,) global stats (Chan et al. parallel variance)."""
    s1, m2, n = all_stats[:, 0], all_stats[:, 1], all_stats[:, 2]
    N = n.sum()
    mean = s1.sum() / N
    M2 = m2.sum() + (n * (s1 / n - mean) ** 2).sum()
    return torch.stack([s1.sum(), M2, N])


def mean_std_from_stats(stats):
    if stats.numel() == 5:          # single shard on the GPU: the kernel already wrote mean and unbiased std
        return stats[3], stats[4]
    return stats[0] / stats[2], torch.sqrt(stats[1] / (stats[2] - 1))
