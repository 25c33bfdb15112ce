"""Networks of the hot path, parameter-compatible with the reference.

* `FullyConnectedUNet`  -- nabla_V(t,x): reference models.py:202-242.
* `SigmoidMLP`          -- M(t,s) with M(t,t)=I: reference models.py:245-275.
* `TwoBoundarySigmoidMLP` -- per-sample M for stopping times: models.py:278-393.

State-dict keys (`down_0.0.weight`, `sigmoid_layers.4.bias`, ...) and the order in
which `nn.Linear` layers are constructed (hence the RNG stream consumed from a
given `torch.manual_seed`) match the reference, so checkpoints and seeds carry
over.  The modules are ordinary torch modules (autograd through hipBLASLt GEMMs);
`FullyConnectedUNet.packed()` additionally exposes the MFMA-fragment image that
the fused rollout kernel streams.
"""
import torch
import torch.nn as nn

from . import _lib

# (name, fan_in, fan_out, relu) in reference construction order (models.py:212-228);
# sizes as functions of (d, h0, h1, h2)
_UNET_SPEC = (
    ("down_0", lambda d, h: (d + 1, h[0]), True),
    ("down_1", lambda d, h: (h[0], h[1]), True),
    ("down_2", lambda d, h: (h[1], h[2]), True),
    ("res_0", lambda d, h: (d + 1, d), False),
    ("res_1", lambda d, h: (h[0], h[0]), False),
    ("res_2", lambda d, h: (h[1], h[1]), False),
    ("up_2", lambda d, h: (h[2], h[1]), True),
    ("up_1", lambda d, h: (h[1], h[0]), True),
    ("up_0", lambda d, h: (h[0], d), True),
)


class _LinearSplitK(torch.autograd.Function):
    """y = x W^T + b with a batched split-K weight gradient.

    On the trajectory rows (R = (K+1)*B ~ 25k) the weight gradient dW = dY^T X is a GEMM with tiny M, N
    (<= 256) and a huge reduction dimension; the library runs it on (M/32)*(N/32) <= 64 workgroups without
    split-K (53-96 us per layer on MI355X).  Splitting R into S slabs as a bmm fills the chip (13-41 us)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        return torch.addmm(bias, x, weight.t())

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gy = gy.contiguous()
        gx = gy @ weight if ctx.needs_input_grad[0] else None
        if gy.is_cuda and gy.dtype == torch.float32 and gy.shape[0] >= 4096 and x.is_contiguous():
            _, gw, gb = _wgrad_bias_fused(gy, x)
            return gx, gw, gb
        gb = _colsum(gy) if ctx.needs_input_grad[2] else None
        return gx, _wgrad_splitk(gy, x), gb


class _LinearReluSplitK(torch.autograd.Function):
    """relu(x W^T + b): ReLU in the GEMM epilogue (hipBLASLt), its backward fused with the bias-gradient reduction
    (socmx_relu_bwd_colsum_f32), split-K weight gradient as in _LinearSplitK."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        y = torch._addmm_activation(bias, x, weight.t(), use_gelu=False)
        ctx.save_for_backward(x, weight, y)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, y = ctx.saved_tensors
        gz, gw, gb = _wgrad_bias_fused(gy.contiguous(), x, y=y)
        gx = gz @ weight if ctx.needs_input_grad[0] else None
        return gx, gw, gb


def _colsum(gy):
    """Bias gradient gy.sum(0) of a tall contiguous fp32 (R, C) matrix: socmx_colsum_f32 on the GPU (one HBM pass at
    several TB/s; the generic reduction kernel runs this shape at ~1.3 TB/s)."""
    if not (gy.is_cuda and gy.dtype == torch.float32 and gy.dim() == 2 and gy.shape[0] >= 4096):
        return gy.sum(0)
    L = _lib.lib()
    R, C = gy.shape
    nblk = L.socmx_colsum_blocks(R, C)
    partial = torch.empty(nblk * C, dtype=torch.float32, device=gy.device)
    out = torch.empty(C, dtype=torch.float32, device=gy.device)
    with _lib.on_device(gy.device):
        _lib.check(L.socmx_colsum_f32(_lib.ptr(gy), R, C, _lib.ptr(partial), _lib.ptr(out), _lib.stream_ptr(gy.device)),
                   "socmx_colsum_f32")
    return out


def _wgrad_splitk(gy, x, S=16):
    """gy^T x for (R, out) x (R, in) with R >> out, in: S row slabs as one bmm (+ a short tail GEMM)."""
    R = x.shape[0]
    main = (R // S) * S
    if main == 0:
        return gy.t() @ x
    gw = torch.bmm(gy[:main].view(S, main // S, -1).transpose(1, 2), x[:main].view(S, main // S, -1)).sum(0)
    if main < R:
        gw = gw + gy[main:].t() @ x[main:]
    return gw


def _wgrad_bias_fused(gz, x, y=None, S=16):
    """(gz' , dW, db) of a Linear(+ReLU) layer over many GPU rows: column-sum partials (with the ReLU mask when `y` is
    given), the split-K bmm, then ONE launch that adds the S weight-gradient slabs and the bias partials."""
    L = _lib.lib()
    R, C = gz.shape
    dev = gz.device
    f32 = dict(dtype=torch.float32, device=dev)
    nblk = L.socmx_colsum_blocks(R, C)
    partial = torch.empty(nblk * C, **f32)
    if y is not None:
        gy, gz = gz, torch.empty_like(gz)
        with _lib.on_device(dev):
            _lib.check(L.socmx_relu_bwd_colsum_f32(_lib.ptr(gy), _lib.ptr(y), R, C, _lib.ptr(gz), _lib.ptr(partial), None,
                                                   _lib.stream_ptr(dev)), "socmx_relu_bwd_colsum_f32")
    else:
        with _lib.on_device(dev):
            _lib.check(L.socmx_colsum_f32(_lib.ptr(gz), R, C, _lib.ptr(partial), None, _lib.stream_ptr(dev)),
                       "socmx_colsum_f32")
    main = (R // S) * S
    parts = torch.bmm(gz[:main].view(S, main // S, -1).transpose(1, 2), x[:main].view(S, main // S, -1))
    tail = (gz[main:].t() @ x[main:]).contiguous() if main < R else None
    gw = torch.empty(C, x.shape[1], **f32)
    gb = torch.empty(C, **f32)
    with _lib.on_device(dev):
        _lib.check(L.socmx_linear_bwd_finish_f32(_lib.ptr(parts), S, gw.numel(), _lib.ptr(tail), _lib.ptr(gw),
                                                 _lib.ptr(partial), nblk, C, _lib.ptr(gb), _lib.stream_ptr(dev)),
                   "socmx_linear_bwd_finish_f32")
    return gz, gw, gb


def _linear(seq, x, splitk):
    lin = seq[0]
    if splitk and len(seq) > 1:
        return _LinearReluSplitK.apply(x, lin.weight, lin.bias)
    y = _LinearSplitK.apply(x, lin.weight, lin.bias) if splitk else lin(x)
    return torch.relu(y) if len(seq) > 1 else y


def _scale_(seq, factor):
    for m in seq:
        if isinstance(m, nn.Linear):
            m.weight.data *= factor
            m.bias.data *= factor


class FullyConnectedUNet(nn.Module):
    def __init__(self, dim=2, hdims=(256, 128, 64), scaling_factor=1.0):
        super().__init__()
        self.dim = int(dim)
        self.hdims = [int(h) for h in hdims]
        for name, dims, relu in _UNET_SPEC:
            fin, fout = dims(self.dim, self.hdims)
            seq = nn.Sequential(nn.Linear(fin, fout), nn.ReLU()) if relu else nn.Sequential(nn.Linear(fin, fout))
            setattr(self, name, seq)
        # the reference scales in the order down_*, res_*, up_0, up_1, up_2 -- order is irrelevant
        for name, _, _ in _UNET_SPEC:
            _scale_(getattr(self, name), scaling_factor)
        self._packed = None

    def forward(self, x):
        # many rows on the GPU with gradients on: same math, split-K weight gradients (see _LinearSplitK)
        sk = (x.is_cuda and x.dtype == torch.float32 and x.shape[0] >= 8192 and torch.is_grad_enabled()
              and x.is_contiguous())
        r1 = _linear(self.down_0, x, sk)
        r2 = _linear(self.down_1, r1, sk)
        r3 = _linear(self.down_2, r2, sk)
        o2 = _linear(self.up_2, r3, sk) + _linear(self.res_2, r2, sk)
        o1 = _linear(self.up_1, o2, sk) + _linear(self.res_1, r1, sk)
        return _linear(self.up_0, o1, sk) + _linear(self.res_0, x, sk)

    # ---- HIP side ---------------------------------------------------------------
    def hip_lib(self):
        """The library handle for the calls that depend on the architecture (rollout, forward, backward): the default
        library or the variant compiled for these hidden widths (socmx._lib.variant)."""
        return _lib.variant(self.hdims)

    def c_struct(self):
        s = _lib.Unet(d=self.dim, hdims=_lib.i3(self.hdims))
        keep = []
        for i, name in enumerate(_lib.UNET_LAYERS):
            lin = getattr(self, name)[0]
            w = lin.weight.detach().contiguous()
            b = lin.bias.detach().contiguous()
            keep += [w, b]
            s.weight[i] = _lib.ptr(w)
            s.bias[i] = _lib.ptr(b)
        return s, keep

    def packed(self):
        """Fragment-ordered image of the CURRENT weights.  Re-packed on every call (one ~3 us kernel per rollout): an
        in-place update cannot be detected reliably -- torch's fused multi-tensor Adam does not bump the parameters'
        `_version`, and a cached image silently kept the rollout on the initial weights."""
        L = _lib.lib()
        dev = next(self.parameters()).device
        n = L.socmx_unet_packed_floats(self.dim, _lib.i3(self.hdims))
        if self._packed is None or self._packed.numel() != n or self._packed.device != dev:
            self._packed = torch.empty(n, dtype=torch.float32, device=dev)
        s, keep = self.c_struct()
        with _lib.on_device(dev):
            _lib.check(L.socmx_unet_pack_f32(s, _lib.ptr(self._packed), _lib.stream_ptr(dev)), "socmx_unet_pack_f32")
        return self._packed

    def packed_bwd(self):
        """Fragment-ordered image of the TRANSPOSED weights of the seven layers the backward chain multiplies by
        (socmx_unet_pack_bwd_f32); rebuilt on every call like `packed()`."""
        L = _lib.lib()
        dev = next(self.parameters()).device
        n = L.socmx_unet_packed_bwd_floats(self.dim, _lib.i3(self.hdims))
        pt = self.__dict__.get("_packed_bwd")
        if pt is None or pt.numel() != n or pt.device != dev:
            pt = self.__dict__["_packed_bwd"] = torch.empty(n, dtype=torch.float32, device=dev)
        s, keep = self.c_struct()
        with _lib.on_device(dev):
            _lib.check(L.socmx_unet_pack_bwd_f32(s, _lib.ptr(pt), _lib.stream_ptr(dev)), "socmx_unet_pack_bwd_f32")
        return pt

    def __getstate__(self):  # keep the solver picklable (reference main.py:466 pickles the module)
        st = self.__dict__.copy()
        st["_packed"] = None
        st.pop("_packed_bwd", None)
        return st


_warned = set()


def warn_library_fallback(what, why):
    """Once per process and per kind: a run on the GPU that leaves the hand-written kernels for torch autograd + library GEMMs
    (still on the GPU, same results, slower) says so -- `socmx_capabilities` lists the ranges the kernels take."""
    if what not in _warned:
        _warned.add(what)
        import warnings
        warnings.warn(f"socmx: {what} runs on torch autograd + library GEMMs for this configuration ({why}); "
                      "see socmx_capabilities() for the ranges the HIP kernels take")


def unet_backward_supported(net, n_rows):
    """True when socmx_unet_backward_f32 takes this architecture (its 16-row tiles must fit 160 KiB of LDS)."""
    L = net.hip_lib()
    ws, ng = _lib.C.c_int64(0), _lib.C.c_int64(0)
    ok = L.socmx_unet_backward_sizes(net.dim, _lib.i3(net.hdims), int(n_rows), _lib.C.byref(ws), _lib.C.byref(ng)) == 0
    if not ok:
        warn_library_fallback("the control-network backward", f"arch.hdims={list(net.hdims)}, d={net.dim}: tiles beyond 160 KiB of LDS")
    return ok


def unet_backward_hip(net, x, ts, rows_per_t, gout, return_flat=False, packed=None, out=None, packed_bwd=None,
                      gout_scale=None, saved=None):
    """d objective / d parameters of `net` from gout = d objective / d net([ts[r // rows_per_t], x[r]]) for the N rows of
    x (N, d): socmx_unet_backward_f32 (forward recomputed in LDS, no library GEMM).  Returns the gradients in
    `net.parameters()` order (views of one flat buffer).  `gout_scale`: a (1,) fp32 device tensor that multiplies gout as
    the kernel reads it (the iteration's d loss / d objective), instead of an elementwise launch in front of the call.
    `saved`: (workspace, records) that the rollout which produced `x` wrote with the SAME weights (rollout.hip_trajectories'
    act_export): socmx_unet_backward_saved_f32 -- no forward re-computation."""
    L = net.hip_lib()
    dev = x.device
    N, d = x.shape
    assert d == net.dim and gout.shape == (N, d)
    x = x.detach().to(torch.float32).contiguous()
    gout = gout.detach().to(torch.float32).contiguous()
    ts = ts.detach().to(device=dev, dtype=torch.float32).contiguous()
    assert ts.numel() * rows_per_t >= N
    ws, ng = _lib.C.c_int64(0), _lib.C.c_int64(0)
    _lib.check(L.socmx_unet_backward_sizes(d, _lib.i3(net.hdims), N, _lib.C.byref(ws), _lib.C.byref(ng)),
               "socmx_unet_backward_sizes")
    if saved is not None:
        work, records = saved
        assert work.dtype == torch.float32 and work.numel() >= ws.value and work.device == dev and work.is_contiguous()
        assert records.dtype == torch.int32 and records.numel() >= N * 32 and records.device == dev and records.is_contiguous()
    else:
        work = torch.empty(ws.value, dtype=torch.float32, device=dev)
    # (`out`: a caller-owned fp32 buffer of >= n_grad floats -- a sharded Trainer appends its scalars behind the gradient
    #  and all-reduces the whole buffer)
    flat = torch.empty(ng.value, dtype=torch.float32, device=dev) if out is None else out[:ng.value]
    with _lib.on_device(dev):
        # (`packed` / `packed_bwd`: the forward / transposed image of the CURRENT weights when the caller knows they are
        #  fresh -- the one this iteration's rollout just used, the one packed beside it on the second stream -- instead of
        #  re-packing in front of the backward)
        if gout_scale is not None:
            assert gout_scale.dtype == torch.float32 and gout_scale.numel() == 1 and gout_scale.device == x.device
        pk = _lib.ptr(packed if packed is not None else net.packed())
        pkT = _lib.ptr(packed_bwd if packed_bwd is not None else net.packed_bwd())
        if saved is not None:
            _lib.check(L.socmx_unet_backward_saved_f32(pk, pkT, d, _lib.i3(net.hdims), _lib.ptr(x), _lib.ptr(ts), int(rows_per_t), N,
                                                       _lib.ptr(gout), _lib.ptr(gout_scale), records.data_ptr(), _lib.ptr(work),
                                                       _lib.ptr(flat), _lib.stream_ptr(dev)), "socmx_unet_backward_saved_f32")
        else:
            _lib.check(L.socmx_unet_backward_scaled_f32(pk, pkT, d, _lib.i3(net.hdims), _lib.ptr(x), _lib.ptr(ts), int(rows_per_t), N,
                                                        _lib.ptr(gout), _lib.ptr(gout_scale), _lib.ptr(work), _lib.ptr(flat),
                                                        _lib.stream_ptr(dev)), "socmx_unet_backward_scaled_f32")
    # flat is in SOCMX_L_* order (weight, bias per layer); parameters() follows the module construction order = the same
    grads, off = {}, 0
    for name in _lib.UNET_LAYERS:
        lin = getattr(net, name)[0]
        nw, nb = lin.weight.numel(), lin.bias.numel()
        grads[name] = (flat[off:off + nw].view_as(lin.weight), flat[off + nw:off + nw + nb])
        off += nw + nb
    out = []
    for name, _, _ in _UNET_SPEC:                  # parameters() order: construction order of the Sequentials
        out += list(grads[name])
    return (out, flat) if return_flat else out


class UnetOnTrajectory(torch.autograd.Function):
    """nabla_V on the trajectory rows (method.py:272-278) as a function of the network parameters, with the VALUES
    supplied by the rollout kernel (it evaluates the network at every grid point anyway: socmx_rollout_ex_f32's
    `nabla_v`) and the parameter gradients by socmx_unet_backward_f32.  No forward pass, no saved activations."""

    @staticmethod
    def forward(ctx, values, states, ts, net, saved, *params):
        ctx.net = net
        ctx.saved = saved          # (workspace, records) the rollout wrote with these weights, or None: re-compute (unet_backward_hip)
        ctx.save_for_backward(states, ts)
        return values.view_as(values)

    @staticmethod
    def backward(ctx, gout):
        states, ts = ctx.saved_tensors
        Kp, B, d = states.shape
        saved, ctx.saved = ctx.saved, None          # (the backward consumes the workspace: a second backward re-computes)
        grads = unet_backward_hip(ctx.net, states.reshape(Kp * B, d), ts, B, gout.reshape(Kp * B, d), saved=saved)
        return (None, None, None, None, None) + tuple(grads)


def unet_on_trajectory(net, values, states, ts, saved=None):
    return UnetOnTrajectory.apply(values, states, ts, net, saved, *net.parameters())


def unet_forward_hip(net, tx):
    """nabla_V rows through the fused MFMA kernel (no autograd): socmx_unet_forward_f32."""
    L = net.hip_lib()
    tx = tx.detach().to(torch.float32).contiguous()
    out = torch.empty(tx.shape[0], net.dim, dtype=torch.float32, device=tx.device)
    with _lib.on_device(tx.device):
        _lib.check(L.socmx_unet_forward_f32(_lib.ptr(net.packed()), net.dim, _lib.i3(net.hdims), _lib.ptr(tx),
                                            tx.shape[0], _lib.ptr(out), _lib.stream_ptr(tx.device)),
                   "socmx_unet_forward_f32")
    return out


def pair_net_forward(d, hdims, params, t, s, packed=None, z=None, out=None):
    """(net, dnet, packed): socmx_mnet_pack_f32 + socmx_mnet_forward_f32 on plain fp32 tensors; `packed` may be a
    caller-owned buffer (hipGraph mode keeps the image for the deferred backward).  `z` (Np,): the third network input of
    TwoBoundarySigmoidMLP (params[0] is then (h0, 3))."""
    L = _lib.lib()
    dev = t.device
    h2 = _lib.i2(hdims)
    if packed is None:
        packed = torch.empty(L.socmx_mnet_packed_floats(d, h2), dtype=torch.float32, device=dev)
    Np = t.shape[0]
    if out is not None:                       # caller-owned (net, dnet) buffers
        net, dnet = out
    else:
        net = torch.empty(Np, d, d, dtype=torch.float32, device=dev)
        dnet = torch.empty(Np, d, d, dtype=torch.float32, device=dev)
    n_in = int(params[0].shape[1])
    assert (n_in == 3) == (z is not None)
    with _lib.on_device(dev):
        _lib.check(L.socmx_mnet_pack_f32(d, h2, n_in, *[_lib.ptr(p) for p in params], _lib.ptr(packed),
                                         _lib.stream_ptr(dev)), "socmx_mnet_pack_f32")
        _lib.check(L.socmx_mnet_forward_f32(_lib.ptr(packed), d, h2, _lib.ptr(t), _lib.ptr(s), _lib.ptr(z), Np, _lib.ptr(net),
                                            _lib.ptr(dnet), _lib.stream_ptr(dev)), "socmx_mnet_forward_f32")
    return net, dnet, packed


def pair_net_backward(d, hdims, shapes, packed, t, s, g_net, g_dnet, out=None, z=None):
    """Parameter gradients [W0, b0, W1, b1, W2, b2] (views of one flat buffer -- `out` when given): socmx_mnet_backward_f32."""
    L = _lib.lib()
    dev = t.device
    Np = t.shape[0]
    h2 = _lib.i2(hdims)
    n_in = int(shapes[0][1])
    ws, ng = _lib.C.c_int64(0), _lib.C.c_int64(0)
    _lib.check(L.socmx_mnet_backward_sizes(d, h2, n_in, Np, _lib.C.byref(ws), _lib.C.byref(ng)), "socmx_mnet_backward_sizes")
    work = torch.empty(ws.value, dtype=torch.float32, device=dev)
    flat = torch.empty(ng.value, dtype=torch.float32, device=dev) if out is None else out[:ng.value]
    with _lib.on_device(dev):
        _lib.check(L.socmx_mnet_backward_f32(_lib.ptr(packed), d, h2, n_in, _lib.ptr(t), _lib.ptr(s), _lib.ptr(z), Np,
                                             _lib.ptr(g_net), _lib.ptr(g_dnet), _lib.ptr(work), _lib.ptr(flat),
                                             _lib.stream_ptr(dev)),
                   "socmx_mnet_backward_f32")
    grads, off = [], 0
    for shp in shapes:
        n = int(torch.Size(shp).numel())
        grads.append(flat[off:off + n].view(shp))
        off += n
    return grads


class _PairNetHip(torch.autograd.Function):
    """(net, d net / d s) of SigmoidMLP.sigmoid_layers (or, with a third input z, TwoBoundarySigmoidMLP.sigmoid_layers) on the
    pair grid: socmx_mnet_forward_f32 / socmx_mnet_backward_f32 (value and forward tangent share every weight fragment; the
    backward recomputes the forward in LDS)."""

    @staticmethod
    def forward(ctx, t, s, z, d, hdims, w0, b0, w1, b1, w2, b2):
        c = lambda x: x.detach().to(torch.float32).contiguous()
        t, s = c(t), c(s)
        z = None if z is None else c(z)
        params = [c(p) for p in (w0, b0, w1, b1, w2, b2)]
        net, dnet, packed = pair_net_forward(d, hdims, params, t, s, z=z)
        ctx.save_for_backward(packed, t, s, *([] if z is None else [z]))
        ctx.meta = (d, tuple(hdims), [p.shape for p in params])
        return net, dnet

    @staticmethod
    def backward(ctx, g_net, g_dnet):
        packed, t, s, *rest = ctx.saved_tensors
        d, hdims, shapes = ctx.meta
        c = lambda x: x.detach().to(torch.float32).contiguous()
        grads = pair_net_backward(d, hdims, shapes, packed, t, s, c(g_net), c(g_dnet), z=rest[0] if rest else None)
        return (None, None, None, None, None) + tuple(grads)


def pair_net_supported(mlp, n_pairs):
    L = _lib.lib()
    ws, ng = _lib.C.c_int64(0), _lib.C.c_int64(0)
    n_in = int(mlp.sigmoid_layers[0].weight.shape[1])
    ok = L.socmx_mnet_backward_sizes(mlp.dim, _lib.i2(mlp.hdims), n_in, int(n_pairs), _lib.C.byref(ws),
                                     _lib.C.byref(ng)) == 0
    if not ok:
        warn_library_fallback("the pair-grid network (M)", f"d={mlp.dim}, arch.hdims_M={list(mlp.hdims)}")
    return ok


class SigmoidMLP(nn.Module):
    def __init__(self, dim=10, hdims=(128, 128), gamma=3.0, scaling_factor=1.0):
        super().__init__()
        self.dim = int(dim)
        self.hdims = [int(h) for h in hdims]
        self.gamma = gamma  # shared nn.Parameter owned by the NeuralSDE (method.py:134)
        self.sigmoid_layers = nn.Sequential(
            nn.Linear(2, self.hdims[0]), nn.ReLU(),
            nn.Linear(self.hdims[0], self.hdims[1]), nn.ReLU(),
            nn.Linear(self.hdims[1], self.dim * self.dim),
        )
        self.scaling_factor = scaling_factor
        _scale_(self.sigmoid_layers, scaling_factor)

    def net(self, t, s):
        return self.sigmoid_layers(torch.stack((t, s), dim=1)).reshape(-1, self.dim, self.dim)

    def forward(self, t, s):
        decay = torch.exp(-self.gamma * (s - t)).reshape(-1, 1, 1)
        eye = torch.eye(self.dim, device=t.device, dtype=t.dtype)
        return decay * eye + (1.0 - decay) * self.net(t, s)

    def forward_with_ds(self, t, s, raw=False):
        """(M, dM/ds) with the s-derivative as an analytic forward tangent
        (the reference differentiates with functorch.jacrev: method.py:510-515).
        raw=True returns (net, d net/ds) instead: the exp(-gamma (s-t)) blend is then fused into the HIP
        contraction (loss.socm_objective_net)."""
        l0, l2, l4 = self.sigmoid_layers[0], self.sigmoid_layers[2], self.sigmoid_layers[4]
        if (raw and t.is_cuda and t.dtype == torch.float32 and getattr(self, "fused_pair_net", True)
                and pair_net_supported(self, t.shape[0])):
            # hand-written tile kernels (csrc/socmx_unet_bwd.hip, K3): no library GEMM, value + tangent in one pass
            return _PairNetHip.apply(t, s, None, self.dim, tuple(self.hdims), l0.weight, l0.bias, l2.weight, l2.bias,
                                     l4.weight, l4.bias)
        sk = t.is_cuda and t.shape[0] >= 8192 and torch.is_grad_enabled()
        lin = (lambda x, w, b: _LinearSplitK.apply(x, w, b)) if sk else (lambda x, w, b: torch.addmm(b, x, w.T))
        if sk:      # fused Linear+ReLU (the ReLU mask of the tangent path is h > 0, same set as a > 0)
            h1 = _LinearReluSplitK.apply(torch.stack((t, s), dim=1), l0.weight, l0.bias)
            h2 = _LinearReluSplitK.apply(h1, l2.weight, l2.bias)
        else:
            h1 = torch.relu(lin(torch.stack((t, s), dim=1), l0.weight, l0.bias))
            h2 = torch.relu(lin(h1, l2.weight, l2.bias))
        net = lin(h2, l4.weight, l4.bias).reshape(-1, self.dim, self.dim)
        zero2, zero4 = torch.zeros_like(l2.bias), torch.zeros_like(l4.bias)   # tangent path: no bias
        t1 = (h1 > 0).to(h1.dtype) * l0.weight[:, 1]
        t2 = (h2 > 0).to(h2.dtype) * lin(t1, l2.weight, zero2)
        dnet = lin(t2, l4.weight, zero4).reshape(-1, self.dim, self.dim)
        if raw:
            return net, dnet
        decay = torch.exp(-self.gamma * (s - t)).reshape(-1, 1, 1)
        eye = torch.eye(self.dim, device=t.device, dtype=t.dtype)
        M = decay * eye + (1.0 - decay) * net
        dM = self.gamma * decay * (net - eye) + (1.0 - decay) * dnet
        return M, dM


class TwoBoundarySigmoidMLP(nn.Module):
    """Per-sample M(t, s; tau) for the stopping-time SOCM loss (molecular_dynamics)."""

    def __init__(self, dim=10, hdims=(128, 128), gamma=3.0, gamma2=3.0, gamma3=3.0, scaling_factor=1.0, T=1.0):
        super().__init__()
        self.dim = int(dim)
        self.gamma, self.gamma2, self.gamma3 = gamma, gamma2, gamma3
        self.T = T
        self.sigmoid_layers = nn.Sequential(
            nn.Linear(3, hdims[0]), nn.ReLU(),
            nn.Linear(hdims[0], hdims[1]), nn.ReLU(),
            nn.Linear(hdims[1], self.dim * self.dim),
        )
        self.scaling_factor = scaling_factor
        _scale_(self.sigmoid_layers, scaling_factor)

    def gates(self, t, s, tau):
        """The scalar gate fields of models.py:341-392, each (N, B):  M = w I + c0 net(t,s,0) + c1 net(t,s,1)."""
        st = (s - t).unsqueeze(1)
        ratio = (1 - torch.exp(-self.gamma * st)) / (1 - torch.exp(-self.gamma * torch.abs(tau - t.unsqueeze(1))) + 1e-7)
        factor1 = torch.nan_to_num(1 - torch.minimum(ratio, torch.ones(1, device=t.device)), nan=0.0)
        factor1 = factor1 * (tau - 1e-3 > s.unsqueeze(1)).to(torch.int)
        running = (tau > self.T - 1e-3).to(torch.int)
        e3 = torch.exp(-self.gamma3 * st)
        g2 = lambda x: (1 - torch.exp(-self.gamma2 * x)) * (torch.exp(-self.gamma2 * x) - torch.exp(-self.gamma2))
        return (1 - running) * factor1 + running * e3, (1 - running) * g2(factor1), running * (1 - e3)

    def nets_with_ds(self, t, s):
        """net(t,s,0), net(t,s,1) (N,d,d) and their s-derivatives as analytic forward tangents."""
        l0, l2, l4 = self.sigmoid_layers[0], self.sigmoid_layers[2], self.sigmoid_layers[4]
        d = self.dim
        out = []
        for flag in (0.0, 1.0):
            x = torch.stack((t, s, torch.full_like(s, flag)), 1)
            h1 = torch.relu(torch.addmm(l0.bias, x, l0.weight.T))
            h2 = torch.relu(torch.addmm(l2.bias, h1, l2.weight.T))
            net = torch.addmm(l4.bias, h2, l4.weight.T).reshape(-1, d, d)
            t1 = (h1 > 0).to(h1.dtype) * l0.weight[:, 1]
            t2 = (h2 > 0).to(h2.dtype) * (t1 @ l2.weight.T)
            out.append((net, (t2 @ l4.weight.T).reshape(-1, d, d)))
        return out[0][0], out[1][0], out[0][1], out[1][1]

    def nets_with_ds_hip(self, t, s, grid=None):
        """The same four (N,d,d) tensors from ONE launch of the pair-grid-network kernel on 2 N rows (third input 0 on the
        first N, 1 on the rest): socmx_mnet_forward_f32 / socmx_mnet_backward_f32 with n_in = 3.  `grid`: a dict the
        caller keeps per time grid (the doubled inputs are built once)."""
        N = t.shape[0]
        key = (t.data_ptr(), s.data_ptr(), N)
        if grid is None or grid.get("key") != key:
            c = lambda x: x.detach().to(torch.float32).contiguous()
            built = dict(key=key, t2=torch.cat([c(t), c(t)]), s2=torch.cat([c(s), c(s)]),
                         z=torch.cat([torch.zeros_like(t, dtype=torch.float32), torch.ones_like(t, dtype=torch.float32)]))
            if grid is not None:
                grid.clear()
                grid.update(built)
            else:
                grid = built
        l0, l2, l4 = self.sigmoid_layers[0], self.sigmoid_layers[2], self.sigmoid_layers[4]
        hd = (l0.weight.shape[0], l2.weight.shape[0])
        net, dnet = _PairNetHip.apply(grid["t2"], grid["s2"], grid["z"], self.dim, hd, l0.weight, l0.bias, l2.weight, l2.bias,
                                      l4.weight, l4.bias)
        return net[:N], net[N:], dnet[:N], dnet[N:]

    def hip_supported(self, n_pairs):
        l0, l2 = self.sigmoid_layers[0], self.sigmoid_layers[2]
        L = _lib.lib()
        ws, ng = _lib.C.c_int64(0), _lib.C.c_int64(0)
        ok = L.socmx_mnet_backward_sizes(self.dim, _lib.i2((l0.weight.shape[0], l2.weight.shape[0])), 3, 2 * int(n_pairs),
                                         _lib.C.byref(ws), _lib.C.byref(ng)) == 0
        if not ok:
            warn_library_fallback("the stopping-time pair-grid network", f"d={self.dim}, arch.hdims_M="
                                  f"{[l0.weight.shape[0], l2.weight.shape[0]]}")
        return ok

    def forward(self, t, s, stopping_timestep_values):
        tau = stopping_timestep_values                       # (N, B)
        d = self.dim
        flag = lambda v: torch.full_like(s, v)
        net0 = self.sigmoid_layers(torch.stack((t, s, flag(0.0)), 1)).reshape(-1, 1, d, d)  # stopped
        net1 = self.sigmoid_layers(torch.stack((t, s, flag(1.0)), 1)).reshape(-1, 1, d, d)  # running
        eye = torch.eye(d, device=t.device).reshape(1, 1, d, d)
        st = (s - t).unsqueeze(1)
        ratio = (1 - torch.exp(-self.gamma * st)) / (1 - torch.exp(-self.gamma * torch.abs(tau - t.unsqueeze(1))) + 1e-7)
        factor1 = torch.nan_to_num(1 - torch.minimum(ratio, torch.ones(1, device=t.device)), nan=0.0)
        factor1 = factor1 * (tau - 1e-3 > s.unsqueeze(1)).to(torch.int)
        running = (tau > self.T - 1e-3).to(torch.int)
        e3 = torch.exp(-self.gamma3 * st)
        g2 = lambda x: (1 - torch.exp(-self.gamma2 * x)) * (torch.exp(-self.gamma2 * x) - torch.exp(-self.gamma2))
        w_eye = (1 - running) * factor1 + running * e3
        out = w_eye.unsqueeze(2).unsqueeze(3) * eye
        out = out + ((1 - running) * g2(factor1)).unsqueeze(2).unsqueeze(3) * net0
        out = out + (running * (1 - e3)).unsqueeze(2).unsqueeze(3) * net1
        return out
