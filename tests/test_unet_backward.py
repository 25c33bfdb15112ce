"""socmx_unet_backward_f32 (csrc/socmx_unet_bwd.hip): parameter gradients of the control network over many rows, against
torch autograd through the library-GEMM network of the same weights (fp64 on the device), on the reference-generated
weights of the fixtures.  Covers the three constexpr instantiations (d <= 15, 16..31, 64 with the default widths), the
descriptor-driven kernel (tiny widths: split-K stages) and ragged row counts."""
import os

import numpy as np
import pytest
import torch

from test_host_cpu import build_sde

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _reference_grads(net, tx, gout):
    import copy
    net64 = copy.deepcopy(net).double()
    out = net64(tx.double())
    out.backward(gout.double())
    return [p.grad for p in net64.parameters()]


@pytest.mark.parametrize("name,N,rows_per_t", [
    ("tiny_double_well_d10", 1, 1), ("tiny_double_well_d10", 37, 1), ("tiny_ou_linear_d6", 1000, 8),
    ("tiny_ou_linear_d20", 200, 8), ("tiny_ou_linear_d64", 100, 4),
    ("cfg3_double_well_d10_K200", 16, 1), ("cfg3_double_well_d10_K200", 333, 7), ("cfg3_double_well_d10_K200", 25728, 128),
    ("cfg1_ou_quadratic_easy_d2_K50", 6528, 128),
    ("ouq20_ou_quadratic_easy_d20_K12", 1664, 128), ("cfg5_ou_linear_d64_K20", 2100, 100),
])
def test_unet_backward_kernel_vs_autograd(name, N, rows_per_t):
    from socmx import nets
    sde, aux = build_sde(name, DEV)
    net, d = sde.nabla_V, aux["d"]
    assert nets.unet_backward_supported(net, N)
    g = torch.Generator().manual_seed(N)
    nt = (N + rows_per_t - 1) // rows_per_t
    ts = torch.linspace(0, 1, nt).to(DEV)
    x = (0.7 * torch.randn(N, d, generator=g)).to(DEV)
    gout = torch.randn(N, d, generator=g).to(DEV)
    got = nets.unet_backward_hip(net, x, ts, rows_per_t, gout)
    again = nets.unet_backward_hip(net, x, ts, rows_per_t, gout)
    for a, b in zip(got, again):
        assert torch.equal(a, b)                      # fixed summation order
    tcol = ts[torch.arange(N, device=DEV) // rows_per_t].reshape(-1, 1)
    want = _reference_grads(net, torch.cat([tcol, x], 1), gout)
    assert len(got) == len(want) == 18
    num = den = 0.0
    for (k, p), a, b in zip(net.named_parameters(), got, want):
        assert a.shape == p.shape, k
        e = float(((a.double() - b) ** 2).sum()) ** 0.5
        n_ = float((b ** 2).sum()) ** 0.5
        # fp32 recompute vs an fp64 reference: a pre-activation within rounding of zero flips its ReLU mask and with it
        # that unit's whole contribution of the row.  With 25,728 rows x 842 units ~40 flips are expected, and against the
        # INCOHERENT sums of this test's random gout (norm ~ sqrt(N)) each is worth ~4e-4 of a layer's gradient norm.
        # (The trained loss's gradients are coherent; the end-to-end fixtures hold 1e-3 there.)
        tol = 2e-5 if N <= 1000 else 5e-3
        assert e <= tol * n_ + 1e-6 * max(1.0, float(b.abs().max())), (k, e, n_, N)
        num += e * e
        den += n_ * n_
    assert (num / den) ** 0.5 < (1e-5 if N <= 1000 else 2e-3), (num / den) ** 0.5
    if N >= 1000:
        # additivity over rows (exact up to fp32 summation order): the whole batch = its two halves, each of which takes
        # another tile / slab decomposition -- pins the large-N bookkeeping independently of the ReLU-flip noise above
        h = (N // 2 // rows_per_t) * rows_per_t
        lo = nets.unet_backward_hip(net, x[:h], ts, rows_per_t, gout[:h])
        hi = nets.unet_backward_hip(net, x[h:], ts[h // rows_per_t:], rows_per_t, gout[h:])
        for (k, p), a, b1, b2 in zip(net.named_parameters(), got, lo, hi):
            ref = b1.double() + b2.double()
            e = float(((a.double() - ref) ** 2).sum()) ** 0.5
            assert e <= 3e-6 * float((ref ** 2).sum()) ** 0.5 + 1e-7, (k, e)


def test_unet_backward_zero_gradient_rows_contribute_nothing():
    """Rows whose gout is zero (and the padding rows of the last tile) leave every gradient untouched."""
    from socmx import nets
    sde, aux = build_sde("cfg3_double_well_d10_K200", DEV)
    net, d = sde.nabla_V, aux["d"]
    g = torch.Generator().manual_seed(3)
    x = torch.randn(50, d, generator=g).to(DEV)
    gout = torch.randn(50, d, generator=g).to(DEV)
    ts = torch.linspace(0, 1, 50).to(DEV)
    base = nets.unet_backward_hip(net, x[:21], ts[:21], 1, gout[:21])
    gz = gout.clone()
    gz[21:] = 0
    more = nets.unet_backward_hip(net, x, ts, 1, gz)
    for a, b in zip(base, more):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("name", ["cfg3_double_well_d10_K200", "tiny_ou_linear_d6", "ouq20_ou_quadratic_easy_d20_K12",
                                  "cfg5_ou_linear_d64_K20"])
def test_unet_backward_scaled_equals_the_premultiplied_gradient(name):
    """socmx_unet_backward_scaled_f32: the device scalar multiplies gout as the tiles are read -- the same fp32 products an
    elementwise launch in front of socmx_unet_backward_f32 writes, hence the same gradients bit for bit (constexpr and
    descriptor-driven instantiations, ragged last tile)."""
    from socmx import nets
    sde, aux = build_sde(name, DEV)
    net, d = sde.nabla_V, aux["d"]
    g = torch.Generator().manual_seed(11)
    N = 16 * 9 + 5
    x = torch.randn(N, d, generator=g).to(DEV)
    gout = torch.randn(N, d, generator=g).to(DEV)
    ts = torch.linspace(0, 1, N).to(DEV)
    scale = torch.tensor([0.3718], device=DEV)
    a = nets.unet_backward_hip(net, x, ts, 1, gout * scale)
    b = nets.unet_backward_hip(net, x, ts, 1, gout, gout_scale=scale)
    for u, v in zip(a, b):
        assert torch.equal(u, v)
    assert any(float(u.abs().max()) > 0 for u in a)


@pytest.mark.parametrize("name,extra", [("tiny_double_well_d10", {}), ("tiny_ou_linear_d6", {}), ("tiny_ou_linear_d20", {}),
                                        ("cfg1_ou_quadratic_easy_d2_K50", {}), ("cfg3_double_well_d10_K200", {}),
                                        ("ouq20_ou_quadratic_easy_d20_K12", {}),
                                        # d = 64: the WIDE form (4096 outputs: last layer straight to HBM, split-K backward
                                        # from the row-major gradients, its own weight-gradient kernel) -- 231 pairs / 10 pairs
                                        ("cfg5_ou_linear_d64_K20", {}), ("cfg5_ou_linear_d64_B256_K3", {}),
                                        # ... with rows of 900 floats (d = 30)
                                        ("oul30_ou_linear_d30_K10_B16", {})])
def test_pair_network_kernels_vs_library_autograd(name, extra):
    """K3: socmx_mnet_forward_f32 / _backward_f32 (SigmoidMLP.sigmoid_layers and its s-tangent on the pair grid) against
    the library path of the same module (torch GEMMs + analytic tangent, itself pinned by the reference's jacrev fixtures):
    net, dnet and the six parameter gradients for random upstream gradients."""
    from socmx import loss as L
    sde, aux = build_sde(name, DEV)
    M = sde.M
    K, d = aux["K"], aux["d"]
    ts = aux["ts"]
    t_vec, s_vec, _, _ = L.pair_times(ts, aux["T"], K)
    Np = t_vec.shape[0]
    g = torch.Generator().manual_seed(1)
    gn = torch.randn(Np, d, d, generator=g).to(DEV)
    gd = torch.randn(Np, d, d, generator=g).to(DEV)
    res = []
    for fused in (True, False):
        M.fused_pair_net = fused
        for p in M.parameters():
            p.grad = None
        net, dnet = M.forward_with_ds(t_vec, s_vec, raw=True)
        assert (type(net.grad_fn).__name__ == "_PairNetHipBackward") == fused
        torch.autograd.backward([net, dnet], [gn, gd])
        res.append((net.detach().double(), dnet.detach().double(), [p.grad.double().clone() for p in M.sigmoid_layers.parameters()]))
    (n1, d1, g1), (n0, d0, g0) = res
    sc = lambda x: max(1.0, float(x.abs().max()))
    np.testing.assert_allclose(n1.cpu().numpy(), n0.cpu().numpy(), rtol=1e-5, atol=2e-6 * sc(n0))
    np.testing.assert_allclose(d1.cpu().numpy(), d0.cpu().numpy(), rtol=1e-5, atol=2e-6 * sc(d0))
    for (k, _), a, b in zip(M.sigmoid_layers.named_parameters(), g1, g0):
        e = float(((a - b) ** 2).sum()) ** 0.5
        n_ = float((b ** 2).sum()) ** 0.5
        assert e <= 2e-4 * n_ + 1e-6 * sc(b), (k, e, n_)


@pytest.mark.parametrize("d,hdims,K", [(28, (128, 128), 9), (32, (128, 128), 30), (36, (64, 32), 12), (64, (128, 128), 45),
                                       (64, (16, 16), 7), (40, (128, 64), 20), (48, (48, 80), 17),
                                       # d % 4 != 0: rows of d*d floats that end inside a 16-wide block (d even) or inside a
                                       # 16-byte piece at any 4-byte address (d odd)
                                       (30, (128, 128), 9), (26, (128, 128), 14), (23, (128, 128), 8), (31, (128, 128), 6),
                                       (65, (128, 128), 5), (27, (64, 32), 7), (45, (128, 64), 6), (33, (16, 16), 11),
                                       # last hidden layer beyond 128 units (arch.hdims_M = [256, 256] at d >= 26)
                                       (28, (256, 256), 6), (30, (256, 256), 7), (64, (256, 256), 5), (33, (200, 144), 6)])
def test_wide_pair_network_kernels_vs_fp64(d, hdims, K):
    """The WIDE pair-grid-network kernels (d*d outputs beyond an LDS tile; any d) on random modules of several hidden
    widths and ragged pair counts, against the same module evaluated in fp64 by torch (values, s-tangents and all six
    parameter gradients for random upstream gradients), and additivity of the gradients over the pairs."""
    from socmx import loss as L, nets
    torch.manual_seed(d * 100 + K)
    M = nets.SigmoidMLP(dim=d, hdims=hdims, gamma=torch.nn.Parameter(torch.tensor([1.0])), scaling_factor=0.5).to(DEV)
    ts = torch.linspace(0, 1, K + 1).to(DEV)
    t_vec, s_vec, _, _ = L.pair_times(ts, 1.0, K)
    Np = t_vec.shape[0]
    assert nets.pair_net_supported(M, Np)
    g = torch.Generator().manual_seed(2)
    gn = torch.randn(Np, d, d, generator=g).to(DEV)
    gd = torch.randn(Np, d, d, generator=g).to(DEV)
    net, dnet = M.forward_with_ds(t_vec, s_vec, raw=True)
    assert type(net.grad_fn).__name__ == "_PairNetHipBackward"
    torch.autograd.backward([net, dnet], [gn, gd])
    got = [p.grad.double().cpu().clone() for p in M.sigmoid_layers.parameters()]
    # fp64 reference of the same module on the CPU
    import copy
    M64 = copy.deepcopy(M).double().cpu()
    M64.fused_pair_net = False
    for p_ in M64.parameters():
        p_.grad = None
    n64, d64 = M64.forward_with_ds(t_vec.double().cpu(), s_vec.double().cpu(), raw=True)
    torch.autograd.backward([n64, d64], [gn.double().cpu(), gd.double().cpu()])
    sc = lambda x: max(1.0, float(x.abs().max()))
    np.testing.assert_allclose(net.detach().cpu().numpy(), n64.detach().numpy(), rtol=1e-5, atol=2e-6 * sc(n64))
    np.testing.assert_allclose(dnet.detach().cpu().numpy(), d64.detach().numpy(), rtol=1e-5, atol=2e-6 * sc(d64))
    for (k, p_), a in zip(M64.sigmoid_layers.named_parameters(), got):
        b = p_.grad
        e = float(((a - b) ** 2).sum()) ** 0.5
        n_ = float((b ** 2).sum()) ** 0.5
        assert e <= 1e-5 * n_ + 1e-6 * sc(b), (k, e, n_)
    # additivity over the pairs: the first h pairs + the rest (other tile / slab decomposition)
    h = Np // 3
    c = lambda x: x.detach().to(torch.float32).contiguous()
    params = [c(p_) for l in (0, 2, 4) for p_ in (M.sigmoid_layers[l].weight, M.sigmoid_layers[l].bias)]
    parts = []
    for sl in (slice(0, h), slice(h, Np)):
        _, _, packed = nets.pair_net_forward(d, hdims, params, c(t_vec[sl]), c(s_vec[sl]))
        parts.append(nets.pair_net_backward(d, hdims, [p_.shape for p_ in params], packed, c(t_vec[sl]), c(s_vec[sl]),
                                            c(gn[sl]), c(gd[sl])))
    for a, b1, b2 in zip(got, parts[0], parts[1]):
        ref = b1.double().cpu() + b2.double().cpu()
        e = float(((a - ref) ** 2).sum()) ** 0.5
        assert e <= 1e-5 * float((ref ** 2).sum()) ** 0.5 + 1e-6, e
