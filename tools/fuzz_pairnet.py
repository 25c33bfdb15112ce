#!/usr/bin/env python3
"""Random-shape fuzz of the pair-grid-network kernels (csrc/socmx_unet_bwd.hip, K3) against the same module in fp64 on the CPU:
python3 tools/fuzz_pairnet.py [draws] [seed].  Shapes: d in 1..70, hidden widths 1..256, 1..40 time steps (ragged pair counts);
values, s-tangents and the six parameter gradients for random upstream gradients."""
import sys, os, copy, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "soc-matching_amd"), os.path.join(ROOT, "tests")]
import torch
from socmx import loss as L, nets

DEV = "cuda:0"
draws = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst, bad, skipped = 0.0, 0, 0
for it in range(draws):
    d = rng.choice([rng.randint(1, 70), rng.randint(23, 45), rng.choice([23, 25, 26, 27, 29, 30, 31, 33, 34, 35, 63, 65, 66])])
    hd = (rng.choice([rng.randint(1, 256), 128, 256, 64]), rng.choice([rng.randint(1, 256), 128, 256, 144, 16]))
    K = rng.randint(1, 40 if d < 40 else 12)
    torch.manual_seed(it)
    M = nets.SigmoidMLP(dim=d, hdims=hd, gamma=torch.nn.Parameter(torch.tensor([1.0])), scaling_factor=0.5).to(DEV)
    ts = torch.linspace(0, 1, K + 1).to(DEV)
    t_vec, s_vec, _, _ = L.pair_times(ts, 1.0, K)
    Np = t_vec.shape[0]
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ok = nets.pair_net_supported(M, Np)
    if not ok:
        skipped += 1
        continue
    g = torch.Generator().manual_seed(it)
    gn = torch.randn(Np, d, d, generator=g).to(DEV)
    gd = torch.randn(Np, d, d, generator=g).to(DEV)
    net, dnet = M.forward_with_ds(t_vec, s_vec, raw=True)
    assert type(net.grad_fn).__name__ == "_PairNetHipBackward"
    torch.autograd.backward([net, dnet], [gn, gd])
    got = [p.grad.double().cpu().clone() for p in M.sigmoid_layers.parameters()]
    M64 = copy.deepcopy(M).double().cpu()
    M64.fused_pair_net = False
    for p_ in M64.parameters():
        p_.grad = None
    n64, d64 = M64.forward_with_ds(t_vec.double().cpu(), s_vec.double().cpu(), raw=True)
    torch.autograd.backward([n64, d64], [gn.double().cpu(), gd.double().cpu()])
    errs = [float((net.detach().cpu().double() - n64.detach()).abs().max()) / max(1.0, float(n64.abs().max())),
            float((dnet.detach().cpu().double() - d64.detach()).abs().max()) / max(1.0, float(d64.abs().max()))]
    for p_, a in zip(M64.sigmoid_layers.parameters(), got):
        b = p_.grad
        errs.append(float(((a - b) ** 2).sum()) ** 0.5 / max(float((b ** 2).sum()) ** 0.5, 1e-30))
    e = max(errs)
    worst = max(worst, e)
    flag = "" if e < 2e-5 else "   <-- FAIL"
    bad += e >= 2e-5
    print(f"d={d:3d} hdims_M={hd!s:11s} K={K:3d} Np={Np:4d}: worst relative error {e:.1e}{flag}")
print(f"{draws} draws, {skipped} outside the kernels' ranges, {bad} failures, worst {worst:.1e}")
sys.exit(1 if bad else 0)
