"""Developer script: the WIDE pair-grid-network kernels step by step (forward, backward tile kernel, weight gradient)."""
import os, sys, faulthandler
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "soc-matching_amd"), os.path.join(ROOT, "tests")]
import torch, numpy as np
from socmx import nets, loss as L, _lib
DEV = "cuda:0"
d, hdims, K = int(sys.argv[1]), (int(sys.argv[2]), int(sys.argv[3])), int(sys.argv[4])
torch.manual_seed(0)
M = nets.SigmoidMLP(dim=d, hdims=hdims, gamma=torch.nn.Parameter(torch.tensor([1.0])), scaling_factor=0.5).to(DEV)
ts = torch.linspace(0, 1, K + 1).to(DEV)
t_vec, s_vec, _, _ = L.pair_times(ts, 1.0, K)
Np = t_vec.shape[0]
print("Np", Np, "supported", nets.pair_net_supported(M, Np), flush=True)
c = lambda x: x.detach().to(torch.float32).contiguous()
params = [c(p_) for l in (0, 2, 4) for p_ in (M.sigmoid_layers[l].weight, M.sigmoid_layers[l].bias)]
net, dnet, packed = nets.pair_net_forward(d, hdims, params, c(t_vec), c(s_vec))
torch.cuda.synchronize()
print("forward ok", float(net.abs().max()), flush=True)
M.fused_pair_net = False
n0, d0 = M.forward_with_ds(t_vec, s_vec, raw=True)
print("fwd err", float((net - n0).abs().max()), float((dnet - d0).abs().max()), flush=True)
g = torch.Generator().manual_seed(2)
gn = torch.randn(Np, d, d, generator=g).to(DEV)
gd = torch.randn(Np, d, d, generator=g).to(DEV)
grads = nets.pair_net_backward(d, hdims, [p_.shape for p_ in params], packed, c(t_vec), c(s_vec), gn, gd)
torch.cuda.synchronize()
print("backward ok", flush=True)
torch.autograd.backward([n0, d0], [gn, gd])
for (k, p_), a in zip(M.sigmoid_layers.named_parameters(), grads):
    b = p_.grad
    print(k, float((a - b).norm() / b.norm()), flush=True)
