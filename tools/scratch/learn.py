import os, sys, io, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "soc-matching_amd")]
import torch
from socmx.config import load_config
from socmx.settings import define_variables
from SOC_matching.method import SOC_Solver
from socmx.train import Trainer, make_optimizer
dev = torch.device("cuda:0")
cfg = load_config(["method.setting=OU_quadratic_easy", "method.d=2", "method.num_steps=50", "method.gamma=2.0",
                   "method.scaling_factor_M=0.1", "optim.M_lr=1e-3", "optim.batch_size=128"])
cfg.method.device = "cuda:0"
torch.manual_seed(0)
ts = torch.linspace(0, 1.0, 51).to(dev)
with contextlib.redirect_stdout(io.StringIO()):
    x0, sigma, opt_sde, sde, _ = define_variables(cfg, ts)
solver = SOC_Solver(sde, x0, None, T=1.0, num_steps=50, lmbd=1.0, d=2, sigma=sigma)
opt = make_optimizer(solver, M_lr=1e-3)
for mode in ("default",):
    tr = Trainer(solver, opt, 128, sync_timing=False)
    p0 = sde.nabla_V.packed().clone()
    w0 = sde.nabla_V.down_0[0].weight.detach().clone()
    def ctrl():
        with torch.no_grad():
            m, s, _ = solver.control_objective(128, total_n_samples=65536)
        return float(m), float(s)
    print("before", ctrl())
    for it in range(1500):
        out = tr.step()
        if it in (0, 1, 2): print("it", it, "loss", float(out["loss"]), "gradnorm", float(out.get("grad_norm_sqd", -1)))
    torch.cuda.synchronize()
    print("weight change", float((sde.nabla_V.down_0[0].weight - w0).abs().max()), "version", sde.nabla_V.down_0[0].weight._version)
    print("packed change", float((sde.nabla_V.packed() - p0).abs().max()))
    print("after", ctrl())
