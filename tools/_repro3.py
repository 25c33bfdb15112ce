import faulthandler, sys, os, time
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT, os.path.join(ROOT,"soc-matching_amd")]
import torch, bench
dev=torch.device("cuda",0)
torch.cuda.set_device(dev)
from socmx.train import Trainer, make_optimizer
from socmx import rollout
opts = set(sys.argv[1:])
cfg, ts, x0, sde, solver = bench.build(dev, "OU_quadratic_easy", 2, 50, 2.0, 128)
if "roll" in opts:
    state0 = x0.repeat(128, 1)
    for i in range(5):
        rollout.stochastic_trajectories(sde, state0, ts, 1.0, seed=0, offset=i, row0=0)
    torch.cuda.synchronize()
opt = make_optimizer(solver, nabla_V_lr=cfg.optim.nabla_V_lr, M_lr=cfg.optim.M_lr, adam_eps=cfg.optim.adam_eps)
def run(graph, n):
    tr = Trainer(solver, opt, batch_size=128, normalization_const=1.0, sync_timing=False, gemm_select=("gs" in opts), hip_graph=graph)
    for i in range(n):
        info = tr.step()
    torch.cuda.synchronize()
    tr.join()
    return info
if "eager" in opts:
    keep = run(False, 15 if "long" in opts else 6)
    if "drop" in opts:
        keep = None
    elif "dropout" in opts:
        keep = {k: v for k, v in keep.items() if k != "out"}
run(True, 15 if "long" in opts else 6)
print("done", sorted(opts))
