import faulthandler, sys, os
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT, os.path.join(ROOT,"soc-matching_amd")]
import torch, bench
dev=torch.device("cuda",0)
torch.cuda.set_device(dev)
from socmx.train import Trainer, make_optimizer
mode = sys.argv[1] if len(sys.argv) > 1 else "both"
for (setting, d, K, gamma) in (("double_well", 10, 200, 6.0), ("OU_quadratic_easy", 2, 50, 2.0)):
    cfg, ts, x0, sde, solver = bench.build(dev, setting, d, K, gamma, 128)
    opt = make_optimizer(solver, nabla_V_lr=1e-4, M_lr=1e-3, adam_eps=1e-4)
    for graph in ((False, True) if mode == "both" else (True,)):
        tr = Trainer(solver, opt, batch_size=128, normalization_const=1.0, sync_timing=False, gemm_select=False, hip_graph=graph)
        for i in range(6):
            print(setting, graph, i, flush=True)
            tr.step()
        torch.cuda.synchronize()
        tr.join()
        if mode == "keep":
            globals().setdefault("keep", []).append(tr)
print("done")
