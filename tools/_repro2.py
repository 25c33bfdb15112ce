import faulthandler, sys, os
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT, os.path.join(ROOT,"soc-matching_amd")]
import torch, bench
dev=torch.device("cuda",0)
torch.cuda.set_device(dev)
which = sys.argv[1]
if which == "A":
    print(bench.secondary_config(dev, "cfg1", "OU_quadratic_easy", 2, 50, 128, 2.0, 12, 3, False, 0)["socm_ms_per_iter_graph"])
elif which == "B":
    print(bench.secondary_config(dev, "cfg3", "double_well", 10, 200, 128, 6.0, 12, 3, False, 0)["socm_ms_per_iter_graph"])
    print(bench.secondary_config(dev, "cfg1", "OU_quadratic_easy", 2, 50, 128, 2.0, 12, 3, False, 0)["socm_ms_per_iter_graph"])
elif which == "C":
    print(bench.secondary_config(dev, "cfg1", "OU_quadratic_easy", 2, 50, 128, 2.0, 12, 3, False, 0)["socm_ms_per_iter_graph"])
    print(bench.secondary_config(dev, "cfg5", "OU_linear", 64, 400, 512, 2.0, 4, 2, False, 0)["socm_ms_per_iter_graph"])
print("done", which)
