"""GPU time (hipGraph replay, no launch overhead) of the GEMM shapes of nabla_V-on-trajectory (25,728 rows):
plain wgrad vs batched split-K wgrad, forward, dgrad, bias column-sum."""
import torch
dev = "cuda"
R = 25728

def gpu_us(fn, n=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for (o, i) in [(256, 256), (256, 128), (128, 256), (128, 128), (128, 64), (64, 128), (256, 11), (10, 256)]:
    gy = torch.randn(R, o, device=dev); x = torch.randn(R, i, device=dev); w = torch.randn(o, i, device=dev)
    b = torch.zeros(o, device=dev)
    base = gpu_us(lambda: gy.t() @ x)
    res = {}
    for S in (8, 16, 24, 48):
        gy3 = gy.view(S, R // S, o); x3 = x.view(S, R // S, i)
        res[S] = gpu_us(lambda: torch.bmm(gy3.transpose(1, 2), x3).sum(0))
    fwd = gpu_us(lambda: torch.addmm(b, x, w.t()))
    dg = gpu_us(lambda: gy @ w)
    bs = gpu_us(lambda: gy.sum(0))
    mk = gpu_us(lambda: gy * (gy > 0))
    print(f"out={o:4d} in={i:4d}: wgrad plain {base:6.1f} us | splitK " + " ".join(f"S{S}:{v:6.1f}" for S, v in res.items())
          + f" | fwd {fwd:6.1f} dgrad {dg:6.1f} bias-sum {bs:6.1f} mask-mul {mk:6.1f}")
