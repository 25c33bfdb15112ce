#!/usr/bin/env python3
"""Developer check of the SAVED control-network backward: the one-row rollout's activation slabs / sign records against the slabs
kernel A re-computes, and the gradients of socmx_unet_backward_saved_f32 against socmx_unet_backward_scaled_f32.
    python tools/dbg_saved.py [setting d K B]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "soc-matching_amd"), os.path.join(ROOT, "tests")]
import contextlib, io
import torch
from socmx.config import load_config
from socmx.settings import define_variables
from socmx import rollout, nets, _lib

setting, d, K, B = "double_well", 10, 20, 32
if len(sys.argv) > 4:
    setting, d, K, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
dev = torch.device("cuda:0")
cfg = load_config([f"method.setting={setting}", f"method.d={d}", f"method.num_steps={K}", f"optim.batch_size={B}"])
cfg.method.device = "cuda:0"
torch.manual_seed(0)
ts = torch.linspace(0, 1.0, K + 1).to(dev)
with contextlib.redirect_stdout(io.StringIO()):
    x0, sigma, opt_sde, sde, _ = define_variables(cfg, ts)
state0 = x0.repeat(B, 1)
net = sde.nabla_V
N = (K + 1) * B
print("saves_activations:", rollout.saves_activations(sde, state0, B, K))
ws_n, ng = _lib.C.c_int64(0), _lib.C.c_int64(0)
_lib.check(net.hip_lib().socmx_unet_backward_sizes(d, _lib.i3(net.hdims), N, _lib.C.byref(ws_n), _lib.C.byref(ng)), "sizes")
work = torch.full((ws_n.value,), float("nan"), dtype=torch.float32, device=dev)
rec = torch.zeros(N, 32, dtype=torch.int32, device=dev)
plain = rollout.hip_trajectories(sde, state0, ts, 1.0, seed=3, offset=0, want_nabla_v=True)
out = rollout.hip_trajectories(sde, state0, ts, 1.0, seed=3, offset=0, want_nabla_v=True, act_export=(work, rec))
torch.cuda.synchronize()
def _time(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
t0 = _time(lambda: rollout.hip_trajectories(sde, state0, ts, 1.0, seed=3, offset=0, want_nabla_v=True))
t1 = _time(lambda: rollout.hip_trajectories(sde, state0, ts, 1.0, seed=3, offset=0, want_nabla_v=True, act_export=(work, rec)))
print(f"  rollout {t0:.4f} ms, with the export {t1:.4f} ms")
for a, b, nm in zip(plain, out, "states noises stop frac lpd lps ltw controls nabla_v".split()):
    print(f"  {nm}: bit-identical with / without the export: {bool(torch.equal(a, b))}")
states = out[0]
gout = torch.randn(N, d, device=dev)
x = states.reshape(N, d)
# reference: the re-computing backward on its own workspace (kept: its activation slabs are what the rollout should have written)
work2 = torch.empty(ws_n.value, dtype=torch.float32, device=dev)
g_ref, flat_ref = nets.unet_backward_hip(net, x, ts, B, gout, return_flat=True, saved=None)
flat_ref = flat_ref.clone()
L = net.hip_lib()
with _lib.on_device(dev):
    flat2 = torch.empty(ng.value, dtype=torch.float32, device=dev)
    _lib.check(L.socmx_unet_backward_scaled_f32(_lib.ptr(net.packed()), _lib.ptr(net.packed_bwd()), d, _lib.i3(net.hdims), _lib.ptr(x),
                                                _lib.ptr(ts), B, N, _lib.ptr(gout), None, _lib.ptr(work2), _lib.ptr(flat2),
                                                _lib.stream_ptr(dev)), "scaled")
torch.cuda.synchronize()
widths = [16, 256, 128, 64, 128, 256]
names = ["X", "R1", "R2", "R3", "O2", "A1"]
off = 0
for w, nm in zip(widths, names):
    a = work[off * N:(off + w) * N]
    b = work2[off * N:(off + w) * N]
    if nm == "X":
        print(f"  slab {nm}: (written by kernel A)")
    else:
        bad = int((~torch.isfinite(a)).sum())
        print(f"  slab {nm}: max |rollout - recomputed| = {float((a - b).abs().max()):.3e}  (max |value| {float(b.abs().max()):.3e}, non-finite {bad})")
    off += w
_, flat_s = nets.unet_backward_hip(net, x, ts, B, gout, return_flat=True, saved=(work, rec))
torch.cuda.synchronize()
err = float((flat_s - flat_ref).norm() / flat_ref.norm())
print(f"  gradients: |saved - recomputed| / |recomputed| = {err:.3e}   max abs diff {float((flat_s - flat_ref).abs().max()):.3e}  (norm {float(flat_ref.norm()):.3e})")
off = 0
for name in _lib.UNET_LAYERS:
    lin = getattr(net, name)[0]
    nw, nb = lin.weight.numel(), lin.bias.numel()
    a, b = flat_s[off:off + nw + nb], flat_ref[off:off + nw + nb]
    print(f"    {name}: rel {float((a - b).norm() / (b.norm() + 1e-30)):.3e}")
    off += nw + nb
