#!/usr/bin/env python3
"""Developer check of the one-row rollout kernel: nabla_v hand-over per time index against the oracle's network."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "soc-matching_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from oracle import socm_oracle as O
from test_host_cpu import build_sde, GOLDEN
from socmx import rollout as R
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3_double_well_d10_K200"
sde, aux = build_sde(name, "cuda:0")
pb, vp, mp, gamma, oaux = O.load_fixture(os.path.join(GOLDEN, name + ".npz"))
B, K, d = aux["B"], aux["K"], aux["d"]
x0 = aux["x0"].repeat(B, 1)
ext = R.hip_trajectories(sde, x0, aux["ts"], aux["lmbd"], noise_in=aux["noise"], want_nabla_v=True)
states = torch.from_numpy(aux["z"]["roll_states"])
tx = torch.cat([oaux["ts"].reshape(-1, 1, 1).expand(K + 1, B, 1), states], -1).reshape(-1, d + 1)
with torch.no_grad():
    want = O.unet_forward(vp, tx).reshape(K + 1, B, d).numpy()
got = ext[8].cpu().numpy()
err = np.abs(got - want).reshape(K + 1, -1).max(1)
print("max err per k (first 5, last 5):", err[:5], err[-5:])
bad = np.where(err > 1e-4)[0]
print("bad k:", bad[:20], "count", len(bad))
if len(bad):
    k = bad[0]
    print("k", k, "got", got[k, 0], "want", want[k, 0])
    print("rows bad at k:", np.where(np.abs(got[k] - want[k]).max(1) > 1e-4)[0])
# determinism + truncated grids
for Kt in (1, 2, 3, 7, 50):
    ts = aux["ts"][:Kt + 1].contiguous()
    nz = aux["noise"][:Kt].contiguous()
    e1 = R.hip_trajectories(sde, x0, ts, aux["lmbd"], noise_in=nz, want_nabla_v=True)
    e2 = R.hip_trajectories(sde, x0, ts, aux["lmbd"], noise_in=nz, want_nabla_v=True)
    g = e1[8].cpu().numpy()
    print("K", Kt, "deterministic", bool(torch.equal(e1[8], e2[8])), "err per k", np.abs(g - want[:Kt + 1]).reshape(Kt + 1, -1).max(1)[-3:])
from socmx import nets
txK = torch.cat([aux["ts"][-1].reshape(1, 1).expand(B, 1), ext[0][-1]], -1)
ref = nets.unet_forward_hip(sde.nabla_V, txK).cpu().numpy()
print("terminal vs unet_forward_hip:", np.abs(got[K] - ref).max(), " oracle vs unet_forward_hip", np.abs(want[K] - ref).max())
print("got[K][0]-want", (got[K][0] - want[K][0]))
print("got[K][1]-want", (got[K][1] - want[K][1]))
