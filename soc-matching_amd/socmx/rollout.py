"""Batched Euler-Maruyama rollout: `stochastic_trajectories`.

Drop-in for reference SOC_matching/utils.py:17-128 -- same arguments, same
8-tuple `(states, noises, stop_indicators, fractional_timesteps, lpd, lps, ltw,
controls)` in the same step-major layouts.

Routing (explicit, never silent):
  * tensors on a CUDA (ROCm) device AND the sde is a known problem descriptor
    with the learned control, no warm start, `detach=True`
        -> ONE launch of the fused HIP kernel (csrc/socmx_rollout.hip).  If
           libsocmx.so is missing this RAISES.
  * the same with `use_learned_control` off and `sde.u` one of the tabulated
    ground-truth controls of socmx.ground_truth (LQ Riccati, OU-linear closed
    form, double-well PDE table: reference models.py:10-150), no stopping time
        -> ONE launch of csrc/socmx_rollout_ctrl.hip.
  * anything else (CPU tensors = BASELINE config 0 "plumbing"; a foreign
    `sde.u` callable; `detach=False` for the rel_entropy loss; warm start; a
    foreign NeuralSDE subclass)
        -> `eager_trajectories`, a device-agnostic torch implementation of the
           same recurrence.
Extra keyword-only arguments (not in the reference): `noise_in` injects the
(K,B,d) noise (exact parity runs), `seed`/`offset`/`row0` key the device Philox
stream (row0 = global index of this shard's first trajectory, so multi-GPU runs
are independent of the sharding).
"""
import torch

from . import _lib

import weakref
_TIDX_CHECKED = weakref.WeakKeyDictionary()   # ground-truth control -> (time grid, version, K, table rows) already range-checked
_philox_calls = 0  # advances the Philox offset so successive rollouts draw fresh noise


def _eligible_for_hip(sde, x0, detach):
    return (
        x0.is_cuda
        and detach
        and getattr(sde, "problem", None) is not None
        and getattr(sde, "use_learned_control", False)
        and not (getattr(sde, "use_warm_start", False) and getattr(sde, "u_warm_start", None))
        and hasattr(sde, "nabla_V")
        and hasattr(sde.nabla_V, "packed")
    )


def _eligible_for_hip_control(sde, x0, detach):
    """A tabulated ground-truth control (socmx.ground_truth: LinearControl, ConstantControl, LowDimControl) on a known
    setting without stopping times: socmx_rollout_control_f32."""
    pb = getattr(sde, "problem", None)
    return (x0.is_cuda and detach and pb is not None and not pb.has_phi
            and not getattr(sde, "use_learned_control", False)
            and hasattr(getattr(sde, "u", None), "hip_descriptor"))


def burst_eligible(sde, x0):
    """True when an evaluation burst (utils.py:131-231, method.py:185-221) can run as fused launches."""
    return _eligible_for_hip(sde, x0, True) or _eligible_for_hip_control(sde, x0, True)


def stochastic_trajectories(sde, x0, t, lmbd, detach=True, verbose=False, *, noise_in=None, seed=None,
                            offset=None, row0=0, key=None, want_nabla_v=False, shares_chip=False, act_export=None):
    """`want_nabla_v` (HIP path only; ignored -- no ninth entry -- on the eager path): see hip_trajectories."""
    if _eligible_for_hip(sde, x0, detach):
        return hip_trajectories(sde, x0, t, lmbd, noise_in=noise_in, seed=seed, offset=offset, row0=row0, key=key,
                                want_nabla_v=want_nabla_v, shares_chip=shares_chip, act_export=act_export)
    assert act_export is None, "act_export: the fused HIP rollout only (ask saves_activations first)"
    if _eligible_for_hip_control(sde, x0, detach):
        return hip_trajectories(sde, x0, t, lmbd, noise_in=noise_in, seed=seed, offset=offset, row0=row0)
    return eager_trajectories(sde, x0, t, lmbd, detach=detach, verbose=verbose, noise_in=noise_in)


# --------------------------------------------------------------------------------------
# fused HIP path
# --------------------------------------------------------------------------------------

class PhiloxKey:
    """Philox (seed, offset) held in DEVICE memory (3 x int64: seed, offset, ticket): the keyed rollout reads it when the kernel
    starts and ADVANCES it itself (key[1] += 1 by the last workgroup to have read it: SOCMX_ROLLOUT_ADVANCES_KEY), so a hipGraph
    that captured the launch draws fresh noise on every replay.  Eager use is equivalent to passing `seed`,
    `offset = 0, 1, 2, ...` by value.  `advance()` is the separate one-thread launch (socmx_philox_advance)."""

    def __init__(self, device, seed=None, offset=0):
        seed = torch.initial_seed() if seed is None else int(seed)
        to_i64 = lambda v: ((int(v) & (2**64 - 1)) ^ (1 << 63)) - (1 << 63)      # same 64 bits, as a signed value
        self.key = torch.tensor([to_i64(seed), to_i64(offset), 0], dtype=torch.int64, device=device)

    def advance(self, inc=1):
        with _lib.on_device(self.key.device):
            _lib.check(_lib.lib().socmx_philox_advance(self.key.data_ptr(), int(inc), _lib.stream_ptr(self.key.device)),
                       "socmx_philox_advance")


def saves_activations(sde, x0, B, K, detach=True):
    """True when a rollout of B rows and K steps of this SDE can save the control network's activations for the backward
    (socmx_rollout_saves_activations: the one-row kernel, d <= 15, default widths, (K+1) B a multiple of 16)."""
    if not _eligible_for_hip(sde, x0, detach):
        return False
    net = sde.nabla_V
    return net.hip_lib().socmx_rollout_saves_activations(sde.problem.c_struct(), _lib.i3(net.hdims), int(B), int(K)) == 1


def hip_trajectories(sde, x0, t, lmbd, *, noise_in=None, seed=None, offset=None, row0=0, phase_cycles=None,
                     costs_only=False, key=None, want_nabla_v=False, shares_chip=False, act_export=None):
    """`phase_cycles`: optional int64 CUDA tensor ((B+15)//16, 64) -> run the instrumented kernel.
    `costs_only`: write lpd / lps / ltw only (the five trajectory entries of the returned tuple are None).
    `key`: a PhiloxKey -- seed/offset are read from device memory and advanced behind the launch.
    `want_nabla_v`: also return nabla_V(t_k, X_k), k = 0..K, as a ninth entry (K+1, B, d).
    `act_export`: (workspace fp32, records int32 ((K+1) B, 32)) -- the launch saves the control network's activations and ReLU signs for
    socmx_unet_backward_saved_f32 (include/socmx.h; ask saves_activations() first: a launch that cannot raises)."""
    global _philox_calls
    L = _lib.lib()
    pb = sde.problem
    dev = x0.device
    B, d = x0.shape
    K = t.shape[0] - 1
    x0c = x0.detach().to(torch.float32).contiguous()
    tc = t.detach().to(device=dev, dtype=torch.float32).contiguous()
    f32 = dict(dtype=torch.float32, device=dev)
    if costs_only:
        states = noises = controls = stop = frac = None
    else:
        states = torch.empty(K + 1, B, d, **f32)
        noises = torch.empty(K, B, d, **f32)
        controls = torch.empty(K, B, d, **f32)
        stop = torch.empty(K + 1, B, **f32)
        frac = torch.empty(K, B, **f32)
    lpd = torch.empty(B, **f32)
    lps = torch.empty(B, **f32)
    ltw = torch.empty(B, **f32)
    if noise_in is not None:
        noise_in = noise_in.detach().to(**f32).contiguous()
        assert noise_in.shape == (K, B, d), noise_in.shape
    nabla_v = torch.empty(K + 1, B, d, **f32) if want_nabla_v else None
    if not getattr(sde, "use_learned_control", False):
        # tabulated ground-truth control (models.py:10-150): socmx_rollout_control_f32
        assert key is None and phase_cycles is None and not want_nabla_v
        ckind, table, tidx, n_x, xb, dx = sde.u.hip_descriptor(tc)
        table = table.detach().to(**f32).contiguous()
        tidx = tidx.to(device=dev, dtype=torch.int32).contiguous()
        # the kernel indexes the table with tidx unchecked; an index past the table raises in the reference's lookup
        # (models.py:15-23, 72-75, 98-150) -- checked once per (control, time grid), not per rollout
        # (keyed on the CALLER's time-grid tensor -- the object, held weakly, and its version counter -- not on the fp32 /
        #  contiguous / on-device copy made above, which is a new allocation on every call whenever the caller's grid is not
        #  already in that form: the check's host synchronisation then recurred per rollout, and an address reused by
        #  another grid of the same length skipped it)
        seen = _TIDX_CHECKED.get(sde.u)         # (a module-level weak table: the control object stays picklable)
        if not (seen is not None and seen[0]() is t and seen[1:] == (t._version, K, int(table.shape[0]))):
            lo, hi = int(tidx.min()), int(tidx.max())
            if lo < 0 or hi >= table.shape[0]:
                raise IndexError(f"ground-truth control: time index range [{lo}, {hi}] outside the table of "
                                 f"{table.shape[0]} rows")
            _TIDX_CHECKED[sde.u] = (weakref.ref(t), t._version, K, int(table.shape[0]))
        ctrl = _lib.Control(kind=ckind, n_t=int(table.shape[0]), n_x=int(n_x), table=_lib.ptr(table),
                            tidx=tidx.data_ptr(), xb=float(xb), delta_x=float(dx))
        if seed is None:
            seed = torch.initial_seed()
        if offset is None:
            offset = _philox_calls
            _philox_calls += 1
        with _lib.on_device(dev):
            status = L.socmx_rollout_control_f32(
                pb.c_struct(), ctrl, _lib.ptr(x0c), _lib.ptr(tc), B, K, float(lmbd), int(seed) & (2**64 - 1),
                int(offset) & (2**64 - 1), int(row0), _lib.ptr(noise_in), _lib.ptr(states), _lib.ptr(noises),
                _lib.ptr(controls), _lib.ptr(stop), _lib.ptr(frac), _lib.ptr(lpd), _lib.ptr(lps), _lib.ptr(ltw),
                _lib.stream_ptr(dev))
        _lib.check(status, "socmx_rollout_control_f32")
        return states, noises, stop, frac, lpd, lps, ltw, controls
    net = sde.nabla_V
    L = net.hip_lib()              # (the default library, or the variant compiled for this architecture)
    with _lib.on_device(dev):
        head = (pb.c_struct(), _lib.ptr(net.packed()), _lib.i3(net.hdims), _lib.ptr(x0c), _lib.ptr(tc), B, K,
                float(lmbd))
        tail = (int(row0), _lib.ptr(noise_in), _lib.ptr(states), _lib.ptr(noises), _lib.ptr(controls),
                _lib.ptr(stop), _lib.ptr(frac), _lib.ptr(lpd), _lib.ptr(lps), _lib.ptr(ltw))
        if key is None:
            if seed is None:
                seed = torch.initial_seed()
            if offset is None:
                offset = _philox_calls
                _philox_calls += 1
            mid = (int(seed) & (2**64 - 1), int(offset) & (2**64 - 1))
        else:
            assert phase_cycles is None and key.key.device == dev
            mid = (0, 0)
        if phase_cycles is not None:
            assert phase_cycles.dtype == torch.int64 and phase_cycles.is_cuda and phase_cycles.is_contiguous()
            assert not want_nabla_v and act_export is None
            status = L.socmx_rollout_phase_cycles_f32(*head, *mid, *tail, phase_cycles.data_ptr(), _lib.stream_ptr(dev))
        elif key is not None or want_nabla_v or shares_chip or act_export is not None:
            # (shares_chip: the caller runs chip-filling kernels beside this launch -- SOCMX_ROLLOUT_SHARES_CHIP, include/socmx.h)
            extra = _lib.RolloutExtra(key=None if key is None else key.key.data_ptr(), nabla_v=_lib.ptr(nabla_v),
                                      flags=(_lib.ROLLOUT_SHARES_CHIP if shares_chip else 0) |
                                            (_lib.ROLLOUT_ADVANCES_KEY if key is not None else 0), reserved=0,
                                      act_workspace=None if act_export is None else _lib.ptr(act_export[0]),
                                      act_records=None if act_export is None else act_export[1].data_ptr())
            status = L.socmx_rollout_ex_f32(*head, *mid, *tail, extra, _lib.stream_ptr(dev))
        else:
            status = L.socmx_rollout_f32(*head, *mid, *tail, _lib.stream_ptr(dev))
    _lib.check(status, "socmx_rollout_f32")
    out = (states, noises, stop, frac, lpd, lps, ltw, controls)
    return out + (nabla_v,) if want_nabla_v else out


def burst_log_weights(sde, x0_row, t, lmbd, n, *, noise_in=None, seed=None, offset=None, row0=0, chunk_rows=16384):
    """(lpd, lps, ltw) of `n` independent trajectories from the point x0_row -- the evaluation bursts of
    utils.py:131-231 / method.py:185-221 (n_batches rollouts of batch_size rows each) as costs-only launches of at
    most `chunk_rows` rows: nothing of size (K, n, d) is allocated.  The Philox stream is keyed by the global row
    (row0 + row), so the result does not depend on the chunking (nor on how `n` is split over ranks)."""
    global _philox_calls
    if seed is None:
        seed = torch.initial_seed()
    if offset is None:
        offset = _philox_calls
        _philox_calls += 1
    x0_row = x0_row.reshape(1, -1)
    parts = []
    for r0 in range(0, n, chunk_rows):
        r1 = min(n, r0 + chunk_rows)
        nz = None if noise_in is None else noise_in[:, r0:r1]
        out = hip_trajectories(sde, x0_row.expand(r1 - r0, -1), t, lmbd, noise_in=nz, seed=seed, offset=offset,
                               row0=row0 + r0, costs_only=True)
        parts.append((out[4], out[5], out[6]))
    return tuple(torch.cat([p[i] for p in parts]) for i in range(3))


# --------------------------------------------------------------------------------------
# device-agnostic torch path (CPU plumbing config, foreign controls, detach=False)
# --------------------------------------------------------------------------------------

def eager_trajectories(sde, x0, t, lmbd, detach=True, verbose=False, noise_in=None):
    B = x0.shape[0]
    dev = x0.device
    ones = torch.ones(B, device=dev)
    stopping = hasattr(sde, "Phi")            # the reference keys on the attribute, not the config flag
    sigma = sde.sigma
    xs, eps_all, us, stops, fracs = [x0], [], [], [ones], []
    lpd = torch.zeros(B, device=dev)
    lps = torch.zeros(B, device=dev)
    running = ones
    x = x0
    for k in range(t.shape[0] - 1):
        t0 = t[k]
        dt = t[k + 1] - t0
        eps = noise_in[k] if noise_in is not None else torch.randn_like(x)
        u = sde.control(t0, x, verbose=verbose)
        step = (sde.b(t0, x) + u @ sigma.T) * dt + torch.sqrt(lmbd * dt) * (eps @ sigma.T)
        x_prop = x + running.unsqueeze(1) * step
        if stopping:
            phi_b, phi_a = sde.Phi(x), sde.Phi(x_prop)
            alive = ((phi_b > 0) & (phi_a > 0)).to(torch.float)
            crossed = ((phi_b > 0) & (phi_a < 0)).to(torch.float)
            part = crossed * (phi_b / (phi_b - phi_a + 1e-6) + 1e-6)
            c = crossed.unsqueeze(1)
            x_new = c * (x + part.unsqueeze(1) * running.unsqueeze(1) * step) + (1 - c) * x_prop
            h = crossed * part**2 * dt + alive * dt       # squared fraction, as the reference does
            running = sde.Phi(x_new) > 0
            stops.append(running)
        else:
            x_new = x_prop
            h = dt * ones
            stops.append(ones)
        fracs.append(h)
        lpd = lpd + h / lmbd * (-sde.f(t0, x_new) - 0.5 * (u * u).sum(1))   # f at the NEW state, OLD time
        lps = lps + torch.sqrt(h / lmbd) * (-(u * eps).sum(1))
        xs.append(x_new)
        eps_all.append(eps)
        us.append(u)
        x = x_new
    ltw = -sde.g(x) / lmbd
    out = (torch.stack(xs), torch.stack(eps_all), torch.stack(stops), torch.stack(fracs), lpd, lps, ltw,
           torch.stack(us))
    if detach:
        return tuple(o.detach() for o in out)
    return out[:3] + (out[3].detach(),) + out[4:]
