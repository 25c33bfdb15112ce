"""ctypes binding of libsocmx.so (C ABI: include/socmx.h).

The library is built in-tree by `make -C soc-matching_amd/csrc` (or
`__graft_entry__.build()`) and sits next to this file.  There is no fallback:
`lib()` raises if the shared object is missing or does not export the ABI, so a
GPU run can never silently take another path.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# (SOCMX_LIB: a developer build of the same ABI elsewhere -- `make PROF=1`, plan sweeps -- instead of the shipped library)
LIB_PATH = os.environ.get("SOCMX_LIB") or os.path.join(_HERE, "libsocmx.so")

OU_QUADRATIC, OU_LINEAR, DOUBLE_WELL, MOLECULAR_DYNAMICS = 0, 1, 2, 3
SIGMA_IDENTITY = 1

# order of socmx_unet.weight[] / bias[] (include/socmx.h, SOCMX_L_*)
UNET_LAYERS = ("down_0", "down_1", "down_2", "res_0", "res_1", "res_2", "up_2", "up_1", "up_0")

_fp = C.c_void_p  # device pointers travel as plain addresses


class Problem(C.Structure):
    _fields_ = [("kind", C.c_int32), ("d", C.c_int32), ("flags", C.c_int32), ("reserved", C.c_int32),
                ("sigma", _fp), ("sigma_inv_t", _fp),
                ("A", _fp), ("P", _fp), ("Q", _fp), ("omega", _fp), ("kappa", _fp), ("nu", _fp)]


class RolloutExtra(C.Structure):
    _fields_ = [("key", _fp), ("nabla_v", _fp), ("flags", C.c_uint32), ("reserved", C.c_uint32),
                ("act_workspace", _fp), ("act_records", _fp)]


ROLLOUT_SHARES_CHIP = 1
ROLLOUT_ADVANCES_KEY = 2      # include/socmx.h: the launch itself performs key[1] += 1 (key = {seed, offset, ticket})


class Control(C.Structure):
    _fields_ = [("kind", C.c_int32), ("n_t", C.c_int32), ("n_x", C.c_int32), ("reserved", C.c_int32),
                ("table", _fp), ("tidx", _fp), ("xb", C.c_float), ("delta_x", C.c_float)]


CTRL_LINEAR, CTRL_CONSTANT, CTRL_TABLE = 1, 2, 3


class Unet(C.Structure):
    _fields_ = [("d", C.c_int32), ("hdims", C.c_int32 * 3), ("weight", _fp * 9), ("bias", _fp * 9)]


_I3 = C.c_int32 * 3
_I2 = C.c_int32 * 2
_P3 = _fp * 3

# name -> (restype, argtypes); one entry per declaration in include/socmx.h
PROTOTYPES = {
    "socmx_version": (C.c_int, []),
    "socmx_capabilities": (C.c_int, [C.c_char_p, C.c_int]),
    "socmx_unet_packed_floats": (C.c_size_t, [C.c_int32, C.POINTER(C.c_int32)]),
    "socmx_unet_pack_f32": (C.c_int, [C.POINTER(Unet), _fp, _fp]),
    "socmx_unet_forward_f32": (C.c_int, [_fp, C.c_int32, C.POINTER(C.c_int32), _fp, C.c_int64, _fp, _fp]),
    "socmx_unet_packed_bwd_floats": (C.c_size_t, [C.c_int32, C.POINTER(C.c_int32)]),
    "socmx_unet_pack_bwd_f32": (C.c_int, [C.POINTER(Unet), _fp, _fp]),
    "socmx_unet_backward_sizes": (C.c_int, [C.c_int32, C.POINTER(C.c_int32), C.c_int64, C.POINTER(C.c_int64),
                                            C.POINTER(C.c_int64)]),
    "socmx_unet_backward_f32": (C.c_int, [_fp, _fp, C.c_int32, C.POINTER(C.c_int32), _fp, _fp, C.c_int32, C.c_int64,
                                          _fp, _fp, _fp, _fp]),
    "socmx_unet_backward_scaled_f32": (C.c_int, [_fp, _fp, C.c_int32, C.POINTER(C.c_int32), _fp, _fp, C.c_int32, C.c_int64,
                                                 _fp, _fp, _fp, _fp, _fp]),
    "socmx_unet_backward_saved_f32": (C.c_int, [_fp, _fp, C.c_int32, C.POINTER(C.c_int32), _fp, _fp, C.c_int32, C.c_int64,
                                                _fp, _fp, _fp, _fp, _fp, _fp]),
    "socmx_rollout_saves_activations": (C.c_int, [C.POINTER(Problem), C.POINTER(C.c_int32), C.c_int32, C.c_int32]),
    "socmx_mnet_packed_floats": (C.c_size_t, [C.c_int32, C.POINTER(C.c_int32)]),
    "socmx_mnet_pack_f32": (C.c_int, [C.c_int32, C.POINTER(C.c_int32), C.c_int32] + [_fp] * 8),
    "socmx_mnet_forward_f32": (C.c_int, [_fp, C.c_int32, C.POINTER(C.c_int32), _fp, _fp, _fp, C.c_int64, _fp, _fp, _fp]),
    "socmx_mnet_backward_sizes": (C.c_int, [C.c_int32, C.POINTER(C.c_int32), C.c_int32, C.c_int64, C.POINTER(C.c_int64),
                                            C.POINTER(C.c_int64)]),
    "socmx_mnet_backward_f32": (C.c_int, [_fp, C.c_int32, C.POINTER(C.c_int32), C.c_int32, _fp, _fp, _fp, C.c_int64, _fp,
                                          _fp, _fp, _fp, _fp]),
    "socmx_rollout_f32": (C.c_int, [C.POINTER(Problem), _fp, C.POINTER(C.c_int32), _fp, _fp, C.c_int32,
                                    C.c_int32, C.c_float, C.c_uint64, C.c_uint64, C.c_int64, _fp,
                                    _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp]),
    "socmx_rollout_ex_f32": (C.c_int, [C.POINTER(Problem), _fp, C.POINTER(C.c_int32), _fp, _fp, C.c_int32,
                                       C.c_int32, C.c_float, C.c_uint64, C.c_uint64, C.c_int64, _fp,
                                       _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, C.POINTER(RolloutExtra), _fp]),
    "socmx_rollout_control_f32": (C.c_int, [C.POINTER(Problem), C.POINTER(Control), _fp, _fp, C.c_int32, C.c_int32,
                                            C.c_float, C.c_uint64, C.c_uint64, C.c_int64, _fp,
                                            _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp]),
    "socmx_philox_advance": (C.c_int, [_fp, C.c_uint64, _fp]),
    "socmx_rollout_phase_cycles_f32": (C.c_int, [C.POINTER(Problem), _fp, C.POINTER(C.c_int32), _fp, _fp, C.c_int32,
                                                 C.c_int32, C.c_float, C.c_uint64, C.c_uint64, C.c_int64, _fp,
                                                 _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp]),
    "socmx_weights_stats_f32": (C.c_int, [_fp, _fp, _fp, C.c_int32, _fp, _fp, _fp]),
    "socmx_weights_stats_scalars_f32": (C.c_int, [_fp, _fp, _fp, C.c_int32, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp]),
    "socmx_shard_stats_f32": (C.c_int, [C.c_int32, _fp, C.c_int32, C.c_int32, C.c_int32, _fp, _fp, _fp, _fp]),
    "socmx_num_pairs": (C.c_int64, [C.c_int32]),
    "socmx_matching_target_f32": (C.c_int, [C.c_int32, C.POINTER(Problem), C.c_int32, C.c_int32, _fp, C.c_float, C.c_float,
                                            _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp]),
    "socmx_socm_residual_f32": (C.c_int, [C.POINTER(Problem), C.c_int32, C.c_int32, _fp, _fp, _fp, C.c_float, _fp, _fp, _fp, _fp]),
    "socmx_socm_objective_workspace_floats": (C.c_int64, [C.c_int32, C.c_int32]),
    "socmx_girsanov_fwd_f32": (C.c_int, [C.POINTER(Problem), C.c_int32, C.c_int32, C.c_float, C.c_int32] + [_fp] * 9),
    "socmx_girsanov_bwd_f32": (C.c_int, [C.POINTER(Problem), C.c_int32, C.c_int32, C.c_float] + [_fp] * 9),
    "socmx_socm_prep_f32": (C.c_int, [C.POINTER(Problem), _fp, C.c_int32, C.c_int32, C.c_float,
                                      _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp]),
    "socmx_socm_target_fwd_f32": (C.c_int, [C.POINTER(Problem), C.c_int32, C.c_int32, _fp, _fp, _fp, _fp,
                                            _fp, _fp, _fp, C.c_float, _fp, _fp, _fp, _fp, _fp]),
    "socmx_socm_target_bwd_f32": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _fp, _fp, _fp, _fp, _fp, _fp,
                                            _fp]),
    "socmx_socm_stopping_target_fwd_f32": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _fp, _fp, _fp, _fp, C.c_float]
                                           + [_fp] * 9),
    "socmx_socm_stopping_target_bwd_f32": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _fp, _fp, _fp, _fp, C.c_float]
                                           + [_fp] * 14),
    "socmx_iteration_scalars_f32": (C.c_int, [C.c_int32] + [_fp] * 8 + [C.c_double, C.c_double, _fp, _fp, _fp]),
    "socmx_iteration_scalars_hist_f32": (C.c_int, [C.c_int32] + [_fp] * 8 + [C.c_double, C.c_double, _fp, _fp, _fp, C.c_int32, _fp, _fp]),
    "socmx_adam_step_scalars_hist_f32": (C.c_int, [_fp, C.c_int32, C.c_int64, _fp, _fp, _fp, C.c_double, C.c_float, C.c_float,
                                                   C.c_float, C.c_float, _fp, _fp, _fp, _fp, _fp, _fp, _fp, C.c_double, _fp, _fp,
                                                   C.c_int32, _fp, _fp]),
    "socmx_adam_step_f32": (C.c_int, [_fp, C.c_int32, C.c_int64, _fp, _fp, _fp, C.c_double, C.c_float, C.c_float,
                                      C.c_float, C.c_float, _fp, _fp, _fp]),
    "socmx_adam_step_scalars_f32": (C.c_int, [_fp, C.c_int32, C.c_int64, _fp, _fp, _fp, C.c_double, C.c_float, C.c_float,
                                              C.c_float, C.c_float, _fp, _fp, _fp, _fp, _fp, _fp, _fp, C.c_double, _fp, _fp]),
    "socmx_colsum_blocks": (C.c_int32, [C.c_int64, C.c_int32]),
    "socmx_colsum_f32": (C.c_int, [_fp, C.c_int64, C.c_int32, _fp, _fp, _fp]),
    "socmx_linear_bwd_finish_f32": (C.c_int, [_fp, C.c_int32, C.c_int64, _fp, _fp, _fp, C.c_int32, C.c_int32, _fp, _fp]),
    "socmx_relu_bwd_colsum_f32": (C.c_int, [_fp, _fp, C.c_int64, C.c_int32, _fp, _fp, _fp, _fp]),
    "socmx_socm_target_fwd_net_f32": (C.c_int, [C.POINTER(Problem), C.c_int32, C.c_int32, _fp, _fp, _fp, _fp, _fp,
                                                _fp, _fp, _fp, _fp, C.c_float, _fp, _fp, _fp, _fp, _fp]),
    "socmx_socm_target_bwd_net_f32": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _fp, _fp, _fp, _fp, _fp, _fp, _fp,
                                                _fp, _fp, _fp, _fp, _fp, _fp]),
}

_lib = None


class SocmxError(RuntimeError):
    pass


def available():
    return os.path.exists(LIB_PATH)


def lib():
    """The loaded library.  Raises SocmxError when it is missing -- by design no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SocmxError(
                f"{LIB_PATH} not found: build the HIP extension first "
                "(`make -C soc-matching_amd/csrc` or `python -c 'import __graft_entry__ as g; g.build()'`). "
                "socmx has no CPU/torch fallback for the GPU hot path.")
        # torch is imported above so that its bundled libamdhip64.so.7 is the HIP runtime in this
        # process; libsocmx.so's DT_NEEDED of the same soname then binds to it.
        handle = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in PROTOTYPES.items():
            try:
                fn = getattr(handle, name)
            except AttributeError as e:
                raise SocmxError(f"{LIB_PATH} does not export {name}: stale build?") from e
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


# ---- architecture variants ---------------------------------------------------------------------------------------------
# libsocmx.so carries constexpr-specialised rollout / control-network-backward kernels for the reference's default
# arch.hdims = [256,128,64] (configs/soc.yaml:33-35) and a descriptor-driven form for everything else (~1.6x slower per
# step).  A VARIANT library is the same sources compiled with another architecture's padded widths as the constexpr ones
# (csrc/Makefile VARIANT=h0_h1_h2, ~2.5 min of hipcc): `variant(hdims)` loads it -- building it first when asked to --
# and the calls that depend on the architecture (rollout, control-network forward / backward) go through it.
_variants = {}
DEFAULT_HDIMS = (256, 128, 64)


def _pad16(v):
    return (int(v) + 15) // 16 * 16


def specialize_enabled():
    """Build missing variants at first use?  Off by default (a first use would cost minutes); main.py turns it on from the
    `backend.specialize_arch` config key, `SOCMX_SPECIALIZE=1` does from the environment.  An existing variant library is
    always used."""
    return os.environ.get("SOCMX_SPECIALIZE", "0") not in ("", "0") or _specialize_flag[0]


_specialize_flag = [False]


def set_specialize(on=True):
    _specialize_flag[0] = bool(on)


def variant_path(hdims):
    hp = tuple(_pad16(h) for h in hdims)
    return os.path.join(_HERE, "libsocmx_%d_%d_%d.so" % hp)


def build_variant(hdims, quiet=False):
    """`make -C csrc VARIANT=h0_h1_h2` (hipcc; cross-compiles without a GPU; a no-op when the library is up to date; the
    Makefile links to a temporary name and renames it into place, so a concurrent reader never maps a half-written file).
    Returns the library path."""
    import subprocess
    hp = tuple(_pad16(h) for h in hdims)
    csrc = os.path.join(os.path.dirname(_HERE), "csrc")
    if not quiet and not os.path.exists(variant_path(hdims)):
        print(f"socmx: compiling kernels specialised for arch.hdims -> {list(hp)} (one-time, a few minutes) ...", flush=True)
    res = subprocess.run(["make", "-C", csrc, "-j8", "VARIANT=%d_%d_%d" % hp], capture_output=True, text=True)
    if res.returncode != 0:
        raise SocmxError("building the architecture variant failed:\n" + res.stdout[-2000:] + res.stderr[-2000:])
    return variant_path(hdims)


class _VariantLock:
    """Inter-process lock (flock on `<library>.lock`) around build + load of one variant: the jobs of a sweep, or the ranks
    of a node, that start together with the same non-default arch.hdims build it once and load it complete.  Where the
    package directory is read-only the lock is skipped (nothing can be built there either)."""

    def __init__(self, path):
        self.path, self.fd = path + ".lock", None

    def __enter__(self):
        try:
            import fcntl
            self.fd = os.open(self.path, os.O_CREAT | os.O_RDWR, 0o644)
            fcntl.flock(self.fd, fcntl.LOCK_EX)
        except OSError:
            if self.fd is not None:
                os.close(self.fd)
            self.fd = None
        return self

    def __exit__(self, *exc):
        if self.fd is not None:
            import fcntl
            fcntl.flock(self.fd, fcntl.LOCK_UN)
            os.close(self.fd)


def variant(hdims, build=None):
    """The library whose constexpr kernels match `hdims` (padded to multiples of 16): the default library for the default
    architecture, an existing variant library, a freshly built one when `specialize_enabled()` (or `build=True`), else the
    default library (whose descriptor-driven kernels take any architecture).  With building enabled `make` always runs -- a
    no-op when the library is newer than its sources, a rebuild after a source edit.  A build that fails (no hipcc, a
    read-only package directory) is reported once and the default library serves the architecture: slower, never wrong."""
    hp = tuple(_pad16(h) for h in hdims)
    if hp == DEFAULT_HDIMS:
        return lib()
    if hp in _variants:
        return _variants[hp]
    path = variant_path(hdims)
    base = lib()                                          # (also makes torch's HIP runtime the one in this process)
    with _VariantLock(path):
        if specialize_enabled() if build is None else build:
            try:
                build_variant(hdims)
            except (SocmxError, OSError) as e:
                import warnings
                warnings.warn(f"socmx: could not build the kernel variant for arch.hdims {list(hp)} ({str(e)[-300:]}); "
                              + ("using the library already there" if os.path.exists(path) else
                                 "the descriptor-driven kernels of libsocmx.so serve this architecture (~1.6x slower per step)"))
        handle = None
        if os.path.exists(path):
            handle = C.CDLL(path, mode=C.RTLD_LOCAL)
            for name, (res, args) in PROTOTYPES.items():
                fn = getattr(handle, name)
                fn.restype = res
                fn.argtypes = args
            if handle.socmx_version() != base.socmx_version():
                raise SocmxError(f"{path} is a stale build (version {handle.socmx_version()} != {base.socmx_version()}): "
                                 "delete it or rebuild with make VARIANT=...")
            buf = C.create_string_buffer(512)
            handle.socmx_capabilities(buf, 512)
            assert ("static_hdims=%d,%d,%d" % hp).encode() in buf.value, buf.value
    _variants[hp] = handle if handle is not None else base
    return _variants[hp]


_STATUS_TEXT = {-1: "null pointer", -2: "bad dimension", -3: "unknown problem kind", -4: "workspace too small",
                -5: "does not fit in 160 KiB of LDS (d too large for this setting / these hidden widths)"}


def check(status, what):
    if status != 0:
        kind = _STATUS_TEXT.get(status, "invalid argument") if status < 0 else "hipError_t"
        raise SocmxError(f"{what} failed with status {status} ({kind})")


def ptr(t):
    """Device address of a contiguous fp32 CUDA tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous(), (t.device, t.dtype, t.is_contiguous())
    return t.data_ptr()


class on_device:
    """Device guard for a native call: kernels launch on the CURRENT HIP device, so the device that owns the buffers is
    made current for the duration of the call (a no-op when it already is -- the common case costs one comparison)."""

    __slots__ = ("dev", "prev")

    def __init__(self, dev):
        self.dev = dev.index if dev.index is not None else torch.cuda.current_device()

    def __enter__(self):
        self.prev = torch.cuda.current_device()
        if self.prev != self.dev:
            torch.cuda.set_device(self.dev)

    def __exit__(self, *exc):
        if self.prev != self.dev:
            torch.cuda.set_device(self.prev)


def stream_ptr(device=None):
    return torch.cuda.current_stream(device).cuda_stream


_objective_ws = {}


def objective_workspace(device, K, B):
    """The workspace of the objective's ordered sum (include/socmx.h: socmx_socm_objective_workspace_floats) for launches on
    the CURRENT stream of `device`: zeroed once, re-armed by every launch, one per (device, stream, size) -- launches of one
    stream never overlap, and a buffer is never re-allocated (a captured graph keeps using the address it recorded)."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    n = int(lib().socmx_socm_objective_workspace_floats(int(K), int(B)))
    key = (idx, torch.cuda.current_stream(dev).cuda_stream, n)
    ws = _objective_ws.get(key)
    if ws is None:
        ws = torch.zeros(n, dtype=torch.float32, device=torch.device("cuda", idx))
        if torch.cuda.is_current_stream_capturing():
            return ws            # (memory of the graph's own pool: lives and dies with that graph, never cached)
        _objective_ws[key] = ws
    return ws


def i3(v):
    return _I3(*[int(x) for x in v])


def i2(v):
    return _I2(*[int(x) for x in v])
