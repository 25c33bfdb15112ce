"""The C-ABI library loads without a GPU and exports exactly what include/socmx.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "socmx.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(socmx_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    names = declared_symbols()
    for must in ("socmx_rollout_f32", "socmx_unet_pack_f32", "socmx_unet_forward_f32", "socmx_weights_stats_f32",
                 "socmx_socm_prep_f32", "socmx_socm_target_fwd_f32", "socmx_socm_target_bwd_f32", "socmx_version"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from socmx import _lib
    assert os.path.exists(_lib.LIB_PATH), "build first: make -C soc-matching_amd/csrc"
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(handle, name), f"{name} declared in socmx.h but not exported"
    # and the Python binding knows every one of them (no stale prototype table)
    assert sorted(_lib.PROTOTYPES) == declared_symbols()


def declared_parameter_counts():
    """name -> number of parameters of its prototype in socmx.h (`void` = 0)."""
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = {}
    for name, params in re.findall(r"\b(socmx_[a-z0-9_]+)\s*\(([^()]*)\)\s*;", text):
        params = params.strip()
        out[name] = 0 if params in ("", "void") else params.count(",") + 1
    return out


def test_binding_prototypes_have_the_declared_parameter_counts():
    """A ctypes prototype one argument short still 'works' until a 64-bit argument lands in the untyped tail (the stream
    pointer is the last parameter of every launcher): compare the arities of the binding table with the header."""
    from socmx import _lib
    counts = declared_parameter_counts()
    assert sorted(counts) == declared_symbols()
    for name, (_, argtypes) in _lib.PROTOTYPES.items():
        assert len(argtypes) == counts[name], f"{name}: binding has {len(argtypes)} parameters, socmx.h declares {counts[name]}"


def test_argument_validation_needs_no_gpu():
    from socmx import _lib
    L = _lib.lib()
    assert L.socmx_version() == 149
    assert L.socmx_num_pairs(200) == 201 * 202 // 2
    assert L.socmx_unet_packed_floats(10, _lib.i3([256, 128, 64])) > 170562      # padded image >= parameter count
    assert L.socmx_unet_packed_floats(0, _lib.i3([256, 128, 64])) == 0
    # NULL / bad-dimension arguments are rejected before any HIP call
    assert L.socmx_weights_stats_f32(None, None, None, 4, None, None, None) == -1
    assert L.socmx_socm_target_bwd_f32(0, 4, 4, 1, 1, 1, 1, 1, 1, None) == -2
    pb = _lib.Problem(kind=7, d=2, sigma=1)
    assert L.socmx_rollout_f32(pb, 1, _lib.i3([8, 8, 8]), 1, 1, 4, 4, 1.0, 0, 0, 0, None,
                               1, 1, 1, 1, 1, 1, 1, 1, None) == -3


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from socmx import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libsocmx.so"))
    with pytest.raises(_lib.SocmxError):
        _lib.lib()


def test_capabilities_list_the_supported_ranges():
    from socmx import _lib
    L = _lib.lib()
    buf = ctypes.create_string_buffer(2048)
    need = L.socmx_capabilities(buf, 2048)
    text = buf.value.decode()
    assert need <= 2048 and "gfx950" in text and "static_hdims=256,128,64" in text
    for piece in ("ranges:", "stopping-time SOCM kernels: d <= 16", "pair-grid network:", "control-network backward:",
                  "one row per workgroup"):
        assert piece in text, piece


def test_variant_library_falls_back_when_it_cannot_be_built(monkeypatch):
    """A host without hipcc / make (or a read-only package directory) must not turn `backend.specialize_arch: True` into a
    crash, or into ranks waiting for each other: the build failure is reported and the default library's descriptor-driven
    kernels serve the architecture."""
    from socmx import _lib

    def broken(hdims, quiet=False):
        raise _lib.SocmxError("building the architecture variant failed: make: hipcc: No such file or directory")

    monkeypatch.setattr(_lib, "build_variant", broken)
    monkeypatch.setattr(_lib, "_variants", {})
    hd = [96, 48, 24]
    assert not os.path.exists(_lib.variant_path(hd))
    with pytest.warns(UserWarning, match="could not build the kernel variant"):
        got = _lib.variant(hd, build=True)
    assert got is _lib.lib()
    assert _lib.variant(hd) is _lib.lib()                 # cached: no second attempt, no second warning


def test_variant_library_loads_in_concurrent_processes():
    """Several processes asking for the same prebuilt variant at once (the jobs of a sweep, the ranks of a node) serialise on
    the lock file and each loads the complete library."""
    import subprocess
    import sys
    from socmx import _lib
    path = _lib.variant_path([128, 64, 32])
    if not os.path.exists(path):
        pytest.skip("libsocmx_128_64_32.so not built")
    code = ("import sys; sys.path.insert(0, %r); from socmx import _lib; h = _lib.variant([128, 64, 32], build=False); "
            "assert h is not _lib.lib() and h.socmx_version() == _lib.lib().socmx_version(); print('ok')"
            % os.path.join(ROOT, "soc-matching_amd"))
    procs = [subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for _ in range(4)]
    for p in procs:
        out, err = p.communicate(timeout=300)
        assert p.returncode == 0 and "ok" in out, err[-1500:]


def test_rccl_binding_loads_torchs_library_and_exports_what_it_binds():
    """socmx/rccl.py (the shard's own communicators) binds the librccl.so torch itself is linked against -- one RCCL per
    process -- and every entry point it declares is there (no communicator is created without a GPU: symbols and version only)."""
    import torch
    from socmx import rccl
    path = rccl.library_path()
    assert os.path.exists(path) and os.path.dirname(path) == os.path.join(os.path.dirname(torch.__file__), "lib"), path
    L = rccl.lib()
    for name in ("ncclGetVersion", "ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclCommGetAsyncError",
                 "ncclAllReduce", "ncclAllGather", "ncclGetErrorString"):
        assert hasattr(L, name), name
    v = rccl.version()
    assert v >= 21800, v                       # NCCL_VERSION_CODE: 2.18 or newer (graph capture of collectives)
    assert rccl._DTYPES[torch.float32] == 7 and rccl._OPS["sum"] == 0 and rccl._OPS["min"] == 3      # nccl.h enums
    assert b"unhandled" in L.ncclGetErrorString(1).lower() or L.ncclGetErrorString(1)            # a string comes back
