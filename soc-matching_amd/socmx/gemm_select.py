"""Library-GEMM selection for the torch side of the SOCM iteration.

nabla_V on the (K+1)*B trajectory rows and the pair-grid network are plain `addmm`/`mm`/`bmm` calls with
unusual shapes (25,728 x 256 x 128, 20,301 x 128 x 100, split-K weight gradients ...).  The default
rocBLAS/hipBLASLt heuristic is not the fastest solution for several of them on gfx950; PyTorch's TunableOp
times the candidate solutions once per shape and remembers the winner.  `enable()` switches that on, preloads
the selections measured on an MI355X for the BASELINE configurations (data/gemm_select_gfx950.csv, ignored
by torch when its library-version validators do not match) and keeps new results in a per-user temp file.
Worth ~2.5 % of the cfg3 iteration; numerics stay fp32 (parity tests run with it on as well).
"""
import os
import tempfile

import torch

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "gemm_select_gfx950.csv")
_state = {"on": False}


def enable(tune_new_shapes=True, cache_dir=None):
    """Returns True when selection is active (GPU present and torch.cuda.tunable available)."""
    if _state["on"]:
        return True
    if os.environ.get("SOCMX_NO_GEMM_SELECT") or not torch.cuda.is_available():
        return False
    try:
        import torch.cuda.tunable as T
        T.enable(True)
        T.tuning_enable(bool(tune_new_shapes))
        T.set_max_tuning_duration(15)        # ms per candidate solution
        T.set_max_tuning_iterations(30)
        path = os.path.join(cache_dir or tempfile.gettempdir(), f"socmx_gemm_select_{os.getuid()}.csv")
        T.set_filename(path, insert_device_ordinal=True)
        if os.path.exists(_DATA):
            T.read_file(_DATA)
    except Exception as e:  # an old torch without TunableOp: run with the library's default heuristic
        import warnings
        warnings.warn(f"socmx: GEMM selection unavailable ({e}); using the library default")
        return False
    _state["on"] = True
    return True


def disable():
    if _state["on"]:
        import torch.cuda.tunable as T
        T.enable(False)
        _state["on"] = False
