"""A per-rank RCCL communicator owned by this package (ctypes on the librccl.so that torch itself has loaded).

Why not `torch.distributed.all_reduce` for the iteration's collectives: ProcessGroupNCCL wraps every call in a `Work`
with start / end events on a stream of its own pool and hands it to a watchdog thread that polls those events on its own
schedule.  A stream of that pool that has joined a hipGraph capture makes such a poll answer hipErrorCapturedEvent, which
the watchdog turns into std::terminate (seen in rounds 4-5).  A collective enqueued HERE is one kernel launch on the
caller's stream -- no Work, no event, no second stream, no watchdog -- so it is captured into the iteration's graph like
any other launch and replayed with it (the same arrangement as inference stacks that replay decode steps with their
all-reduces inside the graph).

The process group is still what brings the communicator up (rank 0's ncclUniqueId travels over it) and what bench.py /
main.py use for barriers and object gathers.  No reference counterpart (the reference is single-process).

ncclResult_t ncclGetUniqueId(ncclUniqueId*);                 ncclCommInitRank(ncclComm_t*, int n, ncclUniqueId id, int rank);
ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
ncclResult_t ncclAllGather(const void* send, void* recv, size_t sendcount, ncclDataType_t, ncclComm_t, hipStream_t);
"""
import ctypes as C
import os

import torch

_DTYPES = {torch.int32: 2, torch.int64: 4, torch.float32: 7, torch.float64: 8}      # ncclInt32, ncclInt64, ncclFloat32, ncclFloat64
_OPS = {"sum": 0, "prod": 1, "max": 2, "min": 3}
_lib = None


class RcclError(RuntimeError):
    pass


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_byte * 128)]


def library_path():
    """torch's own copy (the one libtorch_hip.so is linked against: dlopen of the same path returns the same handle, so
    there is ONE RCCL in the process); SOCMX_RCCL_LIB overrides."""
    p = os.environ.get("SOCMX_RCCL_LIB")
    if p:
        return p
    p = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    return p if os.path.exists(p) else "librccl.so.1"


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(library_path())
        L.ncclGetErrorString.restype = C.c_char_p
        L.ncclGetErrorString.argtypes = [C.c_int]
        L.ncclGetVersion.argtypes = [C.POINTER(C.c_int)]
        L.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
        L.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
        L.ncclCommDestroy.argtypes = [C.c_void_p]
        L.ncclCommAbort.argtypes = [C.c_void_p]
        L.ncclCommGetAsyncError.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.ncclAllGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
        for f in ("ncclGetVersion", "ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclCommAbort",
                  "ncclCommGetAsyncError", "ncclAllReduce", "ncclAllGather"):
            getattr(L, f).restype = C.c_int
        _lib = L
    return _lib


def _check(code, what):
    if code != 0:
        raise RcclError(f"{what}: {lib().ncclGetErrorString(code).decode()} (ncclResult_t {code})")


def version():
    v = C.c_int(0)
    _check(lib().ncclGetVersion(C.byref(v)), "ncclGetVersion")
    return v.value


class Communicator:
    """One RCCL communicator over the ranks of `group` (default: the world), one rank per device.  Collectives are in place,
    on the CURRENT torch stream of `device`, asynchronous, capturable."""

    def __init__(self, device, group=None):
        import torch.distributed as dist
        self.device = torch.device(device)
        self.rank, self.world_size = dist.get_rank(group), dist.get_world_size(group)
        L = lib()
        uid = _UniqueId()
        if self.rank == 0:
            _check(L.ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
        box = [bytes(uid.internal) if self.rank == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        C.memmove(C.byref(uid), box[0], 128)
        self._comm = C.c_void_p()
        with torch.cuda.device(self.device):
            _check(L.ncclCommInitRank(C.byref(self._comm), self.world_size, uid, self.rank), "ncclCommInitRank")
        self.calls = 0           # collectives enqueued or captured through this communicator (tests count them)

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _args(self, t):
        if not (t.is_cuda and t.device == self.device and t.is_contiguous() and t.dtype in _DTYPES):
            raise RcclError(f"collective on {t.dtype} {t.device} contiguous={t.is_contiguous()}: expected a contiguous "
                            f"{sorted(str(k) for k in _DTYPES)} tensor on {self.device}")
        return C.c_void_p(t.data_ptr()), _DTYPES[t.dtype]

    def all_reduce_(self, t, op="sum"):
        p, dt = self._args(t)
        with torch.cuda.device(self.device):
            _check(lib().ncclAllReduce(p, p, t.numel(), dt, _OPS[op], self._comm, self._stream()), "ncclAllReduce")
        self.calls += 1
        return t

    def all_gather(self, t):
        """(world, *t.shape) tensor holding every rank's `t`."""
        p, dt = self._args(t)
        out = torch.empty((self.world_size,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        with torch.cuda.device(self.device):
            _check(lib().ncclAllGather(p, C.c_void_p(out.data_ptr()), t.numel(), dt, self._comm, self._stream()),
                   "ncclAllGather")
        self.calls += 1
        return out

    def check_async(self):
        """Raise if the communicator has recorded an asynchronous error (a peer gone, a transport failure)."""
        err = C.c_int(0)
        _check(lib().ncclCommGetAsyncError(self._comm, C.byref(err)), "ncclCommGetAsyncError")
        _check(err.value, "RCCL asynchronous error")

    def destroy(self):
        if self._comm:
            comm, self._comm = self._comm, C.c_void_p()
            with torch.cuda.device(self.device):
                torch.cuda.synchronize(self.device)
                _check(lib().ncclCommDestroy(comm), "ncclCommDestroy")

    def __del__(self):      # (interpreter teardown: never raise, never touch a torn-down runtime)
        self._comm = None
