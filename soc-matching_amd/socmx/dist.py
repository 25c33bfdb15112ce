"""Batch-dimension data parallelism over the GPUs of one node (no reference counterpart: the
reference is single-process).

Trajectories are independent (utils.py:37-101 has no cross-sample term) and the SOCM objective is a
sum over samples (method.py:717-720), so rank r simulates rows [row0, row0+B_r) of the global batch
with the Philox stream keyed by the GLOBAL row index, and the only communication per iteration is
ONE all_reduce(SUM) (stopping-time SOCM: one more, a scalar -- the loss normaliser sum(stop_indicators) scales every
gradient and is needed before the backward pass, solver._stop_normaliser) of a flat fp32 buffer holding every gradient, the objective value, one
(n, mean, M2) slot per rank (each rank fills its own, the sum of zeros elsewhere is exact: mean/std of w pooled by Chan's
rule at one-process accuracy) and, when computed, this rank's share of the weighted L2 error --
over RCCL (torch.distributed backend "nccl") on xGMI; "gloo" on CPU for tests.  The buffer is
0.7-3 MB, i.e. latency-bound: it must stay a single collective, never one call per parameter.
(hipGraph mode, train.py: the pair-grid network's gradients -- produced one iteration later on the
second stream, beside the next rollout -- travel in a second, smaller all_reduce inside the same
captured graph.  `SOC_Solver.loss` called directly on a sharded solver still pools the weight
statistics itself, with an all_gather + Chan's rule.)

Transports (`Shard.transport`), chosen when the Shard is built -- by every rank alike, from the process group's backend:
  "rccl"    backend "nccl" on a GPU: this package's OWN communicators (socmx/rccl.py, one per stream slot) -- a collective is one
            launch on the caller's stream, so the iteration's all-reduces are captured into its hipGraph and replayed with it;
            torch's process group (its Work objects, pooled streams and watchdog thread) carries no data-path call.
  "group"   torch.distributed calls on the process group: gloo on CPU tensors (the tests here), or "nccl" when the own
            communicator could not be brought up or SOCMX_RCCL=0 asks for it.  Never captured over several ranks.
  "staged"  backend "gloo" with CUDA tensors: the buffer crosses through host memory (synchronises the stream; cannot be captured).
            A debugging / test transport: it lets several ranks share ONE GPU (RCCL refuses two ranks on one device), which is
            how tests/test_gpu_dist.py runs the sharded HIP iteration at world sizes 2 and 3 on a one-GPU box.
"""
import os

import torch
import torch.distributed as dist

from . import loss as L


class Shard:
    def __init__(self, rank=None, world_size=None, group=None, device=None, transport=None):
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world_size = dist.get_world_size(group) if world_size is None else world_size
        self._flat = {}          # one flat buffer per call site ("slot"): collectives of different streams never share one
        self._comms = {}         # transport "rccl": one communicator per stream slot
        self.collectives = 0     # data-path collectives issued (or captured) through this object
        self.device = None
        live = rank is None and world_size is None and dist.is_available() and dist.is_initialized()
        backend = dist.get_backend(group) if live else None
        if transport is None:
            transport = "group"
            if backend == "nccl" and os.environ.get("SOCMX_RCCL", "1") != "0":
                transport = "rccl"
            elif backend == "gloo" and (device is not None and torch.device(device).type == "cuda"):
                transport = "staged"
        self.transport = transport
        self.transport_note = None
        self._backend = backend
        if transport in ("rccl", "staged"):
            self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if transport == "rccl":
            # Both communicators come up HERE, collectively (every rank constructs its Shard at the same point of the program),
            # and the ranks agree over the process group that all of them succeeded: a rank that alone fell back to the
            # process group would wait in a different collective than its peers.
            err = None
            try:
                from . import rccl
                for slot in ("main", "side"):
                    self._comms[slot] = rccl.Communicator(self.device, group)
            except Exception as e:                      # noqa: BLE001 -- reported below, the run goes on over the process group
                err = e
            flag = torch.tensor([0 if err is None else 1], dtype=torch.int32, device=self.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
            torch.cuda.synchronize(self.device)
            if int(flag.item()):
                self._comms, self.transport = {}, "group"
                self.transport_note = ("own RCCL communicator not available" + (f" ({type(err).__name__}: {err})" if err else
                                       " on another rank") + ": collectives go through torch's process group, never captured")
                import warnings
                warnings.warn("socmx.dist: " + self.transport_note)

    @property
    def capturable(self):
        """May the iteration's collectives be recorded into a hipGraph?  One rank: yes (the sums are identities or one-rank RCCL
        calls).  Several ranks: only through the own communicators."""
        return self.world_size == 1 or self.transport == "rccl"

    def _reduce(self, t, op="sum", slot="main"):
        """In-place all_reduce of a contiguous tensor on the CURRENT stream, through this shard's transport."""
        self.collectives += 1
        if self.transport == "rccl" and t.is_cuda:
            return self._comms[slot].all_reduce_(t, op)
        rop = {"sum": dist.ReduceOp.SUM, "min": dist.ReduceOp.MIN, "max": dist.ReduceOp.MAX}[op]
        if self.transport == "group" and t.is_cuda and self._backend == "gloo":
            self.transport, self.device = "staged", t.device      # (a gloo group handed a CUDA tensor: through the host from here on)
        if self.transport == "staged" and t.is_cuda:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("socmx.dist: the host-staged transport (gloo + CUDA tensors) cannot be captured into a hipGraph")
            host = t.detach().to("cpu")                 # (synchronises the current stream)
            dist.all_reduce(host, op=rop, group=self.group)
            t.copy_(host)
            return t
        dist.all_reduce(t, op=rop, group=self.group)
        return t

    def _gather(self, t, slot="main"):
        """(world, *t.shape): every rank's `t`."""
        self.collectives += 1
        if self.transport == "rccl" and t.is_cuda:
            return self._comms[slot].all_gather(t.contiguous())
        if self.transport == "group" and t.is_cuda and self._backend == "gloo":
            self.transport, self.device = "staged", t.device
        src = t.detach().to("cpu") if (self.transport == "staged" and t.is_cuda) else t
        gathered = [torch.empty_like(src) for _ in range(self.world_size)]
        dist.all_gather(gathered, src.contiguous(), group=self.group)
        return torch.stack(gathered).to(t.device)

    def close(self):
        for c in self._comms.values():
            c.destroy()
        self._comms = {}

    def local_rows(self, B_global):
        """(B_local, row0): contiguous split, first (B % G) ranks take one extra row."""
        base, rem = divmod(int(B_global), self.world_size)
        B_local = base + (1 if self.rank < rem else 0)
        row0 = self.rank * base + min(self.rank, rem)
        if B_local < 1:
            raise ValueError(f"batch {B_global} smaller than world size {self.world_size}")
        return B_local, row0

    def combine_weight_stats(self, stats):
        if self.world_size == 1 and not dist.is_initialized():
            return stats
        stats = stats[:3].contiguous()       # (sum w, M2, n); a single-shard GPU kernel also appends mean, std
        return L.combine_stats(self._gather(stats))

    def allreduce_gradients(self, params, extra=None, slot="main"):
        """Sum `.grad` of all params (and the optional tensors in `extra`, any shapes) across ranks with ONE
        collective on one flat buffer, enqueued behind the CURRENT stream.  Returns the reduced extras (same shapes)."""
        params = [p for p in params if p.grad is not None]
        extra = [e.detach().reshape(-1).to(torch.float32) for e in (extra or [])]
        if self.world_size == 1 and not dist.is_initialized():
            return extra
        n_extra = sum(e.numel() for e in extra)
        n = sum(p.grad.numel() for p in params) + n_extra
        dev = params[0].grad.device if params else extra[0].device
        flat = self._flat.get(slot)
        if flat is None or flat.numel() != n or flat.device != dev:
            flat = self._flat[slot] = torch.empty(n, dtype=torch.float32, device=dev)
        views, off = [], 0
        for p in params:
            k = p.grad.numel()
            views.append(flat[off:off + k].view_as(p.grad))
            off += k
        if views:
            torch._foreach_copy_(views, [p.grad for p in params])
        if extra:
            flat[off:] = torch.cat(extra)
        self._reduce(flat, "sum", "main")
        if views:
            torch._foreach_copy_([p.grad for p in params], views)
        red = flat[off:].clone()
        out, o = [], 0
        for e in extra:
            out.append(red[o:o + e.numel()])
            o += e.numel()
        return out

    def agree(self, ok):
        """True iff EVERY rank passes ok=True (one tiny all_reduce(MIN), executed eagerly): how the ranks decide together
        whether a captured iteration may be replayed -- a rank that alone fell back to another body would issue a different
        sequence of collectives and deadlock the job."""
        if self.world_size == 1 and not dist.is_initialized():
            return bool(ok)
        backend = dist.get_backend(self.group)
        dev = (self.device if self.device is not None and backend == "nccl" else
               torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu"))
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        self._reduce(flag, "min", "main")
        return bool(flag.item())

    def allreduce_flat_(self, flat, slot="main"):
        """In-place all_reduce(SUM) of a caller-owned flat fp32 buffer (the hipGraph body owns its buffers: gradients,
        objective and weight statistics are written straight into one) on the current stream; `slot` names the stream's
        communicator ("main": the iteration's stream, "side": the second stream's pair-grid-network branch)."""
        if self.world_size == 1 and not dist.is_initialized():
            return flat
        if (self.transport != "rccl" and self.world_size == 1 and flat.is_cuda and torch.cuda.is_current_stream_capturing()):
            # One rank over torch's process group: the sum is the identity -- and it is NOT put into a graph under capture.  A
            # ProcessGroupNCCL call under capture pulls one of the group's pooled streams into the capture, and the group's watchdog
            # thread, which polls the events of earlier calls on its own schedule, dies on one recorded there
            # (hipErrorCapturedEvent -> the whole process: rounds 4-5).  The own communicators have no such thread: transport
            # "rccl" captures the call at every world size.
            return flat
        return self._reduce(flat, "sum", slot)


def weight_stat_slots(weight, rank, world_size):
    """(3 world,) fp32: this rank's (n, mean, M2 = sum (w - mean)^2) in ITS slot, zeros in every other rank's.  Summing zeros is
    exact, so ONE all_reduce(SUM) hands every rank all the triples bit for bit -- an all-gather riding inside the iteration's flat
    gradient all-reduce -- and `mean_std_from_slots` pools them with Chan's rule in fp64: the accuracy of the one-process
    statistics whatever the weights' scale.  (Rounds 4-5 summed (w - c), (w - c)^2 against the running normaliser c, which
    cancelled in fp32 when c was far from mean(w).)"""
    w = weight.detach().double()
    mean = w.mean()
    slots = torch.zeros(3 * world_size, dtype=torch.float32, device=weight.device)
    slots[3 * rank:3 * rank + 3] = torch.stack([torch.tensor(float(w.numel()), dtype=torch.float64, device=w.device), mean,
                                               ((w - mean) ** 2).sum()]).to(torch.float32)
    return slots


def mean_std_from_slots(slots):
    """Pooled mean / unbiased std (method.py:903-904 over the GLOBAL batch) from the reduced slots."""
    t = slots.double().reshape(-1, 3)
    n, mean_r, m2_r = t[:, 0], t[:, 1], t[:, 2]
    N = n.sum()
    mean = (n * mean_r).sum() / N
    m2 = m2_r.sum() + (n * (mean_r - mean) ** 2).sum()
    return mean.to(torch.float32), torch.sqrt(m2 / (N - 1)).to(torch.float32)
