"""Batch-dimension data parallelism over the GPUs of one node (no reference counterpart: the
reference is single-process).

Trajectories are independent (utils.py:37-101 has no cross-sample term) and the SOCM objective is a
sum over samples (method.py:717-720), so rank r simulates rows [row0, row0+B_r) of the global batch
with the Philox stream keyed by the GLOBAL row index, and the only communication per iteration is
ONE all_reduce(SUM) (stopping-time SOCM: one more, a scalar -- the loss normaliser sum(stop_indicators) scales every
gradient and is needed before the backward pass, solver._stop_normaliser) of a flat fp32 buffer holding every gradient, the objective value, the shifted
sums (sum (w - c), sum (w - c)^2, n) that give mean/std of w (c = the running normalisation constant,
identical on every rank) and, when computed, this rank's share of the weighted L2 error --
over RCCL (torch.distributed backend "nccl") on xGMI; "gloo" on CPU for tests.  The buffer is
0.7-3 MB, i.e. latency-bound: it must stay a single collective, never one call per parameter.
(hipGraph mode, train.py: the pair-grid network's gradients -- produced one iteration later on the
second stream, beside the next rollout -- travel in a second, smaller all_reduce inside the same
captured graph.  `SOC_Solver.loss` called directly on a sharded solver still pools the weight
statistics itself, with an all_gather + Chan's rule.)
"""
import torch
import torch.distributed as dist

from . import loss as L


class Shard:
    def __init__(self, rank=None, world_size=None, group=None):
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world_size = dist.get_world_size(group) if world_size is None else world_size
        self._flat = {}          # one flat buffer per call site ("slot"): collectives of different streams never share one

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
        gathered = [torch.empty_like(stats) for _ in range(self.world_size)]
        dist.all_gather(gathered, stats, group=self.group)
        return L.combine_stats(torch.stack(gathered))

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
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
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
        dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        return bool(flag.item())

    def allreduce_flat_(self, flat):
        """In-place all_reduce(SUM) of a caller-owned flat fp32 buffer (the hipGraph body owns its buffers: gradients,
        objective and weight statistics are written straight into one)."""
        if self.world_size == 1 and not dist.is_initialized():
            return flat
        if self.world_size == 1 and flat.is_cuda and torch.cuda.is_current_stream_capturing():
            # One rank: the sum is the identity -- and it is NOT put into a graph under capture.  A captured RCCL call leaves an
            # event "last recorded in a capturing stream" behind, and torch's process-group watchdog thread, which polls its
            # events on its own schedule, dies on it (hipErrorCapturedEvent -> the whole process): seen in one of three full runs
            # of the GPU test suite.  Eager calls at world size 1 still go through RCCL (the path a multi-rank run takes by default).
            return flat
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        return flat


def shifted_weight_sums(weight, shift):
    """(sum (w - c), sum (w - c)^2, n) of this shard's importance weights: summable across ranks (one all_reduce), and
    well conditioned when c is close to mean(w) -- Trainer passes the running normalisation constant."""
    wc = weight.double() - shift.double() if torch.is_tensor(shift) else weight.double() - float(shift)
    # (accumulated in fp64, handed to the fp32 all-reduce buffer rounded once)
    return torch.stack([wc.sum(), (wc * wc).sum(), torch.tensor(float(weight.numel()), device=weight.device,
                                                                dtype=torch.float64)]).to(torch.float32)


def mean_std_from_shifted_sums(sums, shift):
    """Pooled mean / unbiased std from the reduced sums.  The subtraction S2 - S1^2 / n runs in fp64; what remains is the
    fp32 rounding of the TRANSMITTED sums, a relative error of about 6e-8 (1 + ((mean - c) / std)^2) in the variance --
    below 1e-5 while the running normaliser c is within ten standard deviations of the batch mean."""
    s1, s2, n = sums[0].double(), sums[1].double(), sums[2].double()
    sh = shift.double() if torch.is_tensor(shift) else float(shift)
    mean = sh + s1 / n
    std = torch.sqrt(torch.clamp(s2 - s1 * s1 / n, min=0.0) / (n - 1))
    return mean.to(torch.float32), std.to(torch.float32)
