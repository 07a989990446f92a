"""Data-parallel glue for the training step: one process per GPU, scan pairs sharded
across ranks (pairs are independent: SURVEY.md §8e), and ONE collective per step — a
sum all-reduce of the flat fp32 gradient (1 769 905 elements = 7.08 MB) over
``torch.distributed`` (backend "nccl" = RCCL over xGMI on ROCm; "gloo" on CPU for
tests).  The reference has no multi-GPU path; nothing is mirrored here.

All parameter gradients are views into one contiguous buffer, so the collective
needs no packing copies and a single launch: at 7 MB the ring is latency-bound
(~0.1 ms against a step of ~12 ms), so bucketing/overlap would buy nothing.  The
buffer is the one the U-Net backward itself writes (mmk_unet_backward's gradient
block, adopted by autograd as the .grad tensors): the data-parallel step adds one
all-reduce and one scale launch to the single-GPU step, nothing else.
"""
import torch
import torch.distributed as dist


class FlatGradSync:
    def __init__(self, module, process_group=None):
        self.pg = process_group
        self.params = [p for p in module.parameters() if p.requires_grad]
        total = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.views = []
        off = 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        self._attach()
        # measurement hook (bench.py's ``ddp`` block): when a list, every collective appends a (start, end) pair of
        # events recorded on the current stream around it
        self.timing = None
        self.calls = 0

    def _attach(self):
        for p, v in zip(self.params, self.views):
            if p.grad is None or p.grad.data_ptr() != v.data_ptr():
                if p.grad is not None:
                    v.copy_(p.grad)
                p.grad = v

    def world_size(self):
        return dist.get_world_size(self.pg) if dist.is_initialized() else 1

    def sync_params(self, src=0):
        """Rank ``src``'s parameters become everyone's (identical init)."""
        if self.world_size() > 1:
            for p in self.params:
                dist.broadcast(p.data, src=src, group=self.pg)

    def zero_grad(self):
        """Before the backward pass.  The gradients are dropped rather than zero-filled: the hand-written U-Net
        backward returns all 46 gradients as views of ONE fresh contiguous buffer, which autograd adopts as the
        ``.grad`` tensors when none exist -- and which ``__call__`` then all-reduces in place (no packing copies, no
        46 accumulate launches).  Gradients produced any other way are packed into the bucket by ``__call__``."""
        for p in self.params:
            p.grad = None

    def _adopt(self):
        """The one buffer all gradients are views of, if they are (the native U-Net backward's layout)."""
        g0 = self.params[0].grad
        base = g0._base if g0 is not None else None
        if base is None or base.dim() != 1 or base.dtype != torch.float32:
            return None
        lo, hi = base.data_ptr(), base.data_ptr() + base.numel() * 4
        for p in self.params:
            g = p.grad
            if g is None or g._base is not base or not g.is_contiguous() or not (lo <= g.data_ptr() and g.data_ptr() + g.numel() * 4 <= hi):
                return None
        return base

    def __call__(self):
        """Average the gradients over the ranks (call between backward and step)."""
        ws = self.world_size()
        base = self._adopt()
        if base is not None:
            self.flat, self.views = base, [p.grad for p in self.params]
        else:
            # generic path: pack whatever .grad tensors exist into the bucket and make them views of it
            for p, v in zip(self.params, self.views):
                if p.grad is None:
                    v.zero_()
                elif p.grad.data_ptr() != v.data_ptr():
                    v.copy_(p.grad)
                p.grad = v
        if ws > 1:
            ev = None
            if self.timing is not None and self.flat.is_cuda:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.pg)
            if ev is not None:
                ev[1].record()
                self.timing.append(ev)
            self.calls += 1
            self.flat.div_(ws)

    def allreduce_bytes(self):
        """Bytes one gradient all-reduce moves per rank (the flat fp32 buffer)."""
        return int(self.flat.numel()) * 4


def shard_indices(global_batch, rank, world_size, start=0):
    """Pair indices of this rank: rank r takes pairs r::world_size of the global batch."""
    assert global_batch % world_size == 0, "global batch must divide by the number of ranks"
    return [start + rank + j * world_size for j in range(global_batch // world_size)]
