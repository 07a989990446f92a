"""Data-parallel glue for the training step: one process per GPU, scan pairs sharded
across ranks (pairs are independent: SURVEY.md §8e), and ONE collective per step — a
sum all-reduce of the flat fp32 gradient (1 769 905 elements = 7.08 MB) over
``torch.distributed`` (backend "nccl" = RCCL over xGMI on ROCm; "gloo" on CPU for
tests).  The reference has no multi-GPU path; nothing is mirrored here.

All parameter gradients are views into one contiguous buffer, so the collective
needs no packing copies and a single launch: at 7 MB the ring is latency-bound
(~0.1 ms against a step of tens of ms), so bucketing/overlap would buy nothing.
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
        self._attach()
        self.flat.zero_()

    def __call__(self):
        """Average the gradients over the ranks (call between backward and step)."""
        self._attach()
        ws = self.world_size()
        if ws > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.pg)
            self.flat.div_(ws)


def shard_indices(global_batch, rank, world_size, start=0):
    """Pair indices of this rank: rank r takes pairs r::world_size of the global batch."""
    assert global_batch % world_size == 0, "global batch must divide by the number of ranks"
    return [start + rank + j * world_size for j in range(global_batch // world_size)]
