"""Data-parallel glue for the training step: one process per GPU, scan pairs sharded
across ranks (pairs are independent: SURVEY.md §8e), and the only data-path collective
of a step: the sum all-reduce of the fp32 parameter gradients (1 769 905 elements =
7.08 MB) over ``torch.distributed`` (backend "nccl" = RCCL over xGMI on ROCm; "gloo" on
CPU for tests).  The reference has no multi-GPU path; nothing is mirrored here.

All parameter gradients are views into one contiguous buffer -- the one the U-Net
backward itself writes (mmk_unet_backward's gradient block, adopted by autograd as the
.grad tensors) -- so the collective needs no packing copies.  Two forms:

  * ``overlap=False``: ONE all-reduce of the whole block between backward and step;
  * ``overlap=True`` (BASELINE.json configs[3]: "grad all-reduce overlapped with ... bwd"):
    the block is reduced in the three contiguous buckets in which the native backward
    completes it (decoder + final layer, encoder blocks 3-5, encoder blocks 0-2:
    include/mmk.h, mmk_unet_backward_buckets).  The backward records one event per
    bucket; each bucket's all-reduce is enqueued on a communication stream that waits
    for that event only, so the first two (99 % of the bytes) run beside the rest of the
    backward pass and the step waits for the last, 42 KB one.  Same sums as the single
    all-reduce (an all-reduce is element-wise), bit for bit.
"""
import torch
import torch.distributed as dist


class FlatGradSync:
    def __init__(self, module, process_group=None, overlap=False, force_collective=False):
        """``force_collective``: issue the collectives even in a one-rank group (the single-GPU rehearsal of the N-rank path:
        bench.py --force-dist)."""
        self.pg = process_group
        self.overlap = bool(overlap)
        self.force = bool(force_collective)
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
        # events recorded around it on the stream it is ordered on
        self.timing = None
        self.exposed = None         # when a list: (start, end) events on the step's stream around its wait for the collectives
        self.calls = 0
        self.buckets_last = None    # [(first element, one past the last)] of the last call, for tests
        self._comm = None
        self._events = None
        self._armed_at = -1

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
        if self.world_size() > 1 or (self.force and dist.is_initialized()):
            for p in self.params:
                dist.broadcast(p.data, src=src, group=self.pg)

    def zero_grad(self):
        """Before the backward pass.  The gradients are dropped rather than zero-filled: the hand-written U-Net
        backward returns all 46 gradients as views of ONE fresh contiguous buffer, which autograd adopts as the
        ``.grad`` tensors when none exist -- and which ``__call__`` then all-reduces in place (no packing copies, no
        46 accumulate launches).  Gradients produced any other way are packed into the bucket by ``__call__``.
        With ``overlap`` this also arms the native backward's bucket events for the coming pass."""
        for p in self.params:
            p.grad = None
        self._arm()

    def _arm(self):
        from . import unet_hip
        if not (self.overlap and self.params[0].is_cuda):
            unet_hip.GRAD_BUCKET_EVENTS = None
            return
        if self._events is None:
            dev = self.params[0].device
            self._comm = torch.cuda.Stream(device=dev)
            self._events = [torch.cuda.Event() for _ in unet_hip.GRAD_BUCKETS]
            for e in self._events:          # a torch event gets its HIP handle at its first record
                e.record(torch.cuda.current_stream(dev))
        unet_hip.GRAD_BUCKET_EVENTS = self._events
        self._armed_at = unet_hip.GRAD_BUCKET_PASSES[0]

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

    def bucket_ranges(self):
        """Element ranges [lo, hi) of the flat buffer, one per bucket in completion order; together they cover the buffer
        (alignment gaps included).  Needs the 46-parameter layout of the mask U-Net; anything else is one bucket."""
        from . import unet_hip
        n = self.flat.numel()
        if len(self.params) != sum(c for _, c in unet_hip.GRAD_BUCKETS):
            return [(0, n)]
        base = self.flat.data_ptr()
        starts = sorted((self.views[f].data_ptr() - base) // 4 for f, _ in unet_hip.GRAD_BUCKETS)
        assert starts[0] == 0
        ends = {s: (starts[i + 1] if i + 1 < len(starts) else n) for i, s in enumerate(starts)}
        out = []
        for f, _ in unet_hip.GRAD_BUCKETS:
            lo = (self.views[f].data_ptr() - base) // 4
            out.append((int(lo), int(ends[lo])))
        return out

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
        if not (ws > 1 or (self.force and dist.is_initialized())):
            return
        timed = self.timing is not None and self.flat.is_cuda
        if not self.overlap:
            ev = None
            if timed:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.pg)
            if ev is not None:
                ev[1].record()
                self.timing.append(ev)
            self.calls += 1
            self.buckets_last = [(0, int(self.flat.numel()))]
        else:
            ranges = self.bucket_ranges()
            self.buckets_last = ranges
            from . import unet_hip
            # the events count only if exactly this step's backward recorded them
            armed = (base is not None and self.flat.is_cuda and self._events is not None and unet_hip.GRAD_BUCKET_EVENTS is self._events
                     and unet_hip.GRAD_BUCKET_PASSES[0] == self._armed_at + 1)
            if self.flat.is_cuda:
                cur = torch.cuda.current_stream(self.flat.device)
                comm = self._comm or torch.cuda.Stream(device=self.flat.device)
                self._comm = comm
                if not armed:
                    comm.wait_stream(cur)           # gradients from somewhere else: everything enqueued so far
                works = []
                for b, (lo, hi) in enumerate(ranges):
                    if armed:
                        comm.wait_event(self._events[b])
                    with torch.cuda.stream(comm):
                        ev = None
                        if timed:
                            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                            ev[0].record(comm)
                        works.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
                        works[-1].wait()            # orders the communication stream behind the collective (no host wait)
                        if ev is not None:
                            ev[1].record(comm)
                            self.timing.append(ev)
                    self.calls += 1
                ex = None
                if self.exposed is not None:
                    ex = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                    ex[0].record(cur)
                cur.wait_stream(comm)               # the step continues behind the last bucket
                if ex is not None:
                    ex[1].record(cur)
                    self.exposed.append(ex)
                self.flat.record_stream(comm)
            else:
                for lo, hi in ranges:
                    dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.pg)
                    self.calls += 1
        if ws > 1:
            self.flat.div_(ws)

    def allreduce_bytes(self):
        """Bytes the gradient all-reduce(s) of one step move per rank (the flat fp32 buffer)."""
        return int(self.flat.numel()) * 4


def shard_indices(global_batch, rank, world_size, start=0):
    """Pair indices of this rank: rank r takes pairs r::world_size of the global batch."""
    assert global_batch % world_size == 0, "global batch must divide by the number of ranks"
    return [start + rank + j * world_size for j in range(global_batch // world_size)]
