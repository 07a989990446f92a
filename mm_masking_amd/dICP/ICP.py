"""``dICP.ICP.ICP`` — batched differentiable ICP, MI355X implementation.

Keeps the call contract the reference relies on
(mm_masking/icp_weight_policy.py:54-55,281-288; mm_masking/icp_weight_dataset.py:59-61):

    ICP(icp_type='pt2pt'|'pt2pl', config_path=<yaml>, differentiable=True|False,
        max_iterations=int, tolerance=float)
    .target_pad_val
    .icp(source (B,N,3), target (B,M,6|3), T_init=(B,4,4), weight=(B,N),
         trim_dist=5.0, loss_fn={"name": "cauchy"|"huber", "metric": k}, dim=2|3)
        -> {"T": (B,4,4)}

The arithmetic (nearest neighbour, trim + robust weights, Jacobian accumulation,
Gauss-Newton solve, SE(2)/SE(3) update, and the reverse sweep that yields
dL/dweight) runs in the HIP kernels of csrc/mmk_icp.hip through the C ABI
(include/mmk.h); this file only owns buffers and the autograd hook.  The spec
is DESIGN.md §3; upstream dICP's source is absent, so it is OUR spec — see
"parity unpinned" there.  ``T`` maps source -> target (p_t ~ T p_s) and is
iterated from ``T_init``.
"""
import ctypes
import os

import torch
import yaml

from .. import _lib

_DEFAULT_CFG = os.path.join(os.path.dirname(os.path.abspath(__file__)), "config", "dICP_config.yaml")

_ws_cache = {}


def _workspace(nbytes, device):
    """Scratch of the ICP kernels, one buffer per (device, stream): calls enqueued on different streams
    (e.g. micro-batches overlapped on side streams) must not share scratch."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def _state_buffers(p, device):
    K, B, N = p.max_iter, p.B, p.N
    return {
        "idx": torch.empty((K if p.save_state else 1, B, N), dtype=torch.int32, device=device),
        "T": torch.empty((K + 1, B, 16), dtype=torch.float32, device=device),
        "delta": torch.empty((K, B, 6), dtype=torch.float64, device=device),
        "A": torch.empty((K, B, 36), dtype=torch.float64, device=device),
        "active": torch.empty((K + 1, B), dtype=torch.int32, device=device),
    }


ICP_STATUS_UNARMED_KEY = 1          # include/mmk.h: MMK_ICP_STATUS_UNARMED_KEY


class _StatusWatch:
    """Device-side error flags of the ICP kernels, checked WITHOUT synchronising the step: every forward call enqueues a
    4-byte copy of its status word into a pinned host slot behind its kernels (mmk_icp_status) plus an event; the slots
    whose event has completed are examined at the next ICP call (and by ``check(wait=True)``).  A raised flag is an
    internal error of the library (a nearest-neighbour key that no search kernel wrote) and becomes an MmkError instead
    of a silently clamped correspondence."""

    def __init__(self):
        self.pending = []           # (event, pinned int32 tensor)
        self.free = []

    def watch(self, p, ws, dev):
        L = _lib.lib()
        slot = self.free.pop() if self.free else torch.zeros(1, dtype=torch.int32).pin_memory()
        _lib.check(L.mmk_icp_status(ctypes.byref(p), _lib.ptr(ws), ws.numel(), ctypes.c_void_p(slot.data_ptr()),
                                    _lib.stream_ptr(dev)))
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        self.pending.append((ev, slot))

    def check(self, wait=False):
        keep = []
        bad = 0
        for ev, slot in self.pending:
            if wait:
                ev.synchronize()
            if ev.query():
                bad |= int(slot[0])
                self.free.append(slot)
            else:
                keep.append((ev, slot))
        self.pending = keep
        if bad & ICP_STATUS_UNARMED_KEY:
            raise _lib.MmkError("dICP: a source row reached the accumulation stage with a nearest-neighbour key that no search "
                                "kernel wrote (MMK_ICP_STATUS_UNARMED_KEY) -- internal error, the poses of that call are invalid")
        if bad:
            raise _lib.MmkError("dICP: the ICP kernels raised status bits 0x%x" % bad)


_status = _StatusWatch()


def check_errors(wait=True):
    """Raise if any ICP call so far raised a device-side error flag (``wait``: first wait for the calls in flight)."""
    _status.check(wait=wait)


def _forward(p, src, tgt, weight, T_init):
    L = _lib.lib()
    dev = src.device
    _status.check()
    ws = _workspace(L.mmk_icp_workspace_bytes(ctypes.byref(p)), dev)
    st = _state_buffers(p, dev)
    T_out = torch.empty(p.B, 4, 4, dtype=torch.float32, device=dev)
    iters = ctypes.c_int(0)
    _lib.check(L.mmk_icp_forward(ctypes.byref(p), _lib.ptr(src, torch.float32, "source"),
                                 _lib.ptr(tgt, torch.float32, "target"), _lib.ptr(weight, torch.float32, "weight"),
                                 _lib.ptr(T_init, torch.float32, "T_init"), _lib.ptr(T_out), _lib.ptr(st["idx"]),
                                 _lib.ptr(st["T"]), _lib.ptr(st["delta"]), _lib.ptr(st["A"]), _lib.ptr(st["active"]),
                                 _lib.ptr(ws), ws.numel(), ctypes.byref(iters), _lib.stream_ptr(dev)))
    _status.watch(p, ws, dev)
    return T_out, st, iters.value


class _IcpFunction(torch.autograd.Function):
    """Autograd hook: forward = K unrolled iterations in HIP, backward = the
    hand-derived reverse sweep (no autograd graph through the iterations)."""
    last_active = None

    @staticmethod
    def forward(ctx, weight, T_init, src, tgt, p):
        T_out, st, _ = _forward(p, src, tgt, weight, T_init)
        _IcpFunction.last_active = st["active"]      # (K+1,B): pairs still iterating (measurement: bench.py)
        ctx.p = p
        ctx.save_for_backward(weight, src, tgt, st["idx"], st["T"], st["delta"], st["A"], st["active"])
        return T_out

    @staticmethod
    def backward(ctx, grad_T):
        weight, src, tgt, idx, T_hist, delta, A, active = ctx.saved_tensors
        p = ctx.p
        L = _lib.lib()
        dev = src.device
        ws = _workspace(L.mmk_icp_workspace_bytes(ctypes.byref(p)), dev)
        gT = grad_T.contiguous().float()
        gw = torch.empty(p.B, p.N, dtype=torch.float32, device=dev)
        gT0 = torch.empty(p.B, 4, 4, dtype=torch.float32, device=dev)
        _lib.check(L.mmk_icp_backward(ctypes.byref(p), _lib.ptr(src), _lib.ptr(tgt), _lib.ptr(weight), _lib.ptr(idx),
                                      _lib.ptr(T_hist), _lib.ptr(delta), _lib.ptr(A), _lib.ptr(active), _lib.ptr(gT),
                                      _lib.ptr(gw), _lib.ptr(gT0), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev)))
        return gw, gT0, None, None, None


class ICP:
    NN_SEARCH_OVERRIDE = None      # tests / benchmarks: force "brute" or "grid" for every instance

    def __init__(self, icp_type="pt2pl", config_path=None, differentiable=True, max_iterations=100, tolerance=1e-12):
        if icp_type not in _lib.ICP_TYPES:
            raise ValueError("icp_type must be 'pt2pt' or 'pt2pl' (got %r)" % (icp_type,))
        self.icp_type = icp_type
        self.differentiable = bool(differentiable)
        self.max_iterations = int(max_iterations)
        self.tolerance = float(tolerance)
        # The reference's path is CWD-relative and the upstream YAML is absent:
        # fall back to the built-in defaults instead of raising.
        cfg = {}
        for path in (config_path, _DEFAULT_CFG):
            if path and os.path.isfile(path):
                with open(path, "r") as f:
                    cfg = (yaml.safe_load(f) or {}).get("dICP", {}) or {}
                break
        self.config_path = config_path
        self.target_pad_val = float(cfg.get("target_pad_val", 1000.0))
        self.check_every = int((cfg.get("parameters") or {}).get("check_every", 8))
        # "brute": the exhaustive LDS-tiled scan; "grid": exact search through a uniform grid
        # (identical correspondences, ~100x fewer distance evaluations)
        self.nn_search = ICP.NN_SEARCH_OVERRIDE or str((cfg.get("parameters") or {}).get("nn_search", "brute"))
        if self.nn_search not in _lib.NN_METHODS:
            raise ValueError("dICP config: nn_search must be 'brute' or 'grid' (got %r)" % (self.nn_search,))
        self.last_state = None
        self.last_iterations = None

    def _params(self, B, N, M, tgt_cols, dim, loss_fn, trim_dist, save_state):
        name = None if loss_fn is None else loss_fn.get("name")
        if name not in _lib.LOSSES:
            raise ValueError("unknown loss_fn name %r" % (name,))
        return _lib.IcpParams(B=B, N=N, M=M, tgt_cols=tgt_cols, dim=dim, icp_type=_lib.ICP_TYPES[self.icp_type],
                              loss=_lib.LOSSES[name], loss_k=float(1.0 if loss_fn is None else loss_fn.get("metric", 1.0)),
                              trim_dist=float(trim_dist), tolerance=self.tolerance, max_iter=self.max_iterations,
                              save_state=1 if save_state else 0,
                              check_every=0 if save_state else self.check_every,
                              nn_method=_lib.NN_METHODS[self.nn_search])

    def icp(self, source, target, T_init=None, weight=None, trim_dist=5.0, loss_fn=None, dim=3):
        if source.ndim != 3 or source.shape[-1] != 3:
            raise ValueError("source must be (B,N,3)")
        if target.ndim != 3 or target.shape[-1] not in (3, 6):
            raise ValueError("target must be (B,M,3) or (B,M,6)")
        if self.icp_type == "pt2pl" and target.shape[-1] != 6:
            raise ValueError("pt2pl needs target normals: target must be (B,M,6)")
        if dim not in (2, 3):
            raise ValueError("dim must be 2 or 3")
        if source.is_cuda:
            dev = source.device
        elif torch.cuda.is_available():
            dev = torch.device("cuda", torch.cuda.current_device())
        else:
            raise _lib.MmkError("dICP.ICP needs an MI355X/HIP device: the ICP is a set of HIP kernels with no CPU path")
        out_device = source.device
        src = _lib.dev_f32(source, dev)
        tgt = _lib.dev_f32(target, dev)
        B, N, _ = src.shape
        M = tgt.shape[1]
        if T_init is None:
            T0 = torch.eye(4, dtype=torch.float32, device=dev).repeat(B, 1, 1)
        else:
            T0 = T_init.to(device=dev, dtype=torch.float32).contiguous()
        w = None
        if weight is not None:
            w = weight.to(device=dev, dtype=torch.float32).contiguous()
        need_grad = self.differentiable and torch.is_grad_enabled() and (
            (w is not None and w.requires_grad) or T0.requires_grad)
        p = self._params(B, N, M, tgt.shape[-1], dim, loss_fn, trim_dist, save_state=need_grad)
        if need_grad:
            if w is None:
                w = torch.ones(B, N, dtype=torch.float32, device=dev)
            T = _IcpFunction.apply(w, T0, src, tgt, p)
            self.last_iterations = p.max_iter
        else:
            with torch.no_grad():
                T, st, iters = _forward(p, src, tgt, None if w is None else w.detach(), T0.detach())
            self.last_state = st
            self.last_iterations = iters
        return {"T": T if T.device == out_device else T.to(out_device)}
