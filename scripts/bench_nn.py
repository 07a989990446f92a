"""Micro-benchmark of the stand-alone NN kernel at the bench shape (unseeded: no previous correspondents)."""
import ctypes, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mm_masking_amd import _lib
from oracle import _clib
L = _lib.lib()
dev = torch.device("cuda:0")
B, N, M, dim = 32, 5120, 20000, int(os.environ.get("DIM", "2"))
g = torch.Generator(device="cpu").manual_seed(0)
src = (torch.rand(B, N, 3, generator=g) * 140 - 70).to(dev)
tgt = (torch.rand(B, M, 6, generator=g) * 140 - 70).to(dev)
T = torch.eye(4).repeat(B, 1, 1).reshape(B, 16).contiguous().to(dev)
Mpad = L.mmk_nn_padded_m(M)
planar = torch.empty(B, dim, Mpad, device=dev)
_lib.check(L.mmk_pack_target(_lib.ptr(tgt), B, M, 6, dim, _lib.ptr(planar), _lib.stream_ptr(dev)))
ws = torch.empty(L.mmk_nn_workspace_bytes(B, N, M, dim), dtype=torch.uint8, device=dev)
idx = torch.empty(B, N, dtype=torch.int32, device=dev); d2 = torch.empty(B, N, device=dev)
def call():
    _lib.check(L.mmk_nn_search(_lib.ptr(src), _lib.ptr(planar), _lib.ptr(T), B, N, M, dim, _lib.ptr(idx), _lib.ptr(d2), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev)))
for _ in range(5): call()
torch.cuda.synchronize()
reps = 40
_lib.check(L.mmk_nn_profile_begin(reps))
for _ in range(reps): call()
ms = (ctypes.c_float * reps)(); n = ctypes.c_int32(0)
_lib.check(L.mmk_nn_profile_end(ms, reps, ctypes.byref(n)))
t = np.array(ms[:n.value]) * 1e3
ir, dr = _clib.nn_search(src[:2, :, :dim].cpu().numpy(), np.ascontiguousarray(tgt[:2, :, :dim].cpu().numpy()))
ok = np.array_equal(idx[:2].cpu().numpy(), ir) and np.array_equal(d2[:2].cpu().numpy(), dr)
evals = B * N * Mpad
print("%-18s dim %d: median %.1f us  min %.1f us  -> %.2f Tevals/s  bit-exact=%s" % ("matrix-core filter", dim, np.median(t), t.min(), evals / np.median(t) / 1e6, ok), flush=True)
