"""Timing of the final layer's kernels at the bench shape (B=32, 640x640): forward, mask normalisation, backward with the
normalisation's adjoint.  Prints an md5 of the outputs (bit-identity across builds).  Development tool (GPU box)."""
import hashlib, sys
sys.path.insert(0, "."); sys.path.insert(0, "scripts")
import torch
from mm_masking_amd import _lib, unet_hip as uh
from bench_layers import timeit, DEV
B, H = 32, 640
g = torch.Generator().manual_seed(0)
x = (torch.randn(B, H, H, 8, generator=g) * 0.7).clamp_min(0).to(torch.bfloat16).to(DEV)
w = torch.randn(8, generator=g).to(DEV)
b = torch.randn(1, generator=g).to(DEV)
mask = uh.final_fwd(x, w, b)
print("final_fwd %.1f us" % timeit(lambda: uh.final_fwd(x, w, b)))
L = _lib.lib()
P = lambda t: _lib.ptr(t)
npix = H * H
part = torch.empty(B * 64, device=DEV); mask_n = torch.empty_like(mask); amax = torch.empty(B, device=DEV)
_lib.check(L.mmk_mask_normalize(P(mask), B, npix, P(part), P(mask_n), P(amax), _lib.stream_ptr(DEV)))
gm = torch.randn(B, H, H, generator=g).to(DEV)
coef = torch.empty(2 * B, device=DEV); gx = torch.empty_like(x); dW = torch.empty(8, device=DEV); db = torch.empty(1, device=DEV)
ws = torch.empty(_lib.FINAL_BWD_WS_FLOATS, device=DEV)
def bwd():
    _lib.check(L.mmk_final_bwd_normalized(P(x), P(w), P(mask), P(mask_n), P(amax), P(gm), B, npix, 1.05, 0.0, P(part), P(coef), P(gx), P(dW), P(db), P(ws),
                                          _lib.stream_ptr(DEV)))
bwd(); torch.cuda.synchronize()
h = hashlib.md5(gx.view(torch.int16).cpu().numpy().tobytes() + dW.cpu().numpy().tobytes() + db.cpu().numpy().tobytes()).hexdigest()[:12]
print("final_bwd_normalized (4 launches) %.1f us  md5 %s" % (timeit(bwd), h))
