"""Timing of the first-layer and up-sampling kernels at the bench shape (development tool)."""
import sys, ctypes
sys.path.insert(0, "."); sys.path.insert(0, "scripts")
import torch
from mm_masking_amd import _lib, unet_hip as uh
from bench_layers import rnd, timeit, DEV
B = 32
x = torch.randn(B, 1, 640, 640, device=DEV)
w0 = torch.randn(8, 1, 3, 3, device=DEV); b0 = torch.zeros(8, device=DEV)
g = rnd(B, 640, 640, 8)
dw = torch.zeros(8, 1, 3, 3, device=DEV); db = torch.zeros(8, device=DEV)
L = _lib.lib()
p = lambda t: ctypes.c_void_p(t.data_ptr())
print("conv_first %.1f us" % timeit(lambda: uh.conv_first(x, w0, b0)))
print("conv_first_wgrad %.1f us" % timeit(lambda: uh.conv_first_wgrad(x, g, None, dw, db)))
import hashlib
torch.cuda.synchronize()
print("conv_first_wgrad md5", hashlib.md5(dw.cpu().numpy().tobytes() + db.cpu().numpy().tobytes()).hexdigest()[:12])
for H, C in [(640, 16), (320, 32)]:
    s = rnd(B, H // 2, H // 2, C); gy = rnd(B, H, H, C)
    print("up %d c%d fwd %.1f bwd %.1f us" % (H, C, timeit(lambda: uh.upsample(s, H, H)), timeit(lambda: uh.upsample_bwd(gy, H // 2, H // 2, relu_src=s, scale=1.05))))
