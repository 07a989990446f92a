"""Times a few >= 64-channel convolution launches (forward form) at B = 32: development tool for the deep kernels."""
import sys, os
import torch
sys.path.insert(0, ".")
from mm_masking_amd import unet_hip as uh
DEV = torch.device("cuda:0")
B = 32
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
res = []
for H, cin, co in [(40, 128, 128), (40, 128, 256), (40, 256, 256), (40, 256, 128), (80, 128, 64)]:
    x = (torch.randn(B, H, H, cin, device=DEV) * 0.5).to(torch.bfloat16)
    w = torch.randn(co, cin, 3, 3, device=DEV) / (3 * cin ** 0.5)
    wp = uh.pack_weights(w)
    y = torch.empty(B, H, H, co, dtype=torch.bfloat16, device=DEV)
    bias = torch.zeros(co, device=DEV)
    t = timeit(lambda: uh.conv3x3(x, wp, co, bias=bias, relu=True, drop_p=float(os.environ.get("DROP", "0.05")), seed=3, out=y))
    res.append("%d:%d>%d %.1f us %.0f TF/s" % (H, cin, co, t, 2.0 * 9 * cin * co * H * H * B / t * 1e-6))
print(os.path.basename(os.environ.get("MMK_LIB", "default")), "|", " | ".join(res))
