"""Times the thin (<= 32-channel) forward convolutions at B = 32 with and without dropout (DROP=0): what the dropout hash costs."""
import sys, os
import torch
sys.path.insert(0, ".")
from mm_masking_amd import unet_hip as uh
DEV = torch.device("cuda:0")
B = 32
drop = float(os.environ.get("DROP", "0.05"))
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
res = []
for H, cin, co in [(640, 8, 8), (640, 8, 16), (640, 16, 16), (640, 16, 8), (320, 16, 32), (320, 32, 32), (320, 32, 16)]:
    x = (torch.randn(B, H, H, cin, device=DEV) * 0.5).to(torch.bfloat16)
    w = torch.randn(co, cin, 3, 3, device=DEV) / (3 * cin ** 0.5)
    wp = uh.pack_weights(w)
    y = torch.empty(B, H, H, co, dtype=torch.bfloat16, device=DEV)
    bias = torch.zeros(co, device=DEV)
    t = timeit(lambda: uh.conv3x3(x, wp, co, bias=bias, relu=True, drop_p=drop, seed=3, out=y))
    res.append("%d:%d>%d %.1f us %.2f TB/s" % (H, cin, co, t, B * H * H * (cin + co) * 2 / t * 1e-6))
print("drop %.2f |" % drop, " | ".join(res))
if os.environ.get("COPY_REF"):
    # what a plain device copy of the same tensors reaches (read N bytes + write N bytes)
    out = []
    for H, c in [(640, 8), (640, 16), (320, 32)]:
        x = torch.randn(B, H, H, c, device=DEV).to(torch.bfloat16)
        y = torch.empty_like(x)
        t = timeit(lambda: y.copy_(x))
        out.append("%d:%d %.1f us %.2f TB/s" % (H, c, t, 2 * x.numel() * 2 / t * 1e-6))
    print("torch copy |", " | ".join(out))
