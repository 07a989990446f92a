import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm_masking_amd import unet_hip as uh
dev = torch.device("cuda:0")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for C, H in ((8, 640), (16, 640), (16, 320)):
    B = 32
    x = (torch.randn(B, H, H, C, device=dev) * 0.7).clamp_min(0).to(torch.bfloat16)
    g = (torch.randn(B, H, H, C, device=dev) * 0.3).to(torch.bfloat16)
    w = torch.randn(C, C, 3, 3, device=dev) / C
    wpt = uh.pack_weights(w, transposed=True)
    ns = uh.wgrad_slices(C, C, C, B, H, H)
    part = uh.partial_buffer(ns, C, C, dev); dx = torch.empty_like(x)
    def two():
        uh.conv3x3(g, wpt, C, out=dx, relu_src=x, scale=1.05); uh.conv3x3_wgrad_partial(x, g, C, part)
    print("C=%d H=%d  two launches %.1f us   fused %.1f us" % (C, H, timeit(two), timeit(lambda: uh.conv_bwd_fused(x, g, wpt, 1.05, dx, part))))

B, H = 32, 640
x = (torch.randn(B, H, H, 8, device=dev) * 0.7).clamp_min(0).to(torch.bfloat16)
g = (torch.randn(B, H, H, 16, device=dev) * 0.3).to(torch.bfloat16)
w = torch.randn(16, 8, 3, 3, device=dev) / 8
wpt = uh.pack_weights(w, transposed=True)
ns = uh.wgrad_slices(16, 8, 8, B, H, H)
part = uh.partial_buffer(ns, 16, 8, dev); dx = torch.zeros_like(x)
def two():
    uh.conv3x3(g, wpt, 8, out=dx, relu_src=x, scale=1.05, accumulate=True); uh.conv3x3_wgrad_partial(x, g, 16, part)
print("8x16 H=640  two launches %.1f us   fused %.1f us" % (timeit(two), timeit(lambda: uh.conv8x16_bwd_fused(x, g, wpt, 1.05, dx, part))))
