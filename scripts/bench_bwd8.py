"""Timing of the fused backward of an 8 -> 8 layer at 640 x 640 (B = 32) against its two launches, alone and on two streams.
Development tool (GPU box)."""
import sys
import torch
sys.path.insert(0, ".")
from mm_masking_amd import unet_hip as uh

dev = torch.device("cuda:0")
B, H = 32, 640
x = (torch.randn(B, H, H, 8, device=dev) * 0.7).clamp_min(0).to(torch.bfloat16)
g = (torch.randn(B, H, H, 8, device=dev) * 0.3).to(torch.bfloat16)
w = torch.randn(8, 8, 3, 3, device=dev) / 8
wpt = uh.pack_weights(w, transposed=True)
ns = uh.wgrad_slices(8, 8, 8, B, H, H)
part = uh.partial_buffer(ns, 8, 8, dev)
dx = torch.empty_like(x)
side = torch.cuda.Stream()


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def two():
    uh.conv3x3(g, wpt, 8, out=dx, relu_src=x, scale=1.05)
    uh.conv3x3_wgrad_partial(x, g, 8, part)


def two_streams():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        uh.conv3x3_wgrad_partial(x, g, 8, part)
    uh.conv3x3(g, wpt, 8, out=dx, relu_src=x, scale=1.05)
    torch.cuda.current_stream().wait_stream(side)


print("slices", ns)
print("dgrad alone      %.1f us" % timeit(lambda: uh.conv3x3(g, wpt, 8, out=dx, relu_src=x, scale=1.05)))
print("wgrad alone      %.1f us" % timeit(lambda: uh.conv3x3_wgrad_partial(x, g, 8, part)))
print("both, one stream %.1f us" % timeit(two))
print("both, 2 streams  %.1f us" % timeit(two_streams))
print("fused            %.1f us" % timeit(lambda: uh.conv_bwd_fused(x, g, wpt, 1.05, dx, part)))
