"""Profiling driver for rocprofv3 --pmc passes over the U-Net convolution kernels at the bench shape
(B=32, 640x640): one forward (bias+ReLU+dropout) and one data-gradient launch (ReLU source) of an
8->8 and a 16->16 layer, one 128->128 layer at 80x80 and its weight gradient.  The counters of
interest are FETCH_SIZE / WRITE_SIZE (HBM traffic per launch vs the algorithmic bytes)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mm_masking_amd import unet_hip as uh

dev = torch.device("cuda:0")
B = 32
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3


def rnd(*shape):
    return (torch.randn(*shape, device=dev) * 0.5).to(torch.bfloat16)


for cin, cout, H in [(8, 8, 640), (16, 16, 640), (128, 128, 80)]:
    x, g = rnd(B, H, H, cin), rnd(B, H, H, cout)
    w = torch.randn(cout, cin, 3, 3, device=dev) / (3 * cin ** 0.5)
    b = torch.zeros(cout, device=dev)
    wp, wpt = uh.pack_weights(w), uh.pack_weights(w, transposed=True)
    y = torch.empty(B, H, H, cout, dtype=torch.bfloat16, device=dev)
    o = torch.empty(B, H, H, cin, dtype=torch.bfloat16, device=dev)
    ns = uh.wgrad_slices(cout, cin, cin, B, H, H)
    part = uh.partial_buffer(ns, cout, cin, dev)
    for _ in range(reps):
        uh.conv3x3(x, wp, cout, bias=b, relu=True, drop_p=0.05, seed=1, out=y)
        uh.conv3x3(g, wpt, cin, out=o, relu_src=x, scale=1.05)
        uh.conv3x3_wgrad_partial(x, g, cout, part)
    torch.cuda.synchronize()
print("ok")
