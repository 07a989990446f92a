"""Profiling driver for rocprofv3 --pmc passes over conv3x3_deep_kernel: forward (bias + ReLU + dropout) and data gradient
(ReLU source) of a 64->64 layer at 160x160 and a 128->128 layer at 80x80, B = 32.  Development tool (GPU box)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mm_masking_amd import unet_hip as uh

dev = torch.device("cuda:0")
B = 32
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
rnd = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
for cin, cout, H in [(64, 64, 160), (128, 128, 80)]:
    x, g = rnd(B, H, H, cin), rnd(B, H, H, cout)
    w = torch.randn(cout, cin, 3, 3, device=dev) / (3 * cin ** 0.5)
    b = torch.zeros(cout, device=dev)
    wp, wpt = uh.pack_weights(w), uh.pack_weights(w, transposed=True)
    y = torch.empty(B, H, H, cout, dtype=torch.bfloat16, device=dev)
    o = torch.empty(B, H, H, cin, dtype=torch.bfloat16, device=dev)
    for _ in range(reps):
        uh.conv3x3(x, wp, cout, bias=b, relu=True, drop_p=0.05, seed=1, out=y)
        uh.conv3x3(g, wpt, cin, out=o, relu_src=x, scale=1.05)
    torch.cuda.synchronize()
print("ok")
