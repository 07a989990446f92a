"""Micro-benchmark of the conv3x3 / wgrad kernels at the U-Net's layer shapes (B=32)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from mm_masking_amd import unet_hip as uh
dev = torch.device("cuda:0")
B = 32
shapes = [(8, 8, 640), (8, 16, 640), (16, 16, 640), (16, 8, 640), (16, 32, 320), (32, 32, 320), (32, 64, 160), (64, 64, 160),
          (64, 128, 80), (128, 128, 80), (128, 256, 40), (256, 256, 40), (256, 128, 40)]
only = sys.argv[1] if len(sys.argv) > 1 else None
def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for cin, cout, H in shapes:
    if only and only != "%d-%d" % (cin, cout): continue
    x = torch.randn(B, H, H, cin, device=dev).to(torch.bfloat16)
    g = torch.randn(B, H, H, cout, device=dev).to(torch.bfloat16)
    w = torch.randn(cout, cin, 3, 3, device=dev) / (3 * cin ** 0.5)
    b = torch.zeros(cout, device=dev)
    wp = uh.pack_weights(w)
    y = torch.empty(B, H, H, cout, device=dev, dtype=torch.bfloat16)
    t_f = timeit(lambda: uh.conv3x3(x, wp, cout, bias=b, relu=True, out=y))
    dWt = torch.zeros(9, cout, cin, device=dev); db = torch.zeros(cout, device=dev)
    t_w = timeit(lambda: uh.conv3x3_wgrad(x, g, cout, dWt=dWt, db=db))
    px = B * H * H
    byt = px * (cin + cout) * 2
    fl = 2.0 * px * 9 * cin * cout
    print("%3d->%3d @%3d  fwd %7.1f us  %5.2f TB/s %6.1f TF/s | wgrad %7.1f us  %5.2f TB/s %6.1f TF/s" %
          (cin, cout, H, t_f, byt / t_f / 1e6, fl / t_f / 1e6, t_w, byt / t_w / 1e6, fl / t_w / 1e6), flush=True)
