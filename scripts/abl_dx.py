"""Ablation timings of conv3x3_dx_kernel (a -DMMK_DX_ABLATIONS build loaded through MMK_LIB): MMK_DX_ABL bit 0 = no DMA after
the prologue, bit 1 = no fragment reads / MFMAs, bit 2 = no epilogue.  Development tool."""
import os
import sys

import torch

sys.path.insert(0, ".")
from mm_masking_amd import unet_hip as uh  # noqa: E402

DEV = torch.device("cuda:0")
B = 32
ABLS = [0, 1, 2, 3, 4, 5, 6, 7, 8, 12]


def timeit(fn, n=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


layers = [("enc3.2", 160, 64, 64), ("enc4.2", 80, 128, 128), ("enc5.2", 40, 256, 256), ("dec0.2", 40, 128, 128), ("dec1.0u", 80, 128, 64)]
print("%-8s %9s | %s" % ("layer", "deep", "  ".join("abl%d" % a for a in ABLS)))
for name, H, cin, co in layers:
    x = (torch.randn(B, H, H, cin, device=DEV) * 0.5).to(torch.bfloat16)
    w = torch.randn(co, cin, 3, 3, device=DEV) / (3 * cin ** 0.5)
    bias = torch.zeros(co, device=DEV)
    wp = uh.pack_weights(w)
    y = torch.empty(B, H, H, co, dtype=torch.bfloat16, device=DEV)
    os.environ["MMK_CONV_DX"] = "0"
    t_deep = timeit(lambda: uh.conv3x3(x, wp, co, bias=bias, relu=True, drop_p=0.05, seed=3, out=y))
    os.environ["MMK_CONV_DX"] = "1"
    ts = []
    for abl in ABLS:
        os.environ["MMK_DX_ABL"] = str(abl)
        ts.append(timeit(lambda: uh.conv3x3(x, wp, co, bias=bias, relu=True, drop_p=0.05, seed=3, out=y)))
    print("%-8s %9.1f | %s" % (name, t_deep, "  ".join("%5.1f" % t for t in ts)), flush=True)
