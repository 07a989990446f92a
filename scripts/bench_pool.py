"""Fused 2x2 max-pool in the conv epilogue vs conv + pooling kernel at the two levels it covers (B=32).
Development tool (GPU box)."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "scripts")
import torch
from mm_masking_amd import unet_hip as uh
from bench_layers import rnd, timeit, DEV

B = 32
for H, c in [(640, 16), (320, 32)]:
    x = rnd(B, H, H, c)
    w = torch.randn(c, c, 3, 3, device=DEV) / (3 * c ** 0.5)
    b = torch.zeros(c, device=DEV)
    wp = uh.pack_weights(w)
    y = torch.empty(B, H, H, c, dtype=torch.bfloat16, device=DEV)
    p = torch.empty(B, H // 2, H // 2, c, dtype=torch.bfloat16, device=DEV)
    t0 = timeit(lambda: uh.conv3x3(x, wp, c, bias=b, relu=True, drop_p=0.05, seed=3, out=y))
    t1 = timeit(lambda: uh.maxpool2(y))
    t2 = timeit(lambda: uh.conv3x3(x, wp, c, bias=b, relu=True, drop_p=0.05, seed=3, out=y, pool_out=p))
    print("H=%d c=%d: conv %.1f us + pool %.1f us = %.1f   fused %.1f us" % (H, c, t0, t1, t0 + t1, t2))
