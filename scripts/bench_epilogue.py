"""Epilogue cost probe for the full-resolution conv layers (development tool)."""
import sys
import torch
sys.path.insert(0, "."); sys.path.insert(0, "scripts")
from mm_masking_amd import unet_hip as uh
from bench_layers import rnd, timeit, DEV

B, H = 32, 640
for cin, co in [(8, 8), (8, 16), (16, 16), (16, 8)]:
    x = rnd(B, H, H, cin); g = rnd(B, H, H, co)
    w = torch.randn(co, cin, 3, 3, device=DEV) / (3 * cin ** 0.5)
    bias = torch.zeros(co, device=DEV)
    wp, wpt = uh.pack_weights(w), uh.pack_weights(w, transposed=True)
    y = torch.empty(B, H, H, co, dtype=torch.bfloat16, device=DEV)
    o = torch.empty(B, H, H, cin, dtype=torch.bfloat16, device=DEV)
    r = {}
    r["plain"] = timeit(lambda: uh.conv3x3(x, wp, co, out=y))
    r["bias+relu"] = timeit(lambda: uh.conv3x3(x, wp, co, bias=bias, relu=True, out=y))
    r["+drop"] = timeit(lambda: uh.conv3x3(x, wp, co, bias=bias, relu=True, drop_p=0.05, seed=1, out=y))
    r["dgrad plain"] = timeit(lambda: uh.conv3x3(g, wpt, cin, out=o))
    r["dgrad relu_src"] = timeit(lambda: uh.conv3x3(g, wpt, cin, out=o, relu_src=x, scale=1.05))
    r["dgrad relu_src+acc"] = timeit(lambda: uh.conv3x3(g, wpt, cin, out=o, relu_src=x, scale=1.05, accumulate=True))
    print(cin, co, " ".join("%s=%.0f" % kv for kv in r.items()), flush=True)
