"""A/B of two builds of the library on the >= 64-channel layers (development tool):
   python scripts/ab_layers.py build_exp/lib_A.so   (one build per process)"""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "scripts")
from mm_masking_amd import _lib
_lib.SO_PATH = sys.argv[1]
import torch
from mm_masking_amd import unet_hip as uh
from bench_layers import rnd, timeit, DEV

B = 32
tot_f = tot_d = 0.0
for cin, co, H in [(64, 64, 160), (64, 128, 80), (128, 128, 80), (128, 256, 40), (256, 256, 40), (256, 128, 40), (128, 64, 80), (64, 64, 80)]:
    x, g = rnd(B, H, H, cin), rnd(B, H, H, co)
    w = torch.randn(co, cin, 3, 3, device=DEV) / (3 * cin ** 0.5)
    b = torch.zeros(co, device=DEV)
    wp, wpt = uh.pack_weights(w), uh.pack_weights(w, transposed=True)
    y = torch.empty(B, H, H, co, dtype=torch.bfloat16, device=DEV)
    o = torch.empty(B, H, H, cin, dtype=torch.bfloat16, device=DEV)
    tf = timeit(lambda: uh.conv3x3(x, wp, co, bias=b, relu=True, drop_p=0.05, seed=1, out=y), 20)
    td = timeit(lambda: uh.conv3x3(g, wpt, cin, out=o, relu_src=x, scale=1.05), 20)
    tot_f += tf; tot_d += td
print(sys.argv[1], "fwd %.0f us  dgrad %.0f us" % (tot_f, tot_d))
