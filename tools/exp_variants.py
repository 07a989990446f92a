"""Run deep-layer conv probes against an experimental build (development tool):
   python tools/exp_variants.py build_exp/lib_X.so"""
import sys
sys.path.insert(0, ".")
from mm_masking_amd import _lib
_lib.SO_PATH = sys.argv[1]
import torch
from mm_masking_amd import unet_hip as uh
from tools.bench_layers import rnd, timeit, DEV

B = 32
out = []
for cin, co, H in [(64, 64, 160), (128, 128, 80), (256, 256, 40), (256, 128, 40)]:
    x = rnd(B, H, H, cin)
    w = torch.randn(co, cin, 3, 3, device=DEV) / (3 * cin ** 0.5)
    wp = uh.pack_weights(w)
    y = torch.empty(B, H, H, co, dtype=torch.bfloat16, device=DEV)
    out.append("%d>%d@%d %.0f" % (cin, co, H, timeit(lambda: uh.conv3x3(x, wp, co, out=y))))
print(sys.argv[1], " | ".join(out))
