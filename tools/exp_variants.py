"""Run deep-layer wgrad probes against an experimental build (development tool):
   python tools/exp_variants.py build_exp/lib_X.so"""
import sys
sys.path.insert(0, ".")
from mm_masking_amd import _lib
_lib.SO_PATH = sys.argv[1]
import torch
from mm_masking_amd import unet_hip as uh
from tools.bench_layers import rnd, timeit, DEV

out = []
for B in (8, 32):
    for cin, co, H in [(64, 64, 160), (128, 128, 80), (256, 256, 40)]:
        x = rnd(B, H, H, cin); g = rnd(B, H, H, co)
        dWt = torch.zeros(9, co, cin, device=DEV); db = torch.zeros(co, device=DEV)
        out.append("B%d %d>%d@%d %.0f" % (B, cin, co, H, timeit(lambda: uh.conv3x3_wgrad(x, g, co, dWt=dWt, db=db))))
print(sys.argv[1], " | ".join(out))
