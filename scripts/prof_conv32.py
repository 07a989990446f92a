"""rocprofv3 --pmc driver: 32->32 layer at 320x320, B=32 (ring kernel with two 16-channel tiles per pixel)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mm_masking_amd import unet_hip as uh
dev = torch.device("cuda:0")
B, H, cin, cout = 32, 320, 32, 32
x = (torch.randn(B, H, H, cin, device=dev) * 0.5).to(torch.bfloat16)
w = torch.randn(cout, cin, 3, 3, device=dev) / 17
b = torch.zeros(cout, device=dev)
wp = uh.pack_weights(w)
y = torch.empty(B, H, H, cout, dtype=torch.bfloat16, device=dev)
for _ in range(2):
    uh.conv3x3(x, wp, cout, bias=b, relu=True, out=y)
torch.cuda.synchronize()
print("ok")
