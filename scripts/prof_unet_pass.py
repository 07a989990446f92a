"""Profiling driver: the mask U-Net alone, forward + backward at the bench shape (B=32, 640x640, dropout 0.05, amax-normalised
mask), REPS passes on ONE stream (MMK_UNET_SIDE_STREAM=0 set here: with the weight gradients on the side stream the dispatch
order is not the program order and durations overlap).  Run under rocprofv3 (scripts/pmc_unet.sh)."""
import os, sys
os.environ["MMK_UNET_SIDE_STREAM"] = "0"
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mm_masking_amd import train_icp_weights as trn
from mm_masking_amd.icp_weight_policy import LearnICPWeightPolicy
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
params = trn.default_params(dev)
torch.manual_seed(1)
model = LearnICPWeightPolicy(params).to(dev)
model.train()
g = torch.Generator().manual_seed(2)
img = torch.rand(B, 640, 640, generator=g).to(dev)
scan = {"fft_data": img, "fft_cfar": torch.zeros_like(img), "raw_pc": torch.zeros(B, 4, 3, device=dev)}
mp = {"pc": torch.zeros(B, 4, 6, device=dev)}
gsel = torch.randn(B, 640, 640, generator=g).to(dev)
for _ in range(reps):
    model.zero_grad(set_to_none=True)
    m = model(scan, mp, None, mask_only=True)
    (m * gsel).sum().backward()
    torch.cuda.synchronize()
print("ok")
