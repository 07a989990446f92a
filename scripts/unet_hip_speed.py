"""Diagnostic: HIP U-Net forward+backward time at B=32, 640x640 vs the MIOpen path."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from mm_masking_amd import train_icp_weights as trn
from mm_masking_amd.icp_weight_policy import LearnICPWeightPolicy
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
x = torch.rand(B, 640, 640, device=dev)
scan = {"fft_data": x, "fft_cfar": x, "raw_pc": torch.zeros(B, 4, 3, device=dev)}
mp = {"pc": torch.zeros(B, 4, 6, device=dev)}
for backend in ("hip", "torch"):
    p = trn.default_params(dev); p.update({"unet_backend": backend})
    torch.manual_seed(0)
    m = LearnICPWeightPolicy(p).to(dev); m.train()
    ts = []
    for it in range(6):
        torch.cuda.synchronize(); t = time.time()
        mask = m(scan, mp, None, mask_only=True)
        torch.cuda.synchronize(); t1 = time.time()
        mask.sum().backward()
        torch.cuda.synchronize(); t2 = time.time()
        ts.append((t1 - t, t2 - t1))
    print("%-6s fwd %.2f ms  bwd %.2f ms  (first iter %.1f s)" % (backend, min(a for a, b in ts[2:]) * 1e3, min(b for a, b in ts[2:]) * 1e3, sum(ts[0])), flush=True)
    if backend == "hip":
        print("peak mem GB", torch.cuda.max_memory_allocated() / 1e9)
