import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_unet_kernels as T
from mm_masking_amd import unet_hip as uh
DEV = torch.device("cuda:0")
H = int(sys.argv[1]) if len(sys.argv) > 1 else 64
model = T._policy(0.0, torch.float32); model.train()
g = torch.Generator().manual_seed(1)
x = torch.rand(2, 1, H, H, generator=g).to(DEV); gsel = torch.randn(2, H, H, generator=g).to(DEV)
ref = T._emulated_unet(model, x); (ref * gsel).sum().backward()
gref = [q.grad.clone() for q in uh.param_list(model)]; model.zero_grad()
out = uh.unet_mask(model, x, True, 0); (out * gsel).sum().backward()
got = [q.grad.clone() for q in uh.param_list(model)]; model.zero_grad()
# a second emulation run with a different rounding of the final sum order: noise floor estimate = emulation vs itself in fp64? skip
names = [n for n, _ in model.named_parameters()]
print("fwd max err", (out - ref).abs().max().item())
for n, a, b in zip(names, got, gref):
    print("%-28s |ref| %9.4f  rel err %.4f" % (n, b.norm().item(), ((a - b).norm() / (b.norm() + 1e-12)).item()))
