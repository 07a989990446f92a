import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from mm_masking_amd import train_icp_weights as trn, unet_hip as uh
from mm_masking_amd.icp_weight_policy import LearnICPWeightPolicy
DEV = torch.device("cuda:0")
H = int(sys.argv[1]) if len(sys.argv) > 1 else 64
p = trn.default_params(DEV); p.update({"dropout": 0.0, "amp_dtype": torch.float32, "unet_backend": "torch"})
torch.manual_seed(11)
model = LearnICPWeightPolicy(p).to(DEV); model.train()
g = torch.Generator().manual_seed(1)
x = torch.rand(2, 1, H, H, generator=g).to(DEV); gsel = torch.randn(2, H, H, generator=g).to(DEV)
ref = model._unet(x.clone()); (ref * gsel).sum().backward()
gref = [q.grad.clone() for q in uh.param_list(model)]; model.zero_grad()
out = uh.unet_mask(model, x, True, 0); (out * gsel).sum().backward()
got = [q.grad for q in uh.param_list(model)]
names = [n for n, _ in model.named_parameters()]
print("fwd max err", (out - ref).abs().max().item())
for n, a, b in zip(names, got, gref):
    cos = torch.nn.functional.cosine_similarity(a.flatten(), b.flatten(), dim=0).item()
    print("%-28s |ref| %9.4f  rel err %.3f  cos %.4f" % (n, b.norm().item(), ((a - b).norm() / (b.norm() + 1e-12)).item(), cos))
# torch bf16 autocast vs torch fp32, for scale
model.zero_grad(); model.amp_dtype = torch.bfloat16
r2 = model._unet(x.clone()); (r2 * gsel).sum().backward()
g16 = [q.grad.clone() for q in uh.param_list(model)]
print("torch-bf16 vs torch-fp32: fwd max err", (r2 - ref).abs().max().item())
for n, a, b, c in zip(names, g16, gref, got):
    print("%-28s torch-bf16 rel err %.3f   hip-vs-torch-bf16 rel %.3f" % (n, ((a - b).norm() / (b.norm() + 1e-12)).item(), ((c - a).norm() / (a.norm() + 1e-12)).item()))
