"""A/B of the >= 64-channel convolution kernels: every deep forward / data-gradient shape of the network, outputs of the build
under test against a reference file written by another run (MMK_CONV_DEEP2=0/1, or MMK_LIB=...).
   python scripts/ab_deep.py save ref.pt ;  python scripts/ab_deep.py cmp ref.pt"""
import sys
import torch
sys.path.insert(0, ".")
from mm_masking_amd import unet_hip as uh
DEV = torch.device("cuda:0")
mode, path = sys.argv[1], sys.argv[2]
B = 4
shapes = [(16, 32, 0, 64), (160, 32, 0, 64), (160, 64, 0, 64), (80, 64, 0, 128), (80, 128, 0, 128), (40, 128, 0, 256), (40, 256, 0, 256), (40, 256, 0, 128),
          (40, 128, 128, 128), (80, 128, 0, 64), (80, 64, 64, 64), (50, 128, 0, 128), (25, 256, 0, 256), (37, 64, 64, 64)]
outs = {}
g = torch.Generator(device="cuda").manual_seed(3)
def rnd(*s):
    return (torch.randn(*s, device=DEV, generator=g) * 0.5).to(torch.bfloat16)
for H, c1, c2, co in shapes:
    W = {37: 53, 16: 32}.get(H, H)
    cin = c1 + c2
    x1 = rnd(B, H, W, c1); x2 = rnd(B, H, W, c2) if c2 else None
    w = torch.randn(co, cin, 3, 3, device=DEV, generator=g) / (3 * cin ** 0.5)
    bias = torch.randn(co, device=DEV, generator=g) * 0.1
    wp, wpt = uh.pack_weights(w), uh.pack_weights(w, transposed=True)
    y = torch.zeros(B, H, W, co, dtype=torch.bfloat16, device=DEV)
    uh.conv3x3(x1, wp, co, bias=bias, x2=x2, relu=True, drop_p=0.05, seed=3, out=y)
    outs["fwd %d %d+%d>%d" % (H, c1, c2, co)] = y.clone()
    gy = rnd(B, H, W, co)
    if co >= 32 and cin >= 64:
        if c2:
            o1 = torch.zeros(B, H, W, c1, dtype=torch.bfloat16, device=DEV); o2 = rnd(B, H, W, c2)
            uh.conv3x3(gy, wpt, cin, split=c1, out=o1, out2=o2, relu_src2=x2, scale2=1.05, accumulate2=True)
            outs["dgrad %d %d>%d+%d" % (H, co, c1, c2)] = torch.cat([o1, o2], -1).clone()
        else:
            o1 = torch.zeros(B, H, W, c1, dtype=torch.bfloat16, device=DEV)
            uh.conv3x3(gy, wpt, cin, out=o1, relu_src=x1, scale=1.05)
            outs["dgrad %d %d>%d" % (H, co, c1)] = o1.clone()
torch.cuda.synchronize()
if mode == "save":
    torch.save({k: v.cpu() for k, v in outs.items()}, path)
    print("saved", len(outs))
else:
    ref = torch.load(path)
    bad = 0
    for k, v in outs.items():
        eq = torch.equal(v.cpu(), ref[k])
        d = (v.cpu().float() - ref[k].float()).abs().max().item()
        if not eq:
            bad += 1
        print("%-28s %s  max|diff| %.3g" % (k, "bit-identical" if eq else "DIFFERENT", d))
    print("different:", bad)
