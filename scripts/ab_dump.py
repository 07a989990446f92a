"""Dump the forward output of one >= 64-channel launch (bias, ReLU, optional dropout) with the library MMK_LIB points at:
python scripts/ab_dump.py OUT.pt [drop_p]; compare two dumps with scripts/ab_dump.py --cmp A.pt B.pt"""
import sys
import torch
if sys.argv[1] == "--cmp":
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    for k in a:
        x, y = a[k].float(), b[k].float()
        d = (a[k].view(torch.int16) != b[k].view(torch.int16))
        print(k, "bit diffs", int(d.sum()), "of", d.numel(), "value diffs", int((x != y).sum()), "max abs", float((x - y).abs().max()),
              "zeros", int((x == 0).sum()), int((y == 0).sum()))
        if d.any():
            idx = d.nonzero()[:8]
            for i in idx:
                print("   ", i.tolist(), float(x[tuple(i)]), float(y[tuple(i)]))
            print("    channels with diffs", sorted(set(d.nonzero()[:, 3].tolist()))[:64])
    sys.exit(0)
sys.path.insert(0, ".")
from mm_masking_amd import unet_hip as uh  # noqa: E402
DEV = torch.device("cuda:0")
out = {}
for drop in (0.0, 0.05):
    for name, H, cin, co in [("enc3.2", 160, 64, 64), ("enc4.2", 80, 128, 128)]:
        g = torch.Generator(device="cpu").manual_seed(H + cin + co)
        x = (torch.randn(2, H, H, cin, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
        w = (torch.randn(co, cin, 3, 3, generator=g) / (3 * cin ** 0.5)).to(DEV)
        bias = (torch.randn(co, generator=g) * 0.1).to(DEV)
        wp = uh.pack_weights(w)
        for nb in (0, 1):
            y = uh.conv3x3(x, wp, co, bias=bias if nb else None, relu=True, drop_p=drop, seed=3)
            out["%s drop%g bias%d" % (name, drop, nb)] = y.cpu()
torch.save(out, sys.argv[1])
