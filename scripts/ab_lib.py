"""A/B of two builds of the library on the >= 64-channel forward / data-gradient launch shapes: this process loads the library
MMK_LIB points at (default: the in-tree one), runs every shape, prints the median time and an md5 of each output; run it once
per build inside ONE gpurun call and compare the lines.  python scripts/ab_lib.py [B] [rounds]"""
import hashlib
import os
import sys

import torch

sys.path.insert(0, ".")
from mm_masking_amd import unet_hip as uh  # noqa: E402

DEV = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 5


def timeit(fn, n=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def md5(*ts):
    h = hashlib.md5()
    for t in ts:
        h.update(t.view(torch.int16).cpu().numpy().tobytes())
    return h.hexdigest()[:10]


layers = [("enc3.0", 160, 32, 0, 64), ("enc3.2", 160, 64, 0, 64), ("enc4.0", 80, 64, 0, 128), ("enc4.2", 80, 128, 0, 128), ("enc5.0", 40, 128, 0, 256),
          ("enc5.2", 40, 256, 0, 256), ("dec0.0u", 40, 256, 0, 128), ("dec0.2", 40, 128, 0, 128), ("dec0.0c", 40, 128, 128, 128),
          ("dec1.0u", 80, 128, 0, 64), ("dec1.2", 80, 64, 0, 64), ("dec1.0c", 80, 64, 64, 64), ("dec2.0u", 160, 64, 0, 32), ("dec2.0c", 160, 32, 32, 32)]
tot_f = tot_d = 0.0
print("lib", os.environ.get("MMK_LIB", "in-tree"))
for name, H, c1, c2, co in layers:
    cin = c1 + c2
    g = torch.Generator(device="cpu").manual_seed(H + cin + co)
    x1 = (torch.randn(B, H, H, c1, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
    x2 = (torch.randn(B, H, H, c2, generator=g) * 0.5).to(torch.bfloat16).to(DEV) if c2 else None
    w = (torch.randn(co, cin, 3, 3, generator=g) / (3 * cin ** 0.5)).to(DEV)
    bias = (torch.randn(co, generator=g) * 0.1).to(DEV)
    gy = (torch.randn(B, H, H, co, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
    wp, wpt = uh.pack_weights(w), uh.pack_weights(w, transposed=True)
    y = torch.zeros(B, H, H, co, dtype=torch.bfloat16, device=DEV)
    fwd = lambda: uh.conv3x3(x1, wp, co, bias=bias, x2=x2, relu=True, drop_p=0.05, seed=3, out=y)
    if c2:
        o = (torch.zeros(B, H, H, c1, dtype=torch.bfloat16, device=DEV), torch.zeros(B, H, H, c2, dtype=torch.bfloat16, device=DEV))
        dgr = lambda: uh.conv3x3(gy, wpt, cin, split=c1, out=o[0], out2=o[1], relu_src2=x2, scale2=1.05)
    else:
        o = (torch.zeros(B, H, H, c1, dtype=torch.bfloat16, device=DEV),)
        dgr = lambda: uh.conv3x3(gy, wpt, cin, out=o[0], relu_src=x1, scale=1.05)
    tf = sorted(timeit(fwd) for _ in range(ROUNDS))[ROUNDS // 2]
    td = sorted(timeit(dgr) for _ in range(ROUNDS))[ROUNDS // 2]
    tot_f += tf
    tot_d += td
    print("%-8s %4d %4d>%-4d fwd %7.1f us %s | dgrad %7.1f us %s" % (name, H, cin, co, tf, md5(y), td, md5(*o)), flush=True)
print("sum us: fwd %.0f dgrad %.0f" % (tot_f, tot_d))
