"""Half-period stamps of conv3x3_pp_kernel (diagnostic build -DMMK_DEEP_STAMPS, MMK_LIB=build_exp/lib_stamps.so): per group, the
cycles of each MFMA phase / rest phase and the time spent waiting at the barrier in front of it."""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from mm_masking_amd import _lib  # noqa: E402
from mm_masking_amd import unet_hip as uh  # noqa: E402

DEV = torch.device("cuda:0")
B = 32
L = _lib.lib()
L.mmk_debug_deep_stamps.restype = ctypes.c_int
L.mmk_debug_deep_stamps.argtypes = [ctypes.c_void_p]
buf = torch.zeros(64 * 8 * 32 * 4 + 8192, dtype=torch.int64, device=DEV)
_lib.check(L.mmk_debug_deep_stamps(buf.data_ptr()))
for name, H, cin, co in [("enc3.2", 160, 64, 64), ("enc4.2", 80, 128, 128), ("enc5.2", 40, 256, 256)]:
    g = torch.Generator(device="cpu").manual_seed(H + cin + co)
    x = (torch.randn(B, H, H, cin, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
    w = (torch.randn(co, cin, 3, 3, generator=g) / (3 * cin ** 0.5)).to(DEV)
    bias = (torch.randn(co, generator=g) * 0.1).to(DEV)
    gy = (torch.randn(B, H, H, co, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
    wp, wpt = uh.pack_weights(w), uh.pack_weights(w, transposed=True)
    y = torch.zeros(B, H, H, co, dtype=torch.bfloat16, device=DEV)
    o = torch.zeros(B, H, H, cin, dtype=torch.bfloat16, device=DEV)
    for role, fn in (("fwd", lambda: uh.conv3x3(x, wp, co, bias=bias, relu=True, drop_p=0.05, seed=3, out=y)),
                     ("dgrad", lambda: uh.conv3x3(gy, wpt, cin, out=o, relu_src=x, scale=1.05))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        buf.zero_()
        fn()
        torch.cuda.synchronize()
        t = buf.cpu().numpy()[:64 * 8 * 32 * 4].reshape(64, 8, 32, 4).astype(np.float64)
        print("%s %s  %d x %d, %d -> %d" % (name, role, H, H, cin, co))
        for grp in (0, 1):
            tw = t[:, 4 * grp:4 * grp + 4]                       # (block, wave, h, slot)
            line = []
            for h in range(1, 14):
                ok = tw[:, :, h, 0] > 0
                if not ok.any():
                    break
                dur = (tw[:, :, h, 1] - tw[:, :, h, 0])[ok].mean()
                wait = (tw[:, :, h, 0] - tw[:, :, h - 1, 1])[ok & (tw[:, :, h - 1, 1] > 0)].mean() if h > 0 else 0.0
                kind = int(np.median(tw[:, :, h, 2][ok]))
                line.append("%s%5.0f(+%4.0f)" % ({0: "-", 1: "M", 2: "r", 3: "E"}[kind], dur, wait))
            print("   group %d: " % grp + " ".join(line))
