"""Where a stage of conv3x3_deep_kernel spends its cycles: reads the s_memtime stamps of a -DMMK_DEEP_STAMPS build of the library
(bash scripts/build_variant.sh stamps -DMMK_DEEP_STAMPS; run with MMK_LIB=build_exp/lib_stamps.so).  Per layer shape and role:
mean cycles between the eight stamp points of a stage, over the first 64 blocks' eight waves, stage by stage.
  0 loop top | 1 next stage's addresses set | 2 MFMA loop (with the prefetch loads inside) issued | 3 epilogue done (last chunk)
  | 4 prefetch delivered | 5 barrier (fragments read) | 6 stage written to LDS | 7 barrier"""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from mm_masking_amd import _lib  # noqa: E402
from mm_masking_amd import unet_hip as uh  # noqa: E402

DEV = torch.device("cuda:0")
B = 32
STAGES = 24
L = _lib.lib()
L.mmk_debug_deep_stamps.restype = ctypes.c_int
L.mmk_debug_deep_stamps.argtypes = [ctypes.c_void_p]
buf = torch.zeros(64 * 8 * STAGES * 8, dtype=torch.int64, device=DEV)
_lib.check(L.mmk_debug_deep_stamps(buf.data_ptr()))
names = ["set stage", "MFMA+loads", "epilogue", "deliver", "barrier A", "write LDS", "barrier B"]
layers = [("enc3.2", 160, 64, 64), ("enc4.2", 80, 128, 128), ("enc5.2", 40, 256, 256), ("dec1.2", 80, 64, 64)]
for name, H, cin, co in layers:
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
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        t = buf.cpu().numpy().reshape(64, 8, STAGES, 8).astype(np.float64)
        used = t[..., 0] > 0                                   # (block, wave, stage) recorded
        nst = int(used[0, 0].sum())
        print("%s %s  %d x %d, %d -> %d   %.1f us (instrumented)   stages recorded per block: %d" % (name, role, H, H, cin, co,
                                                                                               e0.elapsed_time(e1) * 1e3, nst))
        d = np.diff(t, axis=-1)                                 # (64, 8, STAGES, 7)
        print("   stage " + " ".join("%11s" % n for n in names) + "      total")
        for st in range(min(nst, 12)):
            m = d[:, :, st][used[:, :, st]].reshape(-1, 7)
            # a stamp that was not taken in this stage (no barrier A / epilogue on some stages) shows as a non-positive difference
            m = np.where(np.abs(m) > 1e9, np.nan, m)
            mean = np.nanmean(m, axis=0)
            tot = (t[:, :, st, 7] - t[:, :, st, 0])[used[:, :, st]]
            print("   %5d " % st + " ".join("%11.0f" % v for v in mean) + "  %9.0f" % np.mean(tot))
        # spread of the stage start between the waves of a block and between blocks (stage 2)
        st = 2
        t0 = t[:, :, st, 0]
        print("   stage-2 start: spread over the waves of a block %.0f cycles (mean), over blocks %.0f cycles (std)" % (
            np.mean(t0.max(axis=1) - t0.min(axis=1)), np.std(t0.mean(axis=1))))
