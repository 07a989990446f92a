"""Where a tile of conv3x3_ring_kernel spends its cycles: reads the s_memtime stamps of a -DMMK_DEEP_STAMPS build of the library
(bash scripts/build_variant.sh rstamps -DMMK_DEEP_STAMPS; run with MMK_LIB=build_exp/lib_rstamps.so).  Per shape: mean cycles
between the six stamp points of a stage (= tile), over the first 64 blocks' four waves, stage by stage.
  0 stage top | 1 behind the barrier | 2 tile k+1 written to LDS (with its wait for the loads) | 3 loads of tile k+1+RD issued
  | 4 MFMA section issued | 5 epilogue + store issued"""
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
buf = torch.zeros(64 * 4 * STAGES * 8 + 2 * 4096, dtype=torch.int64, device=DEV)
_lib.check(L.mmk_debug_deep_stamps(buf.data_ptr()))
names = ["barrier", "wait+LDS wr", "issue loads", "MFMA", "epilogue+st", "to next top"]
for H, cin, co, drop in [(640, 8, 8, 0.05), (640, 8, 16, 0.0), (640, 16, 16, 0.05), (640, 16, 8, 0.0), (320, 32, 32, 0.05)]:
    x = (torch.randn(B, H, H, cin, device=DEV) * 0.5).to(torch.bfloat16)
    w = torch.randn(co, cin, 3, 3, device=DEV) / (3 * cin ** 0.5)
    wp = uh.pack_weights(w)
    y = torch.empty(B, H, H, co, dtype=torch.bfloat16, device=DEV)
    bias = torch.zeros(co, device=DEV)
    fn = lambda: uh.conv3x3(x, wp, co, bias=bias, relu=True, drop_p=drop, seed=3, out=y)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    buf.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    raw = buf.cpu().numpy()
    t = raw[:64 * 4 * STAGES * 8].reshape(64, 4, STAGES, 8).astype(np.float64)
    life = raw[64 * 4 * STAGES * 8:].reshape(4096, 2).astype(np.float64)
    used = t[..., 0] > 0
    print("%d x %d, %d -> %d, dropout %.2f   %.1f us (instrumented)   grid %d blocks, %d tiles per block" % (
        H, H, cin, co, drop, e0.elapsed_time(e1) * 1e3, int(t[0, 0, 0, 6]), int(t[0, 0, 0, 7])))
    dm, dr = t[:, 0, 2, 6] - t[:, 0, 1, 6], t[:, 0, 2, 7] - t[:, 0, 1, 7]
    print("   a block lives %.0f stamp counts = %.1f us of the 100 MHz clock: the stamps count at %.2f GHz" % (dm.mean(), dr.mean() / 100.0, dm.mean() / dr.mean() * 0.1))
    nblk = int(t[0, 0, 0, 6])
    lf = life[:nblk]
    lf = (lf - lf[:, 0].min()) / 100.0                      # us since the first block started
    order = np.argsort(lf[:, 0])
    print("   block starts (us): " + " ".join("%.0f" % v for v in np.percentile(lf[:, 0], [0, 10, 25, 50, 60, 75, 90, 100])) +
          "   ends: " + " ".join("%.0f" % v for v in np.percentile(lf[:, 1], [0, 10, 25, 50, 75, 90, 100])) +
          "   lives: " + " ".join("%.0f" % v for v in np.percentile(lf[:, 1] - lf[:, 0], [0, 25, 50, 75, 100])))
    lv = lf[:, 1] - lf[:, 0]
    print("   mean life by XCD (block %% 8): " + " ".join("%.0f" % lv[x::8].mean() for x in range(8)))
    per = lv.reshape(-1, 8).mean(axis=1)                     # by index inside the XCD (block // 8)
    print("   mean life by block // 8, in 16 groups: " + " ".join("%.0f" % v.mean() for v in np.array_split(per, 16)))
    np.save("gpurun_out/ring_life_%d_%d_%d.npy" % (H, cin, co), lf)
    print("   blocks that started within 5 us of the first: %d of %d" % (int((lf[:, 0] < 5).sum()), nblk))
    print("   stage " + " ".join("%11s" % n for n in names) + "      total")
    for st in range(3, 15):
        ok = used[:, :, st] & used[:, :, st + 1]
        if not ok.any():
            break
        seq = np.concatenate([t[:, :, st, :6], t[:, :, st + 1, :1]], axis=-1)      # (64, 4, 7)
        d = np.diff(seq, axis=-1)[ok]
        print("   %5d " % st + " ".join("%11.0f" % v for v in d.mean(axis=0)) + "  %9.0f" % d.sum(axis=1).mean())
    st = 6
    t0 = t[:, :, st, 0]
    print("   stage-6 top: spread over the waves of a block %.0f cycles (mean)" % np.mean(t0.max(axis=1) - t0.min(axis=1)))
