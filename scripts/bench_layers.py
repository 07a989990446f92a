"""Per-layer timing of the U-Net kernels at the bench shape (B=32, 640x640): achieved
TFLOP/s and GB/s (algorithmic bytes) for every distinct forward / data-gradient / weight-
gradient launch.  Development tool (run on the GPU box)."""
import sys

import torch

sys.path.insert(0, ".")
from mm_masking_amd import unet_hip as uh  # noqa: E402

DEV = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 32
DEEP_ONLY = len(sys.argv) > 2 and sys.argv[2] == "deep"      # only the >= 64-channel layers, no elementwise kernels
ELEM_ONLY = len(sys.argv) > 2 and sys.argv[2] == "elem"      # only the elementwise kernels
H0 = 640


def rnd(*shape):
    return (torch.randn(*shape, device=DEV) * 0.5).to(torch.bfloat16)


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3   # us


def main():
    rows = []
    # (name, H, cin1, cin2, cout)
    layers = [("enc0.2", 640, 8, 0, 8), ("enc1.0", 640, 8, 0, 16), ("enc1.2", 640, 16, 0, 16)]
    layers += [("enc2.0", 320, 16, 0, 32), ("enc2.2", 320, 32, 0, 32), ("enc3.0", 160, 32, 0, 64), ("enc3.2", 160, 64, 0, 64),
               ("enc4.0", 80, 64, 0, 128), ("enc4.2", 80, 128, 0, 128), ("enc5.0", 40, 128, 0, 256), ("enc5.2", 40, 256, 0, 256)]
    layers += [("dec0.0u", 40, 256, 0, 128), ("dec0.2", 40, 128, 0, 128), ("dec0.0c", 40, 128, 128, 128),
               ("dec1.0u", 80, 128, 0, 64), ("dec1.2", 80, 64, 0, 64), ("dec1.0c", 80, 64, 64, 64),
               ("dec2.0u", 160, 64, 0, 32), ("dec2.2", 160, 32, 0, 32), ("dec2.0c", 160, 32, 32, 32),
               ("dec3.0u", 320, 32, 0, 16), ("dec3.2", 320, 16, 0, 16), ("dec3.0c", 320, 16, 16, 16),
               ("dec4.0u", 640, 16, 0, 8), ("dec4.2", 640, 8, 0, 8), ("dec4.0c", 640, 8, 8, 8)]
    tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    print("%-9s %4s %9s | %8s %7s %7s | %8s %7s %7s | %8s %7s" % ("layer", "H", "cin>cout", "fwd us", "TF/s", "GB/s",
                                                                    "dgrad us", "TF/s", "GB/s", "wgrad us", "TF/s"))
    for name, H, c1, c2, co in layers:
        cin = c1 + c2
        if ELEM_ONLY or (DEEP_ONLY and max(cin, co) < 64):
            continue
        x1 = rnd(B, H, H, c1)
        x2 = rnd(B, H, H, c2) if c2 else None
        w = torch.randn(co, cin, 3, 3, device=DEV) / (3 * cin ** 0.5)
        bias = torch.zeros(co, device=DEV)
        wp, wpt = uh.pack_weights(w), uh.pack_weights(w, transposed=True)
        y = torch.empty(B, H, H, co, dtype=torch.bfloat16, device=DEV)
        g = rnd(B, H, H, co)
        flop = 2.0 * 9 * cin * co * H * H * B
        npx = B * H * H
        t_f = timeit(lambda: uh.conv3x3(x1, wp, co, bias=bias, x2=x2, relu=True, drop_p=0.05, seed=3, out=y))
        by_f = npx * (cin + co) * 2
        if c2:
            o1 = torch.empty(B, H, H, c1, dtype=torch.bfloat16, device=DEV)
            o2 = torch.empty(B, H, H, c2, dtype=torch.bfloat16, device=DEV)
            t_d = timeit(lambda: uh.conv3x3(g, wpt, cin, split=c1, out=o1, out2=o2, relu_src2=x2, scale2=1.05))
            by_d = npx * (co + cin + c2) * 2
        else:
            o1 = torch.empty(B, H, H, c1, dtype=torch.bfloat16, device=DEV)
            t_d = timeit(lambda: uh.conv3x3(g, wpt, cin, out=o1, relu_src=x1, scale=1.05))
            by_d = npx * (co + 2 * cin) * 2
        dWt = torch.zeros(9, co, cin, device=DEV)
        db = torch.zeros(co, device=DEV)
        t_w = timeit(lambda: uh.conv3x3_wgrad(x1, g, co, x2=x2, dWt=dWt, db=db))
        tot["fwd"] += t_f; tot["dgrad"] += t_d; tot["wgrad"] += t_w
        print("%-9s %4d %4d>%-4d | %8.1f %7.1f %7.0f | %8.1f %7.1f %7.0f | %8.1f %7.1f" % (
            name, H, cin, co, t_f, flop / t_f * 1e-6, by_f / t_f * 1e-3, t_d, flop / t_d * 1e-6, by_d / t_d * 1e-3,
            t_w, flop / t_w * 1e-6), flush=True)
        del x1, x2, y, g, o1
    print("sum us: fwd %.0f dgrad %.0f wgrad %.0f" % (tot["fwd"], tot["dgrad"], tot["wgrad"]))
    if DEEP_ONLY:
        return
    # elementwise layers
    for H, C in [(640, 8), (320, 16), (160, 32), (80, 64), (40, 128)]:
        pass
    x = torch.randn(B, 1, 640, 640, device=DEV)
    w0 = torch.randn(8, 1, 3, 3, device=DEV); b0 = torch.zeros(8, device=DEV)
    t = timeit(lambda: uh.conv_first(x, w0, b0))
    print("conv_first us %.1f  (%.0f GB/s)" % (t, (x.numel() * 4 + B * 640 * 640 * 16) / t * 1e-3))
    import ctypes
    from mm_masking_amd import _lib
    gz = rnd(B, 640, 640, 8)
    dw = torch.zeros(8, 1, 3, 3, device=DEV); dbb = torch.zeros(8, device=DEV)
    vp = ctypes.c_void_p
    t = timeit(lambda: uh.conv_first_wgrad(x, gz, None, dw, dbb))
    print("conv_first_wgrad us %.1f  (%.0f GB/s)" % (t, (x.numel() * 4 + B * 640 * 640 * 16) / t * 1e-3))
    for H, C in [(640, 16), (320, 32), (160, 64), (80, 128), (40, 256)]:
        d = rnd(B, H, H, C)
        t_p = timeit(lambda: uh.maxpool2(d))
        gy = rnd(B, H // 2, H // 2, C)
        t_pb = timeit(lambda: uh.maxpool2_bwd(d, gy, 1.05))
        print("pool %4d c%-3d fwd %.1f us (%.0f GB/s)  bwd %.1f us (%.0f GB/s)" % (
            H, C, t_p, d.numel() * 2 * 1.25 / t_p * 1e-3, t_pb, d.numel() * 2 * 2.25 / t_pb * 1e-3))
    for H, C in [(640, 16), (320, 32), (160, 64), (80, 128), (40, 256)]:
        s = rnd(B, H // 2, H // 2, C)
        t_u = timeit(lambda: uh.upsample(s, H, H))
        gy = rnd(B, H, H, C)
        t_ub = timeit(lambda: uh.upsample_bwd(gy, H // 2, H // 2, relu_src=s, scale=1.05))
        print("up   %4d c%-3d fwd %.1f us (%.0f GB/s)  bwd %.1f us (%.0f GB/s)" % (
            H, C, t_u, gy.numel() * 2 * 1.25 / t_u * 1e-3, t_ub, gy.numel() * 2 * 1.5 / t_ub * 1e-3))


if __name__ == "__main__":
    main()
