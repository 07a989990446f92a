"""A/B of one library switch in ONE process: every >= 64-channel forward / data-gradient launch shape of the network at the bench
size is run with the environment variable VAR = "0" (column "deep") and VAR = "1" (column "dx"), interleaved; bit-identity of the
outputs and median timings.  Development tool (run on the GPU box):  python scripts/ab_env.py MMK_CONV_SPLIT [B] [rounds]"""
import os
import sys

import torch

sys.path.insert(0, ".")
from mm_masking_amd import unet_hip as uh  # noqa: E402

DEV = torch.device("cuda:0")
VAR = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
ROUNDS = int(sys.argv[3]) if len(sys.argv) > 3 else 5


def rnd(*shape):
    return (torch.randn(*shape, device=DEV) * 0.5).to(torch.bfloat16)


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


def main():
    # (name, H, cin1, cin2, cout)
    layers = [("enc3.2", 160, 64, 0, 64), ("enc4.0", 80, 64, 0, 128), ("enc4.2", 80, 128, 0, 128), ("enc5.0", 40, 128, 0, 256),
              ("enc5.2", 40, 256, 0, 256), ("dec0.0u", 40, 256, 0, 128), ("dec0.2", 40, 128, 0, 128), ("dec0.0c", 40, 128, 128, 128),
              ("dec1.0u", 80, 128, 0, 64), ("dec1.2", 80, 64, 0, 64), ("dec1.0c", 80, 64, 64, 64), ("enc3.0", 160, 32, 0, 64)]
    tot = {}
    print("%-8s %4s %9s | %-5s %8s %8s %6s %7s | %-5s %8s %8s %6s %7s" % ("layer", "H", "cin>cout", "fwd", "deep us", "dx us", "ratio", "dx TF/s",
                                                                          "dgrad", "deep us", "dx us", "ratio", "dx TF/s"))
    for name, H, c1, c2, co in layers:
        cin = c1 + c2
        x1 = rnd(B, H, H, c1)
        x2 = rnd(B, H, H, c2) if c2 else None
        w = torch.randn(co, cin, 3, 3, device=DEV) / (3 * cin ** 0.5)
        bias = torch.randn(co, device=DEV) * 0.1
        wp, wpt = uh.pack_weights(w), uh.pack_weights(w, transposed=True)
        g = rnd(B, H, H, co)
        flop = 2.0 * 9 * cin * co * H * H * B

        def fwd(out):
            return uh.conv3x3(x1, wp, co, bias=bias, x2=x2, relu=True, drop_p=0.05, seed=3, out=out)

        def dgrad(outs):
            if c2:
                return uh.conv3x3(g, wpt, cin, split=c1, out=outs[0], out2=outs[1], relu_src2=x2, scale2=1.05)
            return uh.conv3x3(g, wpt, cin, out=outs[0], relu_src=x1, scale=1.05)

        res = {}
        for mode in ("0", "1"):
            os.environ[VAR] = mode
            y = torch.zeros(B, H, H, co, dtype=torch.bfloat16, device=DEV)
            fwd(y)
            if c2:
                o = (torch.zeros(B, H, H, c1, dtype=torch.bfloat16, device=DEV), torch.zeros(B, H, H, c2, dtype=torch.bfloat16, device=DEV))
            else:
                o = (torch.zeros(B, H, H, c1, dtype=torch.bfloat16, device=DEV),)
            dgrad(o)
            torch.cuda.synchronize()
            res[mode] = (y, o)
        same_f = torch.equal(res["0"][0], res["1"][0])
        same_d = all(torch.equal(p, q) for p, q in zip(res["0"][1], res["1"][1]))
        tf = {"0": [], "1": []}
        td = {"0": [], "1": []}
        for _ in range(ROUNDS):
            for mode in ("0", "1"):
                os.environ[VAR] = mode
                tf[mode].append(timeit(lambda: fwd(res[mode][0])))
                td[mode].append(timeit(lambda: dgrad(res[mode][1])))
        f0, f1, d0, d1 = (sorted(v)[len(v) // 2] for v in (tf["0"], tf["1"], td["0"], td["1"]))
        for k, v in (("f0", f0), ("f1", f1), ("d0", d0), ("d1", d1)):
            tot[k] = tot.get(k, 0.0) + v
        print("%-8s %4d %4d>%-4d | %-5s %8.1f %8.1f %6.2f %7.0f | %-5s %8.1f %8.1f %6.2f %7.0f" % (
            name, H, cin, co, "same" if same_f else "DIFF", f0, f1, f1 / f0, flop / f1 * 1e-6,
            "same" if same_d else "DIFF", d0, d1, d1 / d0, flop / d1 * 1e-6), flush=True)
        if not (same_f and same_d):
            a, b = res["0"][0].float(), res["1"][0].float()
            print("   fwd max |diff| %.4g, differing elements %d of %d" % ((a - b).abs().max().item(), int((a != b).sum()), a.numel()))
    print("sum us: fwd deep %.0f dx %.0f | dgrad deep %.0f dx %.0f" % (tot["f0"], tot["f1"], tot["d0"], tot["d1"]))
    os.environ.pop(VAR, None)


if __name__ == "__main__":
    main()
