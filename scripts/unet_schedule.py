"""The launches of one U-Net forward + backward pass (mmk_unet_forward / mmk_unet_backward, csrc/mmk_unet_driver.hip; default
network: ReLU, dropout, amax-normalised mask, cin = 1) in program order, each with its layer, role, algorithmic bytes
(every tensor it must read or write once, NHWC bf16 unless noted) and FLOPs (2 x 9 x cin x cout per output pixel for a
3x3 convolution; data gradient and weight gradient the same).  With MMK_UNET_SIDE_STREAM=0 this is the order of the
dispatches in a rocprofv3 trace: scripts/unet_layers_report.py zips the two and checks the kernel names."""

ENC = [8, 16, 32, 64, 128, 256]


def is_deep(cin, cout):
    return cin >= 64 or (cin == 32 and cout >= 64)


def conv_kernel(cin, cout):
    return "conv3x3_deep_kernel" if is_deep(cin, cout) else "conv3x3_ring_kernel"


def wgrad_kernel(cin, cout, c1):
    return "conv3x3_wgrad_deep_kernel" if (cin % 64 == 0 and cout % 64 == 0 and c1 % 64 == 0) else "conv3x3_wgrad_kernel"


def schedule(B=32, H=640, W=640, cin0=1):
    rh = [H >> i for i in range(6)]
    rw = [W >> i for i in range(6)]
    out = []

    def px(i):
        return B * rh[i] * rw[i]

    def add(kernel, layer, role, rd, wr, flop=0.0):
        out.append({"kernel": kernel, "layer": layer, "role": role, "read_bytes": int(rd), "write_bytes": int(wr), "flop": float(flop)})

    def split_images(lvl, cin, cout):
        """dispatch_conv_deep's sub-batch split (csrc/mmk_unet.hip, round 4): forward launches of >= 64-channel layers whose
        tiles fill 1.25 rounds of the chip run as two launches; -> images of the first launch (0: one launch)."""
        if not is_deep(cin, cout) or cout % 64 != 0:
            return 0
        BM = 128 if cout >= 128 else 64
        NT = 3 if rw[lvl] <= 48 else 5
        TH = 4 if BM >= 128 else 8

        def rounds(b, bm, th):
            tpi = -(-rw[lvl] // (NT * 16)) * -(-rh[lvl] // th)
            groups = -(-cout // bm)
            nb = max(1, 32 // groups)
            return -(-(-(-b * tpi // 8)) // nb), tpi, groups, nb
        r, tpi, groups, nb = rounds(B, BM, TH)
        if r != 2 or B * tpi * groups * 10 >= 256 * 2 * 7:
            return 0
        bm = (8 * nb) // tpi
        if bm < 1 or bm >= B or rounds(bm, BM, TH)[0] != 1 or rounds(B - bm, 32, 8)[0] != 1:
            return 0
        return bm

    def conv(layer, role, lvl, cin, cout, extra_rd=0, extra_wr=0):
        n = px(lvl)
        bm = split_images(lvl, cin, cout) if role.startswith("fwd") else 0
        if bm:
            for part, frac in (("first %d images" % bm, bm / B), ("last %d images, 32-channel blocks" % (B - bm), (B - bm) / B)):
                add(conv_kernel(cin, cout), layer + " [" + part + "]", role, (n * cin * 2 + extra_rd) * frac, (n * cout * 2 + extra_wr) * frac,
                    2.0 * 9 * cin * cout * n * frac)
            return
        add(conv_kernel(cin, cout), layer, role, n * cin * 2 + extra_rd, n * cout * 2 + extra_wr, 2.0 * 9 * cin * cout * n)

    # ---------------- forward
    add("pack_conv_weights_batch_kernel", "all", "pack weights", 1769905 * 4, 1769905 * 2)
    add("conv_first_x4_kernel", "enc0.0", "fwd", px(0) * cin0 * 4, px(0) * 8 * 2, 2.0 * 9 * cin0 * 8 * px(0))
    conv("enc0.2", "fwd", 0, 8, 8)
    for i in range(1, 6):
        conv("enc%d.0" % i, "fwd", i - 1, ENC[i - 1], ENC[i])
        fused_pool = ENC[i] in (16, 32)
        codes = px(i) * ENC[i] // 2         # arg-max codes of the pooling: one nibble per pooled element (round 4)
        if fused_pool:
            # the pooling second convolution writes the pooled tensor + codes and NOT its full-resolution output
            n = px(i - 1)
            add(conv_kernel(ENC[i], ENC[i]), "enc%d.2" % i, "fwd+pool (pooled + codes only)", n * ENC[i] * 2, px(i) * ENC[i] * 2 + codes,
                2.0 * 9 * ENC[i] * ENC[i] * n)
        else:
            conv("enc%d.2" % i, "fwd", i - 1, ENC[i], ENC[i])
            add("maxpool2_fwd_kernel", "enc%d.pool" % i, "fwd", px(i - 1) * ENC[i] * 2, px(i) * ENC[i] * 2 + codes)
    for j in range(5):
        cs, lvl = ENC[4 - j], 4 - j
        add("upsample_fwd_kernel", "dec%d.up" % j, "fwd", px(lvl + 1) * 2 * cs * 2, px(lvl) * 2 * cs * 2)
        conv("dec%d.0 (1st application)" % j, "fwd", lvl, 2 * cs, cs)
        conv("dec%d.2 (1st application)" % j, "fwd", lvl, cs, cs)
        conv("dec%d.0 (2nd application, skip | d1)" % j, "fwd", lvl, 2 * cs, cs)
        conv("dec%d.2 (2nd application)" % j, "fwd", lvl, cs, cs)
    add("final_fwd_kernel", "final", "fwd", px(0) * 8 * 2, px(0) * 4, 2.0 * 8 * px(0))
    add("mask_segmax_kernel", "mask amax", "fwd", px(0) * 4, 0)
    add("mask_scale_kernel", "mask / amax", "fwd", px(0) * 4, px(0) * 4)
    n_fwd = len(out)
    # ---------------- backward
    add("pack_conv_weights_batch_kernel", "all", "pack weights (transposed)", 1769905 * 4, 1769905 * 2)
    add("mask_norm_bwd_partial_kernel", "mask / amax", "bwd", px(0) * 8, 0)
    add("mask_norm_bwd_final_kernel", "mask / amax", "bwd", 0, 0)
    add("final_bwd_kernel", "final", "bwd", px(0) * (8 * 2 + 4 + 4), px(0) * 8 * 2, 2.0 * 2 * 8 * px(0))
    add("final_bwd_reduce_kernel", "final", "bwd reduce", 0, 0)

    def wgrad(layer, lvl, cin, cout, c1):
        n = px(lvl)
        add(wgrad_kernel(cin, cout, c1), layer, "wgrad", n * (cin + cout) * 2, 0, 2.0 * 9 * cin * cout * n)

    def dgrad(layer, lvl, cin, cout, src_ch, acc_ch=0):
        # data gradient: reads g (cout channels) + ReLU source(s) (src_ch channels) (+ the accumulate target), writes cin channels
        n = px(lvl)
        add(conv_kernel(cout, cin), layer, "dgrad", n * (cout + src_ch + acc_ch) * 2, n * cin * 2, 2.0 * 9 * cin * cout * n)

    def fused(kname, layer, lvl, cin, cout, src_ch, extra_rd=0):
        # one launch: reads x (cin) and g (cout) once, writes dx (cin): data + weight gradient
        n = px(lvl)
        add(kname, layer, "dgrad+wgrad", n * (cin + cout) * 2 + extra_rd, n * cin * 2, 2.0 * 2 * 9 * cin * cout * n)

    for j in range(4, -1, -1):
        cs, lvl = ENC[4 - j], 4 - j
        fuse = cs in (8, 16)
        # second application of the block's second conv
        if fuse:
            fused("conv_bwd_fused_kernel<%d, %d," % (cs, cs), "dec%d.2 (2nd application)" % j, lvl, cs, cs, cs)
        else:
            wgrad("dec%d.2 (2nd application)" % j, lvl, cs, cs, cs)
            dgrad("dec%d.2 (2nd application)" % j, lvl, cs, cs, cs)
        if j == 4:
            fused("conv_bwd_fused_kernel<16, 8,", "dec4.0 (2nd application)", lvl, 16, 8, 16)
        else:
            wgrad("dec%d.0 (2nd application)" % j, lvl, 2 * cs, cs, cs)
            dgrad("dec%d.0 (2nd application)" % j, lvl, 2 * cs, cs, cs)          # (ReLU source: d1 half only)
        if fuse:
            fused("conv_bwd_fused_kernel<%d, %d," % (cs, cs), "dec%d.2 (1st application)" % j, lvl, cs, cs, cs)
        else:
            wgrad("dec%d.2 (1st application)" % j, lvl, cs, cs, cs)
            dgrad("dec%d.2 (1st application)" % j, lvl, cs, cs, cs)
        if j == 4:
            fused("conv_bwd_fused_kernel<16, 8,", "dec4.0 (1st application)", lvl, 16, 8, 0)
        else:
            wgrad("dec%d.0 (1st application)" % j, lvl, 2 * cs, cs, 2 * cs)
            dgrad("dec%d.0 (1st application)" % j, lvl, 2 * cs, cs, 0)
        add("upsample_bwd_kernel", "dec%d.up" % j, "bwd", px(lvl) * 2 * cs * 2 + (px(lvl + 1) * 2 * cs * 2 if j > 0 else 0), px(lvl + 1) * 2 * cs * 2)
    add("unpack_wgrad_batch_kernel", "dec", "reduce weight-gradient slices", 0, 0)
    for i in range(5, 0, -1):
        ch, lvl = ENC[i], i - 1
        pool_fused = False      # (round 4 tried the pooling adjoint inside the fused launch: slower, scripts/experiments/r04_pool_fused_backward.patch)
        if not pool_fused:
            add("maxpool2_bwd_arg_kernel", "enc%d.pool" % i, "bwd (by the codes)", px(i) * ch // 2 + px(i) * ch * 2, px(lvl) * ch * 2)
        if pool_fused:
            # reads x, the block's output d (for the routing) and the pooled gradient; writes dx
            add("conv_bwd_fused_kernel<16, 16, true>", "enc%d.pool + enc%d.2" % (i, i), "pool bwd + dgrad+wgrad",
                px(lvl) * (ch + ch) * 2 + px(i) * ch * 2, px(lvl) * ch * 2, 2.0 * 2 * 9 * ch * ch * px(lvl))
        elif ch in (8, 16):
            fused("conv_bwd_fused_kernel<%d, %d," % (ch, ch), "enc%d.2" % i, lvl, ch, ch, ch)
        else:
            wgrad("enc%d.2" % i, lvl, ch, ch, ch)
            dgrad("enc%d.2" % i, lvl, ch, ch, ch)
        if i == 1:
            fused("conv_bwd_fused_kernel<8, 16,", "enc1.0", lvl, 8, 16, 8, extra_rd=px(lvl) * 8 * 2)       # (+ the skip gradient it adds to)
        else:
            wgrad("enc%d.0" % i, lvl, ENC[i - 1], ch, ENC[i - 1])
            dgrad("enc%d.0" % i, lvl, ENC[i - 1], ch, 0, acc_ch=ENC[i - 1])
        if i == 3:
            add("unpack_wgrad_batch_kernel", "enc3-5", "reduce weight-gradient slices", 0, 0)
    fused("conv_bwd_fused_kernel<8, 8,", "enc0.2", 0, 8, 8, 8)
    add("unpack_wgrad_batch_kernel", "enc0-2", "reduce weight-gradient slices", 0, 0)
    add("conv_first_wgrad_x4_kernel", "enc0.0", "wgrad", px(0) * (cin0 * 4 + 8 * 2), 0, 2.0 * 9 * cin0 * 8 * px(0))
    add("conv_first_wgrad_reduce_kernel", "enc0.0", "wgrad reduce", 0, 0)
    return out, n_fwd


if __name__ == "__main__":
    sch, nf = schedule()
    print(len(sch), "launches,", nf, "forward")
    print("forward GFLOP/sample %.2f" % (sum(e["flop"] for e in sch[:nf]) / 32 / 1e9))
    print("backward GFLOP/sample %.2f" % (sum(e["flop"] for e in sch[nf:]) / 32 / 1e9))
    print("bytes fwd %.2f GB  bwd %.2f GB" % (sum(e["read_bytes"] + e["write_bytes"] for e in sch[:nf]) / 1e9,
                                           sum(e["read_bytes"] + e["write_bytes"] for e in sch[nf:]) / 1e9))
