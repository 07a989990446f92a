"""Batch-norm variant of the mask U-Net on the hand-written kernels (``params["batch_norm"] = True``:
mm_masking/icp_weight_policy.py:104-125 puts an nn.BatchNorm2d behind the ReLU of each convolution:
Conv, ReLU, BN, Conv, ReLU, BN, [Dropout], [MaxPool]).

The convolutions, pooling, up-sampling and the final layer are the kernels of unet_hip.py; the BatchNorm is
three streaming kernels of its own (csrc/mmk_unet.hip: mmk_bn_forward_stats / mmk_bn_apply / mmk_bn_backward).
Because a BatchNorm sits between a ReLU and the next convolution, nothing can be folded into the convolution
epilogues here: the data-gradient kernels run without a ReLU source and ``mmk_bn_backward`` forms the gradient
with respect to the convolution's pre-activation (BatchNorm adjoint x ReLU factor; the block's dropout mask is
read off the stored output: dropped = +0.0, a kept zero = -0.0).  This is the non-default configuration of the
reference (train_icp_weights.py:381 sets batch_norm False) and is scheduled launch by launch from Python.
"""
import torch

from . import _lib
from . import unet_hip as uh

BF16 = torch.bfloat16
_p, _sp = uh._p, uh._sp


def _blocks(module):
    return list(module.encoder) + list(module.decoder)


def param_list(module):
    """Per block: conv A weight, bias, BN A weight, bias, conv B weight, bias, BN B weight, bias; then the final layer."""
    out = []
    for blk in _blocks(module):
        out += [blk[0].weight, blk[0].bias, blk[2].weight, blk[2].bias, blk[3].weight, blk[3].bias, blk[5].weight, blk[5].bias]
    out += [module.final_layer[0].weight, module.final_layer[0].bias]
    return out


def _bn_forward(a, bn, gamma, beta, training, drop_p=0.0, seed=0):
    """a (B,H,W,C) bf16 -> (y, stat or None, affine)."""
    L = _lib.lib()
    B, H, W, C = a.shape
    npix = B * H * W
    dev = a.device
    affine = torch.empty(C, 2, dtype=torch.float32, device=dev)
    stat = None
    if training or bn.running_mean is None:
        stat = torch.empty(C, 2, dtype=torch.float32, device=dev)
        part = torch.empty(512 * C * 2, dtype=torch.float32, device=dev)
        track = training and bn.track_running_stats and bn.running_mean is not None
        mom = 0.1 if bn.momentum is None else float(bn.momentum)
        if track:
            bn.num_batches_tracked += 1
            if bn.momentum is None:
                mom = 1.0 / float(bn.num_batches_tracked)
        _lib.check(L.mmk_bn_forward_stats(_p(a), npix, C, _p(gamma), _p(beta), float(bn.eps), mom,
                                          _p(bn.running_mean) if track else None, _p(bn.running_var) if track else None,
                                          _p(part), _p(stat), _p(affine), _sp(dev)))
    else:
        scale = gamma / torch.sqrt(bn.running_var + bn.eps)
        affine = torch.stack((scale, beta - bn.running_mean * scale), dim=1).float().contiguous()
    y = torch.empty_like(a)
    _lib.check(L.mmk_bn_apply(_p(a), npix, C, _p(affine), float(drop_p), int(seed) & 0xFFFFFFFF, _p(y), _sp(dev)))
    return y, stat, affine


def _bn_backward(gd, d, drop_scale, a, stat, affine, gamma, slope, dgamma, dbeta, accumulate):
    L = _lib.lib()
    B, H, W, C = a.shape
    dev = a.device
    part = torch.empty(512 * C * 2, dtype=torch.float32, device=dev)
    coef = torch.empty(C * 3, dtype=torch.float32, device=dev)
    gz = torch.empty_like(a)
    _lib.check(L.mmk_bn_backward(_p(gd), _p(d), float(drop_scale), _p(a), B * H * W, C, _p(stat), _p(affine), _p(gamma), float(slope),
                                 1 if accumulate else 0, _p(part), _p(coef), _p(dgamma), _p(dbeta), _p(gz), _sp(dev)))
    return gz


class _UNetBN(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pre, drop_p, seed, training, norm, slope, module, *params):
        dev = x.device
        x = x.contiguous().float()
        B, cin, H, W = x.shape
        P = [p.detach().float().contiguous() for p in params]
        blocks = _blocks(module)
        p_drop = float(drop_p) if training else 0.0
        sl = float(slope)
        ctr = [int(seed) * 64]

        def next_seed():
            ctr[0] += 1
            return ctr[0]

        def bp(k):         # parameters of block k
            return P[8 * k:8 * k + 8]

        # packed weights of every 3x3 conv except the very first (fp32 first layer)
        ws = []
        for k in range(11):
            ws += [bp(k)[0], bp(k)[4]]
        packs = uh.pack_weights_batch(ws[1:])
        pk = {i + 1: t for i, t in enumerate(packs)}          # conv index 2k (A), 2k+1 (B)

        def block(k, x1, x2=None, first=None):
            wA, bA, gA, beA, wB, bB, gB, beB = bp(k)
            blk = blocks[k]
            cout = wA.shape[0]
            if first is not None:
                aA = uh.conv_first(first, wA, bA, pre, slope=sl)
            else:
                aA = uh.conv3x3(x1, pk[2 * k], cout, bias=bA, x2=x2, relu=True, slope=sl)
            yA, sA, afA = _bn_forward(aA, blk[2], gA, beA, training)
            aB = uh.conv3x3(yA, pk[2 * k + 1], cout, bias=bB, relu=True, slope=sl)
            d, sB, afB = _bn_forward(aB, blk[5], gB, beB, training, p_drop, next_seed())
            return d, (x1, x2, aA, yA, aB, d, sA, afA, sB, afB)

        saved = {}
        d, saved[("e", 0)] = block(0, None, first=x)
        t = [d]
        for i in range(1, 6):
            d, saved[("e", i)] = block(i, t[i - 1])
            t.append(uh.maxpool2(d))
        cur = t[5]
        for j in range(5):
            skip = t[4 - j]
            u = uh.upsample(cur, skip.shape[1], skip.shape[2])
            d1, saved[("d", j, 0)] = block(6 + j, u)
            d2, saved[("d", j, 1)] = block(6 + j, skip, x2=d1)
            cur = d2
        wf, bf = P[88], P[89]
        mask = uh.final_fwd(cur, wf.reshape(8).contiguous(), bf)
        if uh.DEBUG is not None:
            uh.DEBUG["fwd_bn"] = {"t": t, "saved": saved}
        ctx.x, ctx.pre, ctx.t, ctx.saved, ctx.P, ctx.module = x, pre, t, saved, P, module
        ctx.mask = mask.detach()
        ctx.scale = uh.dropout_scale(p_drop)
        ctx.slope = sl
        ctx.training = bool(training)
        ctx.norm = bool(norm)
        if ctx.norm:
            part = torch.empty(B * 64, dtype=torch.float32, device=dev)
            mask_n = torch.empty_like(mask)
            amax = torch.empty(B, dtype=torch.float32, device=dev)
            _lib.check(_lib.lib().mmk_mask_normalize(_p(mask), B, H * W, _p(part), _p(mask_n), _p(amax), _sp(dev)))
            ctx.mask_n, ctx.amax = mask_n.detach(), amax
            return mask_n
        return mask

    @staticmethod
    def backward(ctx, gmask):
        L = _lib.lib()
        x, t, P, saved = ctx.x, ctx.t, ctx.P, ctx.saved
        dev = x.device
        s, sl = ctx.scale, ctx.slope
        B = x.shape[0]
        gmask = gmask.contiguous().float()
        grads = [torch.zeros_like(p) for p in P]

        def bp(k):
            return P[8 * k:8 * k + 8]

        ws = []
        for k in range(11):
            ws += [bp(k)[0], bp(k)[4]]
        packs_t = uh.pack_weights_batch(ws[1:], transposed=True)
        pkt = {i + 1: tt for i, tt in enumerate(packs_t)}
        part, part_sets = {}, {}

        def wgrad(ci, x1, g, x2=None):
            w = ws[ci]
            cout_k, cin_k = w.shape[0], w.shape[1]
            ns = uh.wgrad_slices(cout_k, cin_k, x1.shape[3], x1.shape[0], x1.shape[1], x1.shape[2])
            if ns <= 0:
                raise _lib.MmkError("U-Net backward: no weight-gradient kernel for %d -> %d channels" % (cin_k, cout_k))
            # (a decoder convolution is applied twice: each application writes its own set of slices)
            if ci not in part:
                part[ci] = uh.partial_buffer(ns * (2 if ci >= 12 else 1), cout_k, cin_k, dev)
                part_sets[ci] = 0
            u = part_sets[ci]
            part_sets[ci] = u + 1
            uh.conv3x3_wgrad_partial(x1, g, cout_k, part[ci][u * ns:(u + 1) * ns], x2=x2)

        seen = set()

        def block_bwd(k, key, g_d, want_split=None, out=None, accumulate_out=False):
            """g_d: gradient w.r.t. the block's output d.  Returns the gradient(s) w.r.t. the block's input."""
            x1, x2, aA, yA, aB, d, sA, afA, sB, afB = saved[key]
            wA, bA, gA, beA, wB, bB, gB, beB = bp(k)
            acc = k in seen
            seen.add(k)
            gzB = _bn_backward(g_d, d, s, aB, sB if ctx.training else None, afB, gB, sl, grads[8 * k + 6], grads[8 * k + 7], acc)
            wgrad(2 * k + 1, yA, gzB)
            g_yA = uh.conv3x3(gzB, pkt[2 * k + 1], wB.shape[1], slope=sl)
            gzA = _bn_backward(g_yA, None, 1.0, aA, sA if ctx.training else None, afA, gA, sl, grads[8 * k + 2], grads[8 * k + 3], acc)
            if k == 0:
                uh.conv_first_wgrad(x, gzA, ctx.pre, grads[0], grads[1])
                return None
            wgrad(2 * k, x1, gzA, x2=x2)
            cin_k = wA.shape[1]
            if want_split is not None:
                return uh.conv3x3(gzA, pkt[2 * k], cin_k, split=want_split, slope=sl)
            if out is not None:
                uh.conv3x3(gzA, pkt[2 * k], cin_k, out=out, accumulate=accumulate_out, slope=sl)
                return out
            return uh.conv3x3(gzA, pkt[2 * k], cin_k, slope=sl)

        # ---- final layer: gradient w.r.t. d2 of the last decoder block (dropout mask only: scale 1, "slope 1")
        d2_4 = saved[("d", 4, 1)][5]
        wf8 = P[88].reshape(8).contiguous()
        g_fw, g_fb = grads[88].view(-1), grads[89]
        gz = torch.empty_like(d2_4)
        red = torch.empty(_lib.FINAL_BWD_WS_FLOATS, dtype=torch.float32, device=dev)
        if ctx.norm:
            npix = gmask.shape[1] * gmask.shape[2]
            wsb = torch.empty(B * 130, dtype=torch.float32, device=dev)
            _lib.check(L.mmk_final_bwd_normalized(_p(d2_4), _p(wf8), _p(ctx.mask), _p(ctx.mask_n), _p(ctx.amax), _p(gmask), B, npix,
                                                  1.0, 1.0, _p(wsb[:B * 128]), _p(wsb[B * 128:]), _p(gz), _p(g_fw), _p(g_fb), _p(red),
                                                  _sp(dev)))
        else:
            _lib.check(L.mmk_final_bwd(_p(d2_4), _p(wf8), _p(ctx.mask), _p(gmask), gmask.numel(), 1.0, 1.0, _p(gz), _p(g_fw), _p(g_fb),
                                       _p(red), _sp(dev)))
        # ---- decoder
        g_skip = [None] * 5
        g_cur = gz
        for j in range(4, -1, -1):
            cs = t[4 - j].shape[3]
            gsk, g_d1 = block_bwd(6 + j, ("d", j, 1), g_cur, want_split=cs)
            g_skip[4 - j] = gsk
            g_u = block_bwd(6 + j, ("d", j, 0), g_d1)
            if j > 0:
                pd2 = saved[("d", j - 1, 1)][5]
                g_cur = uh.upsample_bwd(g_u, pd2.shape[1], pd2.shape[2])
            else:
                g_cur = uh.upsample_bwd(g_u, t[5].shape[1], t[5].shape[2])
        # ---- encoder
        g_t = g_cur
        for i in range(5, 0, -1):
            d_i = saved[("e", i)][5]
            g_d = uh.maxpool2_bwd(d_i, g_t, 1.0, 1.0)          # routing + "not dropped" factor; the BN adjoint applies 1/keep
            block_bwd(i, ("e", i), g_d, out=g_skip[i - 1], accumulate_out=True)
            g_t = g_skip[i - 1]
        block_bwd(0, ("e", 0), g_t)
        # ---- conv weight / bias gradients of the 21 3x3 layers
        assert all(part_sets[ci] == (2 if ci >= 12 else 1) for ci in range(1, 22))
        items, idxs = [], []
        for ci in range(1, 22):
            w = ws[ci]
            k, which = ci // 2, ci % 2
            gb = grads[8 * k + (5 if which else 1)]
            items.append((part[ci], w.shape[0], w.shape[1], gb))
            idxs.append(8 * k + (4 if which else 0))
        for gi, gw in zip(idxs, uh.wgrad_unpack_batch(items)):
            grads[gi] = gw
        grads[88] = g_fw.view(1, 8, 1, 1)
        return (None,) * 8 + tuple(g.to(p.dtype) for g, p in zip(grads, P))


def unet_mask(module, x, training, seed, norm=False, pre=None, slope=0.0):
    return _UNetBN.apply(x, pre, float(module.dropout), int(seed), bool(training), bool(norm), float(slope), module, *param_list(module))
