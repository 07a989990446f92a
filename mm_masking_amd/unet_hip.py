"""Hand-written gfx950 path of the mask U-Net (csrc/mmk_unet.hip through the C ABI).

Low-level wrappers (NHWC bf16 tensors) + the fused forward/backward of the reference's
default network configuration (ReLU, no batch norm; mm_masking/icp_weight_policy.py:84-99,
161-184).  The parameters stay in the module's nn.Conv2d objects (state_dict
compatibility); every step packs them to bf16 MFMA-fragment order on the device.
"""
import ctypes
import os

import torch

from . import _lib

BF16 = torch.bfloat16


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def pack_weights(w, transposed=False):
    """fp32 (COUT,CIN,3,3) -> packed bf16 for mmk_conv3x3 (transposed: data-gradient operator)."""
    L = _lib.lib()
    cout, cin = w.shape[0], w.shape[1]
    n = L.mmk_conv3x3_packed_elems(cout, cin, 1 if transposed else 0)
    if n == 0:
        raise _lib.MmkError("unsupported conv channel counts %d -> %d" % (cin, cout))
    out = torch.empty(n, dtype=BF16, device=w.device)
    wf = w.detach().float().contiguous()
    _lib.check(L.mmk_conv3x3_pack_weights(_lib.ptr(wf), cout, cin, 1 if transposed else 0, _lib.ptr(out),
                                          _lib.stream_ptr(w.device)))
    return out


def pack_weights_batch(ws, transposed=False):
    """Pack a list of fp32 (COUT,CIN,3,3) weights in one launch -> list of packed bf16 views of one buffer."""
    L = _lib.lib()
    n = len(ws)
    tr = 1 if transposed else 0
    sizes = [L.mmk_conv3x3_packed_elems(w.shape[0], w.shape[1], tr) for w in ws]
    if any(sz == 0 for sz in sizes):
        raise _lib.MmkError("unsupported conv channel counts in pack_weights_batch")
    offs = [0]
    for sz in sizes:
        offs.append(offs[-1] + (sz + 127) // 128 * 128)
    buf = torch.empty(offs[-1], dtype=BF16, device=ws[0].device)
    base = buf.data_ptr()
    wsrc = [w.detach().float().contiguous() for w in ws]
    Wp = (ctypes.c_void_p * n)(*[w.data_ptr() for w in wsrc])
    Op = (ctypes.c_void_p * n)(*[base + 2 * o for o in offs[:-1]])
    co = (ctypes.c_int32 * n)(*[w.shape[0] for w in ws])
    ci = (ctypes.c_int32 * n)(*[w.shape[1] for w in ws])
    _lib.check(L.mmk_conv3x3_pack_weights_batch(n, Wp, co, ci, tr, Op, _lib.stream_ptr(ws[0].device)))
    return [buf[offs[i]:offs[i] + sizes[i]] for i in range(n)]


def conv3x3(x1, wpack, cout, bias=None, x2=None, relu=False, drop_p=0.0, seed=0, out=None, out2=None, split=None,
            relu_src=None, scale=1.0, relu_src2=None, scale2=1.0, accumulate=False, accumulate2=False, pool_out=None,
            slope=0.0, pool_arg=None):
    """3x3 / pad 1 convolution on NHWC bf16.  input = concat(x1, x2); output channels
    [0,split) -> out, [split,cout) -> out2 (split=None: single output).  ``pool_out`` (B,H//2,W//2,cout):
    the 2x2 max-pool of the output, written by the same pass (layers for which pool_fusable() holds).
    ``slope`` > 0: the LeakyReLU variant (forward activation and the relu_src factors, see include/mmk.h).
    ``pool_arg`` (B,H//2,W//2,cout//2) uint8, with ``pool_out``: the arg-max codes of the pooling windows; the
    full-resolution output is then NOT written and None is returned (include/mmk.h: mmk_conv_desc.pool_arg)."""
    B, H, W, C1 = x1.shape
    C2 = 0 if x2 is None else x2.shape[3]
    O1 = cout if split is None else split
    O2 = cout - O1
    if out is None and pool_arg is None:
        out = torch.empty(B, H, W, O1, dtype=BF16, device=x1.device)
    if O2 > 0 and out2 is None:
        out2 = torch.empty(B, H, W, O2, dtype=BF16, device=x1.device)
    d = _lib.ConvDesc(x1=_p(x1), x2=_p(x2), C1=C1, C2=C2, wpack=_p(wpack), bias=_p(bias), y1=_p(out),
                      relu_src1=_p(relu_src), O1=O1, accumulate1=1 if accumulate else 0, scale1=float(scale),
                      y2=_p(out2), relu_src2=_p(relu_src2), O2=O2, accumulate2=1 if accumulate2 else 0,
                      scale2=float(scale2), B=B, H=H, W=W, relu=1 if relu else 0, leaky_slope=float(slope),
                      drop_p=float(drop_p),
                      seed=int(seed) & 0xFFFFFFFF, pool_y=_p(pool_out), pool_arg=_p(pool_arg))
    _lib.check(_lib.lib().mmk_conv3x3(ctypes.byref(d), _lib.stream_ptr(x1.device)))
    if pool_arg is not None:
        return None
    return (out, out2) if O2 > 0 else out


def pool_fusable(cin, cout, B, H, W):
    return bool(_lib.lib().mmk_conv3x3_pool_fusable(cin, cout, B, H, W))


def conv3x3_wgrad(x1, g, cout, x2=None, dWt=None, db=None):
    """Weight / bias gradient added to dWt (9,cout,cin) fp32 and db (cout,) fp32 (a fresh zero dWt when none is given):
    the partial-sum kernel (per-workgroup slices, plain stores) followed by a sum over the slices -- the form every layer
    of the network uses; there is no float-atomic path."""
    B, H, W, C1 = x1.shape
    C2 = 0 if x2 is None else x2.shape[3]
    cin = C1 + C2
    ns = wgrad_slices(cout, cin, C1, B, H, W)
    if ns <= 0:
        raise _lib.MmkError("conv3x3_wgrad: unsupported shape %d -> %d channels" % (cin, cout))
    part = partial_buffer(ns, cout, cin, x1.device)
    conv3x3_wgrad_partial(x1, g, cout, part, x2=x2)
    tot = part.sum(dim=0)
    w = tot[:9 * cout * cin].view(9, cout, cin)
    if dWt is None:
        dWt = w.clone()
    else:
        dWt += w
    if db is not None:
        db += tot[9 * cout * cin:]
    return dWt


def wgrad_unpack(dWt, accumulate_into=None):
    """(9,cout,cin) -> (cout,cin,3,3); accumulates into an existing gradient when given."""
    w = dWt.permute(1, 2, 0).reshape(dWt.shape[1], dWt.shape[2], 3, 3)
    if accumulate_into is not None:
        accumulate_into += w
        return accumulate_into
    return w.contiguous()


def dropout_scale(p):
    """1/keep as the kernels apply it: the drop probability is quantised to 16 bits
    (csrc/mmk_unet.hip: dropout_params), and the scale is the exact inverse of that keep rate."""
    thr = int(float(p) * 65536.0 + 0.5)
    return 65536.0 / (65536 - thr) if thr else 1.0


def wgrad_slices(cout, cin, c1, B, H, W):
    """Number of per-workgroup partial slices of the partial-sum weight-gradient kernel (0: unsupported)."""
    return int(_lib.lib().mmk_conv3x3_wgrad_slices(cout, cin, c1, B, H, W))


def partial_buffer(ns, cout, cin, device):
    """(slices, 9*cout*cin + cout) fp32: per slice the (9,cout,cin) weight sums, then the cout bias sums."""
    return torch.empty(ns, 9 * cout * cin + cout, dtype=torch.float32, device=device)


def conv3x3_wgrad_partial(x1, g, cout, partials, x2=None, accumulate=False):
    """Weight + bias gradient as per-workgroup partial sums into `partials` (see partial_buffer)."""
    B, H, W, C1 = x1.shape
    C2 = 0 if x2 is None else x2.shape[3]
    _lib.check(_lib.lib().mmk_conv3x3_wgrad_partial(_p(x1), _p(x2), C1, C2, _p(g), cout, B, H, W, _p(partials),
                                                    1 if accumulate else 0, _lib.stream_ptr(x1.device)))
    return partials


def conv_bwd_fused(x, g, wpack_t, scale, dx, partials, accumulate=False):
    """Data gradient (ReLU source = x) and partial weight-gradient slices of a C -> C convolution (C = 8, 16) in one launch:
    bit-identical to conv3x3(g, wpack_t, C, out=dx, relu_src=x, scale=scale) + conv3x3_wgrad_partial(x, g, C, partials)."""
    B, H, W, C = x.shape
    assert C in (8, 16) and g.shape == x.shape and dx.shape == x.shape
    _lib.check(_lib.lib().mmk_conv_bwd_fused(_p(x), _p(g), _p(wpack_t), float(scale), B, H, W, C, _p(dx), _p(partials),
                                             1 if accumulate else 0, _lib.stream_ptr(x.device)))
    return dx, partials


def conv8x16_bwd_fused(x, g, wpack_t, scale, dx, partials, accumulate=False):
    """dx += ((x > 0) ? scale : 0) * conv_T(g) and the partial weight-gradient slices of an 8 -> 16 convolution in one launch:
    bit-identical to conv3x3(g, wpack_t, 8, out=dx, relu_src=x, scale=scale, accumulate=True) + conv3x3_wgrad_partial(x, g, 16, partials)."""
    B, H, W, C = x.shape
    assert C == 8 and g.shape == (B, H, W, 16) and dx.shape == x.shape
    _lib.check(_lib.lib().mmk_conv8x16_bwd_fused(_p(x), _p(g), _p(wpack_t), float(scale), B, H, W, _p(dx), _p(partials),
                                                 1 if accumulate else 0, _lib.stream_ptr(x.device)))
    return dx, partials


def conv16x8_bwd_fused(x1, x2, g, wpack_t, scale, dx1, dx2, partials, accumulate=False):
    """Backward of a 16 -> 8 convolution on concat(x1, x2) in one launch: the two halves of its data gradient, each masked by its
    own input activation, and the partial weight-gradient slices; bit-identical to conv3x3(g, wpack_t, 16, out=dx1, out2=dx2,
    split=8, relu_src=x1, relu_src2=x2, scale=scale2=scale) + conv3x3_wgrad_partial(x1, g, 8, partials, x2=x2)."""
    B, H, W, C = x1.shape
    if x2 is None:        # one 16-channel input, one unmasked 16-channel output
        assert C == 16 and dx2 is None and g.shape == (B, H, W, 8) and dx1.shape == x1.shape
    else:
        assert C == 8 and x2.shape == x1.shape and g.shape == x1.shape and dx1.shape == x1.shape and dx2.shape == x1.shape
    _lib.check(_lib.lib().mmk_conv16x8_bwd_fused(_p(x1), _p(x2), _p(g), _p(wpack_t), float(scale), B, H, W, _p(dx1), _p(dx2),
                                                 _p(partials), 1 if accumulate else 0, _lib.stream_ptr(x1.device)))
    return dx1, dx2, partials


def wgrad_unpack_batch(items):
    """One launch for a list of layers.  Each item is a tuple (partials, cout, cin[, db_out]) of the partial-sum form (or a
    plain (9,cout,cin) tensor: transposed only); returns the (cout,cin,3,3) gradients (views of one buffer); bias sums are
    written to db_out (cout,) when given."""
    n = len(items)
    srcs, couts, cins, slices, dbs = [], [], [], [], []
    for it in items:
        if isinstance(it, (tuple, list)):
            t, co, ci = it[0], it[1], it[2]
            srcs.append(t); couts.append(co); cins.append(ci); slices.append(t.shape[0])
            dbs.append(it[3] if len(it) > 3 else None)
        else:
            srcs.append(it); couts.append(it.shape[1]); cins.append(it.shape[2]); slices.append(0); dbs.append(None)
    sizes = [9 * co * ci for co, ci in zip(couts, cins)]
    offs = [0]
    for sz in sizes:
        offs.append(offs[-1] + (sz + 3) // 4 * 4)
    dev = srcs[0].device
    buf = torch.empty(offs[-1], dtype=torch.float32, device=dev)
    base = buf.data_ptr()
    Sp = (ctypes.c_void_p * n)(*[t.data_ptr() for t in srcs])
    Op = (ctypes.c_void_p * n)(*[base + 4 * o for o in offs[:-1]])
    Dp = (ctypes.c_void_p * n)(*[(d.data_ptr() if d is not None else None) for d in dbs])
    sl = (ctypes.c_int32 * n)(*slices)
    co = (ctypes.c_int32 * n)(*couts)
    ci = (ctypes.c_int32 * n)(*cins)
    _lib.check(_lib.lib().mmk_conv3x3_wgrad_unpack_batch(n, Sp, sl, co, ci, Op, Dp, _lib.stream_ptr(dev)))
    return [buf[offs[i]:offs[i] + sizes[i]].view(couts[i], cins[i], 3, 3) for i in range(n)]


# ----------------------------------------------------------------------------- small wrappers
def _sp(dev):
    return _lib.stream_ptr(dev)


def conv_first_wgrad(x, gz, pre, dW, db):
    """First layer's weight / bias gradient WRITTEN to dW (8,cin,3,3) and db (8,) (block partials + ordered reduction)."""
    L = _lib.lib()
    B, cin, H, W = x.shape
    nb = int(L.mmk_conv_first_wgrad_ws_bytes(cin))
    ws = torch.empty(nb // 4, dtype=torch.float32, device=x.device)
    _lib.check(L.mmk_conv_first_wgrad(_p(x), cin, _p(gz), _p(pre), B, H, W, _p(dW), _p(db), _p(ws), nb, _sp(x.device)))


def conv_first(x, w, b, pre=None, slope=0.0):
    """x fp32 (B,cin,H,W) -> bf16 (B,H,W,8), + bias + ReLU (encoder.0.0).  ``pre`` (cin,2): per-channel
    (offset, reciprocal scale) applied to x while loading (see channel_minmax)."""
    B, cin, H, W = x.shape
    y = torch.empty(B, H, W, 8, dtype=BF16, device=x.device)
    _lib.check(_lib.lib().mmk_conv_first(_p(x), cin, _p(w), _p(b), _p(pre), B, H, W, float(slope), _p(y), _sp(x.device)))
    return y


FORCE_COLLECTIVES = False     # one-rank rehearsal of the N-rank path (bench.py --force-dist): issue the collective at world size 1 too


def channel_minmax(x, process_group=None, global_reduce=False):
    """(min, 1 / (max - min)) per channel of fp32 (B,C,H,W) over (B,H,W) -> (C,2) fp32: the offset and
    reciprocal scale of the policy's min-max normalisation (icp_weight_policy.py:151-155).
    ``global_reduce``: in a data-parallel job the minimum / maximum are reduced over the ranks (one MAX
    all-reduce of 2C floats), so that the normalisation stays global over the whole batch as in the
    single-process reference."""
    B, C, H, W = x.shape
    part = torch.empty(C * 2048, dtype=torch.float32, device=x.device)
    pre = torch.empty(C, 2, dtype=torch.float32, device=x.device)
    import torch.distributed as dist
    reduce = global_reduce and dist.is_available() and dist.is_initialized() and (dist.get_world_size(process_group) > 1 or FORCE_COLLECTIVES)
    mm = torch.empty(C, 2, dtype=torch.float32, device=x.device) if reduce else None
    _lib.check(_lib.lib().mmk_channel_minmax(_p(x), B, C, H * W, _p(part), _p(pre), _p(mm), _sp(x.device)))
    if reduce:
        t = torch.stack((-mm[:, 0], mm[:, 1]), dim=1)            # max(-min) = -min over the ranks
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=process_group)
        pre = torch.stack((-t[:, 0], 1.0 / (t[:, 1] + t[:, 0])), dim=1).contiguous()
    return pre


def maxpool2(x):
    B, H, W, C = x.shape
    y = torch.empty(B, H // 2, W // 2, C, dtype=BF16, device=x.device)
    _lib.check(_lib.lib().mmk_maxpool2_fwd(_p(x), B, H, W, C, _p(y), _sp(x.device)))
    return y


def maxpool2_arg(x):
    """maxpool2 plus the arg-max codes of its windows, (B,H//2,W//2,C//2) uint8 (include/mmk.h: mmk_conv_desc.pool_arg)."""
    B, H, W, C = x.shape
    y = torch.empty(B, H // 2, W // 2, C, dtype=BF16, device=x.device)
    arg = torch.empty(B, H // 2, W // 2, C // 2, dtype=torch.uint8, device=x.device)
    _lib.check(_lib.lib().mmk_maxpool2_fwd_arg(_p(x), B, H, W, C, _p(y), _p(arg), _sp(x.device)))
    return y, arg


def maxpool2_bwd_arg(arg, gy, H, W, scale):
    """maxpool2_bwd of the ReLU network from the arg-max codes instead of the full-resolution tensor: (B,H,W,C) gradient."""
    B, _, _, C = gy.shape
    gz = torch.empty(B, H, W, C, dtype=BF16, device=gy.device)
    _lib.check(_lib.lib().mmk_maxpool2_bwd_arg(_p(arg), _p(gy), B, H, W, C, float(scale), _p(gz), _sp(gy.device)))
    return gz


def maxpool2_bwd(d, gy, scale, slope=0.0):
    B, H, W, C = d.shape
    gz = torch.empty_like(d)
    _lib.check(_lib.lib().mmk_maxpool2_bwd(_p(d), _p(gy), B, H, W, C, float(scale), float(slope), _p(gz), _sp(d.device)))
    return gz


def upsample(x, Ho, Wo):
    B, Hs, Ws, C = x.shape
    y = torch.empty(B, Ho, Wo, C, dtype=BF16, device=x.device)
    _lib.check(_lib.lib().mmk_upsample_fwd(_p(x), B, Hs, Ws, C, Ho, Wo, _p(y), _sp(x.device)))
    return y


def upsample_bwd(gy, Hs, Ws, relu_src=None, scale=1.0, slope=0.0):
    B, Ho, Wo, C = gy.shape
    gx = torch.empty(B, Hs, Ws, C, dtype=BF16, device=gy.device)
    _lib.check(_lib.lib().mmk_upsample_bwd(_p(gy), B, Hs, Ws, C, Ho, Wo, _p(relu_src), float(scale), float(slope),
                                           _p(gx), _sp(gy.device)))
    return gx


def final_fwd(x, w, b):
    B, H, W, _ = x.shape
    mask = torch.empty(B, H, W, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().mmk_final_fwd(_p(x), _p(w), _p(b), B * H * W, _p(mask), _sp(x.device)))
    return mask


# ----------------------------------------------------------------------------- the fused network
DEBUG = None      # set to a dict to capture the backward pass's gradient tensors (tests/diagnostics)
ENC = ["encoder.%d" % i for i in range(6)]
DEC = ["decoder.%d" % i for i in range(5)]


def param_list(module):
    """Parameters in state_dict order: encoder.{0..5}.{0,2}, decoder.{0..4}.{0,2}, final_layer.0."""
    out = []
    for blk in list(module.encoder) + list(module.decoder):
        out += [blk[0].weight, blk[0].bias, blk[2].weight, blk[2].bias]
    out += [module.final_layer[0].weight, module.final_layer[0].bias]
    return out


class _UNet(torch.autograd.Function):
    """mask = sigmoid(final(decoder(encoder(x)))) with every layer in hand-written HIP;
    backward is the hand-scheduled reverse pass (no autograd graph inside)."""

    @staticmethod
    def forward(ctx, x, pre, drop_p, seed, training, norm, slope, *params):
        dev = x.device
        x = x.contiguous().float()
        B, cin, H, W = x.shape
        P = [p.detach() for p in params]

        def wb(k):                 # conv k: 0..11 encoder (2 per block), 12..21 decoder, 22 final
            return P[2 * k], P[2 * k + 1]

        p_drop = float(drop_p) if training else 0.0
        sl = float(slope)              # > 0: nn.LeakyReLU(slope) network (icp_weight_policy.py:106)
        ctr = [int(seed) * 64]

        def next_seed():
            ctr[0] += 1
            return ctr[0]

        packs = dict(zip(range(1, 22), pack_weights_batch([wb(k)[0] for k in range(1, 22)])))

        def pk(k):
            return packs[k]

        saved = {}
        # ---- encoder
        w0, b0 = wb(0)
        a = conv_first(x, w0.float().contiguous(), b0.float().contiguous(), pre, slope=sl)
        w1, b1 = wb(1)
        d = conv3x3(a, pk(1), 8, bias=b1, relu=True, drop_p=p_drop, seed=next_seed(), slope=sl)
        saved["e0"] = (a, d)
        t = [d]                                   # t[i] = input of encoder block i+1 / skip tensors
        ch = [8, 16, 32, 64, 128, 256]
        for i in range(1, 6):
            wa, ba = wb(2 * i)
            wc, bc = wb(2 * i + 1)
            a = conv3x3(t[i - 1], pk(2 * i), ch[i], bias=ba, relu=True, slope=sl)
            if sl == 0.0 and pool_fusable(ch[i], ch[i], B, a.shape[1], a.shape[2]):
                # the second conv writes its 2x2 max-pool as well (no re-read of the full-resolution tensor)
                pooled = torch.empty(B, a.shape[1] // 2, a.shape[2] // 2, ch[i], dtype=BF16, device=dev)
                d = conv3x3(a, pk(2 * i + 1), ch[i], bias=bc, relu=True, drop_p=p_drop, seed=next_seed(), pool_out=pooled)
                t.append(pooled)
            else:
                d = conv3x3(a, pk(2 * i + 1), ch[i], bias=bc, relu=True, drop_p=p_drop, seed=next_seed(), slope=sl)
                t.append(maxpool2(d))
            saved["e%d" % i] = (a, d)
        # ---- decoder
        cur = t[5]
        dsaved = []
        for j in range(5):
            skip = t[4 - j]
            cs = skip.shape[3]
            k0, k1 = 12 + 2 * j, 13 + 2 * j
            _, b_a = wb(k0)
            _, b_c = wb(k1)
            u = upsample(cur, skip.shape[1], skip.shape[2])
            a1 = conv3x3(u, pk(k0), cs, bias=b_a, relu=True, slope=sl)
            d1 = conv3x3(a1, pk(k1), cs, bias=b_c, relu=True, drop_p=p_drop, seed=next_seed(), slope=sl)
            a2 = conv3x3(skip, pk(k0), cs, bias=b_a, x2=d1, relu=True, slope=sl)
            d2 = conv3x3(a2, pk(k1), cs, bias=b_c, relu=True, drop_p=p_drop, seed=next_seed(), slope=sl)
            dsaved.append((u, a1, d1, a2, d2))
            cur = d2
        wf, bf = wb(22)
        wf8 = wf.float().reshape(8).contiguous()
        mask = final_fwd(cur, wf8, bf.float().contiguous())
        if DEBUG is not None:
            DEBUG["fwd"] = {"t": t, "enc": saved, "dec": dsaved}
        ctx.x = x
        ctx.pre = pre
        ctx.t = t
        ctx.saved_enc = saved
        ctx.saved_dec = dsaved
        ctx.P = P
        # an alias, not the output object itself: output -> grad_fn -> ctx -> output would be a reference
        # cycle, and the ~4.5 GB of activations hanging off ctx would live until the cyclic GC runs
        ctx.mask = mask.detach()
        ctx.scale = dropout_scale(p_drop)
        ctx.slope = sl
        ctx.n_params = len(params)
        ctx.norm = bool(norm)
        if ctx.norm:
            # mask / amax(mask) per image (icp_weight_policy.py:192-193), fused around the final layer
            npix = H * W
            part = torch.empty(B * 64, dtype=torch.float32, device=dev)
            mask_n = torch.empty_like(mask)
            amax = torch.empty(B, dtype=torch.float32, device=dev)
            _lib.check(_lib.lib().mmk_mask_normalize(_p(mask), B, npix, _p(part), _p(mask_n), _p(amax), _sp(dev)))
            ctx.mask_n = mask_n.detach()
            ctx.amax = amax
            return mask_n
        return mask

    @staticmethod
    def backward(ctx, gmask):
        L = _lib.lib()
        x, t, P = ctx.x, ctx.t, ctx.P
        dev = x.device
        s = ctx.scale
        sl = ctx.slope
        B = x.shape[0]
        gmask = gmask.contiguous().float()

        def W(k):
            return P[2 * k]

        packs_t = dict(zip(range(1, 22), pack_weights_batch([W(k) for k in range(1, 22)], transposed=True)))

        def pkt(k):
            return packs_t[k]

        # one zero-filled fp32 buffer for every gradient accumulator (one fill instead of ~45)
        cin0 = x.shape[1]
        sizes = [8 * cin0 * 9, 8]
        for k in range(1, 22):
            sizes += [9 * W(k).shape[0] * W(k).shape[1], W(k).shape[0]]
        sizes += [8, 1]
        offs = [0]
        for n in sizes:
            offs.append(offs[-1] + (n + 3) // 4 * 4)
        flat = torch.zeros(offs[-1], dtype=torch.float32, device=dev)

        def seg(i):
            return flat[offs[i]:offs[i] + sizes[i]]

        dB = {k: seg(2 * k + 1) for k in range(1, 22)}

        part = {}        # layer -> partial-sum slices (weights + bias)
        part_sets = {}   # layer -> sets of slices written so far

        def wgrad(k, x1, g, x2=None):
            cout_k, cin_k = W(k).shape[0], W(k).shape[1]
            ns = wgrad_slices(cout_k, cin_k, x1.shape[3], x1.shape[0], x1.shape[1], x1.shape[2])
            if ns <= 0:
                raise _lib.MmkError("U-Net backward: no weight-gradient kernel for %d -> %d channels" % (cin_k, cout_k))
            # (a decoder convolution is applied twice: each application writes its own set of slices, second application
            # first -- the order of the native driver)
            if k not in part:
                part[k] = partial_buffer(ns * (2 if k >= 12 else 1), cout_k, cin_k, dev)
                part_sets[k] = 0
            u = part_sets[k]
            part_sets[k] = u + 1
            conv3x3_wgrad_partial(x1, g, cout_k, part[k][u * ns:(u + 1) * ns], x2=x2)

        # ---- final layer
        u4, a1_4, d1_4, a2_4, d2_4 = ctx.saved_dec[4]
        wf8 = W(22).float().reshape(8).contiguous()
        g_fw, g_fb = seg(44), seg(45)
        gz = torch.empty_like(d2_4)
        red = torch.empty(_lib.FINAL_BWD_WS_FLOATS, dtype=torch.float32, device=dev)
        if ctx.norm:
            npix = gmask.shape[1] * gmask.shape[2]
            ws = torch.empty(B * 130, dtype=torch.float32, device=dev)
            _lib.check(L.mmk_final_bwd_normalized(_p(d2_4), _p(wf8), _p(ctx.mask), _p(ctx.mask_n), _p(ctx.amax), _p(gmask), B, npix,
                                                  s, sl, _p(ws[:B * 128]), _p(ws[B * 128:]), _p(gz), _p(g_fw), _p(g_fb), _p(red),
                                                  _sp(dev)))
        else:
            _lib.check(L.mmk_final_bwd(_p(d2_4), _p(wf8), _p(ctx.mask), _p(gmask), gmask.numel(), s, sl, _p(gz), _p(g_fw),
                                       _p(g_fb), _p(red), _sp(dev)))
        dbg = DEBUG
        if dbg is not None:
            dbg["gz_d2_4"] = gz
        # ---- decoder, j = 4..0
        g_skip = [None] * 5            # gradient w.r.t. t[i], i = 0..4, written by the decoder
        g_t5 = None
        for j in range(4, -1, -1):
            u, a1, d1, a2, d2 = ctx.saved_dec[j]
            skip = t[4 - j]
            cs = skip.shape[3]
            cin_first = u.shape[3]
            k0, k1 = 12 + 2 * j, 13 + 2 * j
            # second application
            wgrad(k1, a2, gz)
            gz_a2 = conv3x3(gz, pkt(k1), cs, relu_src=a2, scale=1.0, slope=sl)
            wgrad(k0, skip, gz_a2, x2=d1)
            # skip of dec4 is the post-dropout activation of encoder block 0: apply its factor here
            skip_is_act = (j == 4)
            gsk, gz_d1 = conv3x3(gz_a2, pkt(k0), 2 * cs, split=cs, relu_src=skip if skip_is_act else None,
                                 scale=s, relu_src2=d1, scale2=s, slope=sl)
            g_skip[4 - j] = gsk
            # first application
            wgrad(k1, a1, gz_d1)
            gz_a1 = conv3x3(gz_d1, pkt(k1), cs, relu_src=a1, scale=1.0, slope=sl)
            wgrad(k0, u, gz_a1)
            g_u = conv3x3(gz_a1, pkt(k0), cin_first, slope=sl)
            if dbg is not None:
                dbg["gz_a2_%d" % j], dbg["gz_d1_%d" % j], dbg["gz_a1_%d" % j] = gz_a2, gz_d1, gz_a1
                dbg["g_u_%d" % j], dbg["g_skip_%d" % j] = g_u, gsk.clone()
            if j > 0:
                prev_d2 = ctx.saved_dec[j - 1][4]
                gz = upsample_bwd(g_u, prev_d2.shape[1], prev_d2.shape[2], relu_src=prev_d2, scale=s, slope=sl)
            else:
                g_t5 = upsample_bwd(g_u, t[5].shape[1], t[5].shape[2])
        # ---- encoder, i = 5..1
        g_t = g_t5
        for i in range(5, 0, -1):
            a, d = ctx.saved_enc["e%d" % i]
            gz_d = maxpool2_bwd(d, g_t, s, sl)
            if dbg is not None:
                dbg["g_t_%d" % i], dbg["gz_d_e%d" % i] = g_t.clone(), gz_d
            wgrad(2 * i + 1, a, gz_d)
            gz_a = conv3x3(gz_d, pkt(2 * i + 1), a.shape[3], relu_src=a, scale=1.0, slope=sl)
            wgrad(2 * i, t[i - 1], gz_a)
            tgt = g_skip[i - 1]
            cin_i = t[i - 1].shape[3]
            if i == 1:      # t[0] is an activation: factor on the dgrad, then accumulate
                conv3x3(gz_a, pkt(2 * i), cin_i, out=tgt, accumulate=True, relu_src=t[0], scale=s, slope=sl)
            else:
                conv3x3(gz_a, pkt(2 * i), cin_i, out=tgt, accumulate=True, slope=sl)
            g_t = tgt
        # ---- encoder block 0
        a0, d0 = ctx.saved_enc["e0"]
        gz_d0 = g_t
        wgrad(1, a0, gz_d0)
        gz_a0 = conv3x3(gz_d0, pkt(1), 8, relu_src=a0, scale=1.0, slope=sl)
        g_w0, g_b0 = seg(0).view(8, cin0, 3, 3), seg(1)
        conv_first_wgrad(x, gz_a0, ctx.pre, g_w0, g_b0)
        # ---- assemble parameter gradients in input order
        out = [g_w0, g_b0]
        assert all(part_sets[k] == (2 if k >= 12 else 1) for k in range(1, 22))
        items = [(part[k], W(k).shape[0], W(k).shape[1], dB[k]) for k in range(1, 22)]
        for k, gw in zip(range(1, 22), wgrad_unpack_batch(items)):
            out += [gw, dB[k]]
        out += [g_fw.reshape(1, 8, 1, 1), g_fb]
        out = [g.to(p.dtype) for g, p in zip(out, P)]
        return (None, None, None, None, None, None, None) + tuple(out)


# ----------------------------------------------------------------------------- the network as two C-ABI calls
GRAD_BUCKET_EVENTS = None    # ddp.FlatGradSync(overlap=True) arms this with MMK_UNET_GRAD_BUCKETS recorded torch.cuda.Event objects: the
#                              native backward then records event b behind the last kernel that writes a gradient of bucket b
GRAD_BUCKET_PASSES = [0]     # backward passes that recorded the armed events (a pass that did not -- another driver, the BatchNorm
#                              network -- must not be mistaken for one: stale events have long fired)
GRAD_BUCKETS = ((24, 22), (12, 12), (0, 12))    # (first parameter, count) per bucket, in completion order (mmk_unet_grad_bucket)
DRIVER = os.environ.get("MMK_UNET_DRIVER", "native")     # "native": mmk_unet_forward / _backward (csrc/mmk_unet_driver.hip);
#                                                          "python": the launch-by-launch schedule of _UNet above (the same
#                                                          kernels in the same order: bit-identical results; tests, diagnostics)


def workspace_tensor(ws, B, H, W, cin, tid):
    """View of one activation inside a native-driver workspace (include/mmk.h: mmk_unet_tensor ids)."""
    off, h, w, c = ctypes.c_size_t(0), ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_int32(0)
    _lib.check(_lib.lib().mmk_unet_tensor(B, H, W, cin, tid, ctypes.byref(off), ctypes.byref(h), ctypes.byref(w), ctypes.byref(c)))
    n = B * h.value * w.value * c.value
    return ws[off.value:off.value + 2 * n].view(BF16).view(B, h.value, w.value, c.value)


class _UNetNative(torch.autograd.Function):
    """The same network through mmk_unet_forward / mmk_unet_backward: one C call per pass."""

    @staticmethod
    def forward(ctx, x, pre, drop_p, seed, training, norm, slope, *params):
        L = _lib.lib()
        dev = x.device
        x = x.contiguous().float()
        B, cin, H, W = x.shape
        P = [p.detach().float().contiguous() for p in params]
        nbytes = L.mmk_unet_workspace_bytes(B, H, W, cin)
        if nbytes == 0:
            raise _lib.MmkError("libmmk_hip: %s" % L.mmk_last_error().decode(errors="replace"))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        mask = torch.empty(B, H, W, dtype=torch.float32, device=dev)
        pp = (ctypes.c_void_p * len(P))(*[t.data_ptr() for t in P])
        d = _lib.UNetDesc(B=B, H=H, W=W, cin=cin, x=x.data_ptr(), pre=None if pre is None else pre.data_ptr(), params=pp,
                          drop_p=float(drop_p) if training else 0.0, seed=int(seed) & 0xFFFFFFFF, leaky_slope=float(slope),
                          norm=1 if norm else 0, workspace=ws.data_ptr(), workspace_bytes=nbytes, mask=mask.data_ptr(),
                          keep_full_res=1 if DEBUG is not None else 0)     # (the pre-pool outputs: only diagnostics read them)
        _lib.check(L.mmk_unet_forward(ctypes.byref(d), _sp(dev)))
        if DEBUG is not None:
            def tv(tid):
                return workspace_tensor(ws, B, H, W, cin, tid)
            DEBUG["fwd"] = {"t": [tv(12 + i) for i in range(6)],
                            "enc": {"e%d" % i: (tv(i), tv(6 + i)) for i in range(6)},
                            "dec": [tuple(tv(18 + 5 * j + q) for q in range(5)) for j in range(5)]}
        ctx.desc_args = (B, H, W, cin, float(drop_p) if training else 0.0, int(seed) & 0xFFFFFFFF, float(slope), 1 if norm else 0)
        ctx.x, ctx.pre, ctx.P, ctx.ws = x, pre, P, ws
        # an alias, not the output object itself (output -> grad_fn -> ctx -> output would be a reference cycle
        # that keeps the workspace alive until the cyclic GC runs)
        ctx.mask = mask.detach()
        return mask

    @staticmethod
    def backward(ctx, gmask):
        L = _lib.lib()
        B, H, W, cin, p_drop, seed, slope, norm = ctx.desc_args
        x, P, ws = ctx.x, ctx.P, ctx.ws
        dev = x.device
        gmask = gmask.contiguous().float()
        nscratch = L.mmk_unet_scratch_bytes(B, H, W, cin)
        if nscratch == 0:
            raise _lib.MmkError("libmmk_hip: %s" % L.mmk_last_error().decode(errors="replace"))
        scratch = torch.empty(nscratch, dtype=torch.uint8, device=dev)
        sizes = [p.numel() for p in P]
        offs = [0]
        for n in sizes:
            offs.append(offs[-1] + (n + 3) // 4 * 4)
        flat = torch.zeros(offs[-1], dtype=torch.float32, device=dev)   # (zeros: the alignment gaps are summed by a DDP all-reduce)
        grads = [flat[offs[i]:offs[i] + sizes[i]].view(P[i].shape) for i in range(len(P))]
        pp = (ctypes.c_void_p * len(P))(*[t.data_ptr() for t in P])
        gp = (ctypes.c_void_p * len(P))(*[t.data_ptr() for t in grads])
        d = _lib.UNetDesc(B=B, H=H, W=W, cin=cin, x=x.data_ptr(), pre=None if ctx.pre is None else ctx.pre.data_ptr(), params=pp,
                          drop_p=p_drop, seed=seed, leaky_slope=slope, norm=norm, workspace=ws.data_ptr(),
                          workspace_bytes=ws.numel(), mask=ctx.mask.data_ptr())
        evs = GRAD_BUCKET_EVENTS
        if evs is not None:
            # data-parallel step: event b fires when the gradients of bucket b are final (include/mmk.h: mmk_unet_backward_buckets)
            ep = (ctypes.c_void_p * len(evs))(*[ctypes.c_void_p(int(e.cuda_event)) for e in evs])
            _lib.check(L.mmk_unet_backward_buckets(ctypes.byref(d), _p(gmask), gp, _p(scratch), nscratch, ep, _sp(dev)))
            GRAD_BUCKET_PASSES[0] += 1
        else:
            _lib.check(L.mmk_unet_backward(ctypes.byref(d), _p(gmask), gp, _p(scratch), nscratch, _sp(dev)))
        ctx.ws = None
        return (None, None, None, None, None, None, None) + tuple(grads)


def unet_mask(module, x, training, seed, norm=False, pre=None, slope=0.0, driver=None):
    """sigmoid mask (B,H,W) fp32 of the module's network on fp32 NCHW input x; ``norm``: divided by its
    per-image maximum (the policy's ``norm_weights``), inside the same autograd node; ``pre`` (C,2): the
    input is (x - pre[c,0]) * pre[c,1], applied by the first layer while it loads x; ``slope`` > 0: the
    LeakyReLU network.  ``driver``: "native" (default, one C call per pass) or "python" (launch by launch)."""
    fn = _UNetNative if (driver or DRIVER) == "native" else _UNet
    return fn.apply(x, pre, float(module.dropout), int(seed), bool(training), bool(norm), float(slope), *param_list(module))
