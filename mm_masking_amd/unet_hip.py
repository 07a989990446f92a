"""Hand-written gfx950 path of the mask U-Net (csrc/mmk_unet.hip through the C ABI).

Low-level wrappers (NHWC bf16 tensors) + the fused forward/backward of the reference's
default network configuration (ReLU, no batch norm; mm_masking/icp_weight_policy.py:84-99,
161-184).  The parameters stay in the module's nn.Conv2d objects (state_dict
compatibility); every step packs them to bf16 MFMA-fragment order on the device.
"""
import ctypes

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


def conv3x3(x1, wpack, cout, bias=None, x2=None, relu=False, drop_p=0.0, seed=0, out=None, out2=None, split=None,
            relu_src=None, scale=1.0, relu_src2=None, scale2=1.0, accumulate=False, accumulate2=False):
    """3x3 / pad 1 convolution on NHWC bf16.  input = concat(x1, x2); output channels
    [0,split) -> out, [split,cout) -> out2 (split=None: single output)."""
    B, H, W, C1 = x1.shape
    C2 = 0 if x2 is None else x2.shape[3]
    O1 = cout if split is None else split
    O2 = cout - O1
    if out is None:
        out = torch.empty(B, H, W, O1, dtype=BF16, device=x1.device)
    if O2 > 0 and out2 is None:
        out2 = torch.empty(B, H, W, O2, dtype=BF16, device=x1.device)
    d = _lib.ConvDesc(x1=_p(x1), x2=_p(x2), C1=C1, C2=C2, wpack=_p(wpack), bias=_p(bias), y1=_p(out),
                      relu_src1=_p(relu_src), O1=O1, accumulate1=1 if accumulate else 0, scale1=float(scale),
                      y2=_p(out2), relu_src2=_p(relu_src2), O2=O2, accumulate2=1 if accumulate2 else 0,
                      scale2=float(scale2), B=B, H=H, W=W, relu=1 if relu else 0, drop_p=float(drop_p),
                      seed=int(seed) & 0xFFFFFFFF)
    _lib.check(_lib.lib().mmk_conv3x3(ctypes.byref(d), _lib.stream_ptr(x1.device)))
    return (out, out2) if O2 > 0 else out


def conv3x3_wgrad(x1, g, cout, x2=None, dWt=None, db=None):
    """Accumulate the weight / bias gradient into dWt (9,cout,cin) fp32 and db (cout,) fp32."""
    B, H, W, C1 = x1.shape
    C2 = 0 if x2 is None else x2.shape[3]
    cin = C1 + C2
    if dWt is None:
        dWt = torch.zeros(9, cout, cin, dtype=torch.float32, device=x1.device)
    _lib.check(_lib.lib().mmk_conv3x3_wgrad(_p(x1), _p(x2), C1, C2, _p(g), cout, B, H, W, _p(dWt), _p(db),
                                            _lib.stream_ptr(x1.device)))
    return dWt


def wgrad_unpack(dWt, accumulate_into=None):
    """(9,cout,cin) -> (cout,cin,3,3); accumulates into an existing gradient when given."""
    _, cout, cin = dWt.shape
    out = accumulate_into if accumulate_into is not None else torch.empty(cout, cin, 3, 3, dtype=torch.float32,
                                                                          device=dWt.device)
    _lib.check(_lib.lib().mmk_conv3x3_wgrad_unpack(_p(dWt), cout, cin, 1 if accumulate_into is not None else 0, _p(out),
                                                   _lib.stream_ptr(dWt.device)))
    return out
