"""The hand-written U-Net kernels (NHWC bf16, MFMA) against plain PyTorch fp32 references
of the same operators evaluated on the same bf16-rounded operands."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from mm_masking_amd import unet_hip as uh

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _rand_nhwc(B, H, W, C, seed):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(B, H, W, C, generator=g)).to(DEV).to(torch.bfloat16)


def _ref_conv(x_nhwc, w, b, relu):
    y = F.conv2d(x_nhwc.float().permute(0, 3, 1, 2), w.to(torch.bfloat16).float(), b, padding=1)
    if relu:
        y = F.relu(y)
    return y.permute(0, 2, 3, 1)


@pytest.mark.parametrize("cin,cout,H,W", [(8, 8, 16, 32), (8, 16, 24, 40), (16, 16, 40, 64), (16, 8, 17, 33),
                                           (16, 32, 20, 20), (32, 32, 16, 64), (32, 16, 9, 31), (32, 64, 16, 32),
                                           (64, 64, 20, 20), (64, 32, 8, 32), (64, 128, 10, 10), (128, 128, 12, 20),
                                           (128, 64, 8, 8), (128, 256, 6, 10), (256, 256, 5, 7), (256, 128, 20, 20)])
def test_conv3x3_forward(cin, cout, H, W):
    B = 2
    x = _rand_nhwc(B, H, W, cin, cin * 1000 + cout)
    g = torch.Generator().manual_seed(7)
    w = (torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(9 * cin)).to(DEV)
    b = torch.randn(cout, generator=g).to(DEV)
    wp = uh.pack_weights(w)
    y = uh.conv3x3(x, wp, cout, bias=b, relu=True)
    ref = _ref_conv(x, w, b, True)
    err = (y.float() - ref).abs().max().item()
    assert err < 0.03, err                      # bf16 output rounding of O(1) values
    y2 = uh.conv3x3(x, wp, cout, bias=None, relu=False)
    ref2 = _ref_conv(x, w, None, False)
    assert (y2.float() - ref2).abs().max().item() < 0.03


def test_conv3x3_concat_split_accumulate_relu_src():
    B, H, W = 2, 24, 40
    xa, xb = _rand_nhwc(B, H, W, 8, 1), _rand_nhwc(B, H, W, 8, 2)
    g = torch.Generator().manual_seed(3)
    w = (torch.randn(16, 16, 3, 3, generator=g) / 12).to(DEV)
    wp = uh.pack_weights(w)
    # concat input == conv on the concatenated tensor
    y = uh.conv3x3(xa, wp, 16, x2=xb)
    ref = _ref_conv(torch.cat([xa, xb], dim=3), w, None, False)
    assert (y.float() - ref).abs().max().item() < 0.03
    # split outputs, ReLU-backward factor on the second part, accumulation into the first
    base = _rand_nhwc(B, H, W, 8, 4)
    src = _rand_nhwc(B, H, W, 8, 5)
    o1 = base.clone()
    o1, o2 = uh.conv3x3(xa, wp, 16, x2=xb, split=8, out=o1, accumulate=True, relu_src2=src, scale2=1.25)
    assert (o1.float() - (base.float() + ref[..., :8])).abs().max().item() < 0.05
    want2 = torch.where(src.float() > 0, ref[..., 8:] * 1.25, torch.zeros_like(ref[..., 8:]))
    assert (o2.float() - want2).abs().max().item() < 0.04


def test_conv3x3_transposed_is_data_gradient():
    B, H, W, cin, cout = 2, 20, 36, 16, 32
    x = _rand_nhwc(B, H, W, cin, 11).float().permute(0, 3, 1, 2).requires_grad_(True)
    g = torch.Generator().manual_seed(5)
    w = (torch.randn(cout, cin, 3, 3, generator=g) / 12).to(DEV)
    gy = _rand_nhwc(B, H, W, cout, 12)
    y = F.conv2d(x, w.to(torch.bfloat16).float(), padding=1)
    y.backward(gy.float().permute(0, 3, 1, 2))
    wpt = uh.pack_weights(w, transposed=True)
    gx = uh.conv3x3(gy, wpt, cin)
    assert (gx.float() - x.grad.permute(0, 2, 3, 1)).abs().max().item() < 0.05


def test_conv3x3_dropout_statistics():
    B, H, W = 2, 64, 64
    x = torch.ones(B, H, W, 8, device=DEV, dtype=torch.bfloat16)
    w = torch.zeros(8, 8, 3, 3, device=DEV)
    w[:, :, 1, 1] = torch.eye(8, device=DEV)
    wp = uh.pack_weights(w)
    y = uh.conv3x3(x, wp, 8, relu=True, drop_p=0.25, seed=123).float()
    kept = (y > 0).float().mean().item()
    assert abs(kept - 0.75) < 0.02
    assert torch.allclose(y[y > 0], torch.tensor(1.0 / 0.75, device=DEV), atol=0.01)
    y2 = uh.conv3x3(x, wp, 8, relu=True, drop_p=0.25, seed=124).float()
    assert (y2 != y).any() and torch.equal(uh.conv3x3(x, wp, 8, relu=True, drop_p=0.25, seed=123).float(), y)


@pytest.mark.parametrize("cin,cout,H,W", [(8, 8, 16, 32), (8, 16, 24, 40), (16, 16, 40, 64), (16, 8, 17, 33),
                                           (16, 32, 20, 20), (32, 32, 16, 64), (32, 16, 9, 31), (32, 64, 16, 32),
                                           (64, 64, 20, 20), (64, 32, 8, 32), (64, 128, 10, 10), (128, 128, 12, 20),
                                           (128, 64, 8, 8), (128, 256, 6, 10), (256, 256, 5, 7), (256, 128, 20, 20)])
def test_conv3x3_wgrad(cin, cout, H, W):
    B = 3
    x = _rand_nhwc(B, H, W, cin, 100 + cin)
    gy = _rand_nhwc(B, H, W, cout, 200 + cout)
    w = torch.zeros(cout, cin, 3, 3, device=DEV, requires_grad=True)
    bias = torch.zeros(cout, device=DEV, requires_grad=True)
    y = F.conv2d(x.float().permute(0, 3, 1, 2), w, bias, padding=1)
    y.backward(gy.float().permute(0, 3, 1, 2))
    db = torch.zeros(cout, device=DEV)
    dWt = uh.conv3x3_wgrad(x, gy, cout, db=db)
    dW = uh.wgrad_unpack(dWt)
    scale = w.grad.abs().max().item()
    assert (dW - w.grad).abs().max().item() < 2e-3 * scale + 1e-3, ((dW - w.grad).abs().max().item(), scale)
    assert (db - bias.grad).abs().max().item() < 2e-3 * bias.grad.abs().max().item() + 1e-3
    # accumulation of a second application (shared decoder weights) and into an existing .grad
    dWt2 = uh.conv3x3_wgrad(x, gy, cout, dWt=dWt.clone())
    acc = uh.wgrad_unpack(dWt2, accumulate_into=torch.ones_like(dW))
    assert (acc - (2 * w.grad + 1)).abs().max().item() < 4e-3 * scale + 2e-3


def test_conv3x3_wgrad_concat():
    B, H, W = 2, 24, 40
    xa, xb = _rand_nhwc(B, H, W, 8, 1), _rand_nhwc(B, H, W, 8, 2)
    gy = _rand_nhwc(B, H, W, 8, 3)
    w = torch.zeros(8, 16, 3, 3, device=DEV, requires_grad=True)
    y = F.conv2d(torch.cat([xa, xb], 3).float().permute(0, 3, 1, 2), w, padding=1)
    y.backward(gy.float().permute(0, 3, 1, 2))
    dW = uh.wgrad_unpack(uh.conv3x3_wgrad(xa, gy, 8, x2=xb))
    assert (dW - w.grad).abs().max().item() < 2e-3 * w.grad.abs().max().item() + 1e-3
