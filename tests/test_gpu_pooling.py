"""Pooling by arg-max codes (the pre-pool tensor is neither written nor re-read): codes and routed gradients against
a pooling pass over the stored full-resolution tensor, and the whole U-Net without its pre-pool outputs, bit for bit."""
import numpy as np
import pytest
import torch

from mm_masking_amd import train_icp_weights as trn

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


@pytest.mark.parametrize("B,H,W,C,drop", [(2, 64, 96, 16, 0.1), (1, 40, 64, 32, 0.0), (3, 50, 84, 16, 0.05), (2, 25, 42, 32, 0.2),
                                          (32, 64, 64, 16, 0.05)])
def test_pool_arg_codes_route_the_gradient_like_the_full_resolution_tensor(B, H, W, C, drop):
    """The encoder's second convolutions below 64 channels write the 2x2 max-pool and its arg-max codes and NOT their
    full-resolution output (mmk_conv_desc.pool_arg); the backward pass routes the pooled gradient by the codes.  Against the
    path that stores the tensor: same pooled values, same codes as a pooling pass over the stored tensor makes, and a
    bit-identical routed gradient -- odd sizes (floor pooling), ties (zero windows: dropout, ReLU) and all."""
    from mm_masking_amd import unet_hip as uh
    g = torch.Generator().manual_seed(B + H + C)
    x = (torch.randn(B, H, W, C, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
    w = (torch.randn(C, C, 3, 3, generator=g) / (3 * C ** 0.5)).to(DEV)
    bias = (torch.randn(C, generator=g) * 0.1).to(DEV)
    wp = uh.pack_weights(w)
    assert uh.pool_fusable(C, C, B, H, W)
    Hp, Wp = H // 2, W // 2
    pooled_a = torch.empty(B, Hp, Wp, C, dtype=torch.bfloat16, device=DEV)
    d = uh.conv3x3(x, wp, C, bias=bias, relu=True, drop_p=drop, seed=7, pool_out=pooled_a)
    pooled_b = torch.full((B, Hp, Wp, C), -1.0, dtype=torch.bfloat16, device=DEV)
    codes_b = torch.full((B, Hp, Wp, C // 2), 0xEE, dtype=torch.uint8, device=DEV)
    assert uh.conv3x3(x, wp, C, bias=bias, relu=True, drop_p=drop, seed=7, pool_out=pooled_b, pool_arg=codes_b) is None
    pooled_c, codes_c = uh.maxpool2_arg(d)
    torch.cuda.synchronize()
    assert torch.equal(pooled_a, pooled_b) and torch.equal(pooled_a, pooled_c)
    assert torch.equal(codes_b, codes_c)
    # the codes against a host derivation from the stored tensor
    df = d.float().cpu()[:, :2 * Hp, :2 * Wp]
    win = torch.stack([df[:, 0::2, 0::2], df[:, 0::2, 1::2], df[:, 1::2, 0::2], df[:, 1::2, 1::2]], dim=-1)     # (B,Hp,Wp,C,4)
    mx, arg = win.max(dim=-1)
    first = (win == mx.unsqueeze(-1)).float().argmax(dim=-1)          # first maximal position
    want = (first + 4 * (mx > 0)).to(torch.uint8)
    got = torch.stack([codes_b.cpu() & 0xF, codes_b.cpu() >> 4], dim=-1).reshape(B, Hp, Wp, C)
    assert torch.equal(got, want)
    assert float((mx == 0).float().mean()) > 0.01                      # zero windows (ties) do occur
    gy = (torch.randn(B, Hp, Wp, C, generator=g)).to(torch.bfloat16).to(DEV)
    scale = uh.dropout_scale(drop)
    ref = uh.maxpool2_bwd(d, gy, scale)
    out = uh.maxpool2_bwd_arg(codes_b, gy, H, W, scale)
    torch.cuda.synchronize()
    assert torch.equal(ref.view(torch.int16), out.view(torch.int16))


@pytest.mark.parametrize("B,H,W,drop", [(2, 64, 64, 0.1), (3, 50, 84, 0.0), (1, 160, 320, 0.05)])
def test_unet_without_pre_pool_outputs_is_bit_identical(B, H, W, drop):
    """mmk_unet_forward / _backward with keep_full_res = 0 (the default: blocks 1 and 2 do not store their pre-pool output,
    every pooling's backward goes by the codes) against keep_full_res = 1 (set while unet_hip.DEBUG captures activations)
    and against the launch-by-launch schedule, which routes by the stored tensors: mask and all 46 gradients bit for bit."""
    from mm_masking_amd import unet_hip as uh
    from mm_masking_amd.icp_weight_policy import LearnICPWeightPolicy
    p = trn.default_params(DEV)
    p.update({"dropout": drop})
    torch.manual_seed(3)
    model = LearnICPWeightPolicy(p).to(DEV)
    model.train()
    g = torch.Generator().manual_seed(11)
    x = torch.rand(B, 1, H, W, generator=g).to(DEV)
    gsel = torch.randn(B, H, W, generator=g).to(DEV)
    pre = uh.channel_minmax(x)
    res = []
    for drv, dbg in (("native", False), ("native", True), ("python", False)):
        model.zero_grad(set_to_none=True)
        uh.DEBUG = {} if dbg else None
        try:
            m = uh.unet_mask(model, x, training=True, seed=4, norm=True, pre=pre, driver=drv)
        finally:
            uh.DEBUG = None
        (m * gsel).sum().backward()
        res.append((m.detach().clone(), [q.grad.detach().clone() for q in uh.param_list(model)]))
    for m, grads in res[1:]:
        assert torch.equal(m, res[0][0])
        for a, b in zip(grads, res[0][1]):
            assert torch.equal(a, b)
