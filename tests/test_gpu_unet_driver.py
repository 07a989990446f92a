"""mmk_unet_forward / mmk_unet_backward (one C call per pass, csrc/mmk_unet_driver.hip) against the
launch-by-launch schedule of the same building blocks (unet_hip._UNet): same kernels, same order, so the mask
and every parameter gradient must be bit-identical -- all 46 of them since round 3 (the first / final layer gradients are
block partial sums reduced in a fixed order: no float atomics left)."""
import numpy as np
import pytest
import torch

from mm_masking_amd import train_icp_weights as trn
from mm_masking_amd import unet_hip as uh
from mm_masking_amd.icp_weight_policy import LearnICPWeightPolicy

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _model(cin, dropout, leaky):
    p = trn.default_params(DEV)
    p.update({"dropout": dropout, "leaky": leaky, "cfar_input": cin >= 2, "range_input": cin >= 3})
    torch.manual_seed(17)
    m = LearnICPWeightPolicy(p).to(DEV)
    m.train()
    return m


@pytest.mark.parametrize("B,cin,H,W,drop,leaky,norm", [(2, 1, 64, 64, 0.1, False, True), (3, 1, 50, 84, 0.0, False, False),
                                                       (2, 3, 96, 96, 0.05, True, True), (1, 1, 160, 320, 0.05, False, True)])
def test_native_driver_is_bit_identical_to_the_per_call_schedule(B, cin, H, W, drop, leaky, norm):
    model = _model(cin, drop, leaky)
    g = torch.Generator().manual_seed(5)
    x = torch.rand(B, cin, H, W, generator=g).to(DEV)
    gsel = torch.randn(B, H, W, generator=g).to(DEV)
    pre = uh.channel_minmax(x) if not leaky else None
    slope = 0.1 if leaky else 0.0
    res = {}
    for drv in ("python", "native"):
        model.zero_grad(set_to_none=True)
        uh.DEBUG = {}
        try:
            m = uh.unet_mask(model, x, training=True, seed=9, norm=norm, pre=pre, slope=slope, driver=drv)
            fwd = uh.DEBUG["fwd"]
        finally:
            uh.DEBUG = None
        (m * gsel).sum().backward()
        res[drv] = (m.detach().clone(), [p.grad.detach().clone() for p in uh.param_list(model)],
                    [t.clone() for t in fwd["t"]], [tuple(v.clone() for v in fwd["dec"][j]) for j in range(5)])
    (m_p, g_p, t_p, d_p), (m_n, g_n, t_n, d_n) = res["python"], res["native"]
    assert torch.equal(m_p, m_n)
    for a, b in zip(t_p, t_n):
        assert a.shape == b.shape and torch.equal(a, b)
    for j in range(5):
        for a, b in zip(d_p[j], d_n[j]):
            assert torch.equal(a, b)
    names = [n for n, _ in model.named_parameters()]
    for i, (n, a, b) in enumerate(zip(names, g_p, g_n)):
        assert a.shape == b.shape, n
        assert torch.equal(a, b), n


def test_native_driver_argument_checks():
    import ctypes
    from mm_masking_amd import _lib
    L = _lib.lib()
    assert L.mmk_unet_workspace_bytes(2, 16, 64, 1) == 0 and b"unsupported" in L.mmk_last_error()
    assert L.mmk_unet_workspace_bytes(2, 64, 64, 5) == 0
    n = L.mmk_unet_workspace_bytes(32, 640, 640, 1)
    assert 4 * 2 ** 30 < n < 8 * 2 ** 30                      # ~4.5 GB of bf16 activations at the BASELINE batch
    d = _lib.UNetDesc(B=2, H=64, W=64, cin=1)
    assert L.mmk_unet_forward(ctypes.byref(d), None) == -1 and b"NULL" in L.mmk_last_error()
