"""Round-4 GPU tests: the loss terms as single launches (csrc/mmk_loss.hip) against PyTorch's own operators."""
import numpy as np
import pytest
import torch

from mm_masking_amd import train_icp_weights as trn

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


@pytest.mark.parametrize("shape", [(2, 64, 64), (3, 33, 7), (1, 640, 640)])
def test_fused_bce_mean_matches_torch(shape):
    """_BceMeanFn == torch.nn.BCELoss() (train_icp_weights.py:180,223-226): value, gradient through an upstream factor, the
    -100 clamp of the logs at x = 0 / 1, sizes that are not multiples of four."""
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.rand(*shape, generator=g)
    x.view(-1)[:4] = torch.tensor([0.0, 1.0, 1e-30, 1.0 - 1e-7])
    t = (torch.rand(*shape, generator=g) > 0.7).float()
    t.view(-1)[:4] = torch.tensor([1.0, 0.0, 1.0, 0.0])
    xa = x.clone().to(DEV).requires_grad_(True)
    xb = x.clone().to(DEV).requires_grad_(True)
    la = trn._bce_mean(xa, t.to(DEV))
    lb = torch.nn.BCELoss()(xb, t.to(DEV))
    assert la.shape == lb.shape == ()
    np.testing.assert_allclose(la.item(), lb.item(), rtol=2e-6)
    (0.37 * la).backward()
    (0.37 * lb).backward()
    np.testing.assert_allclose(xa.grad.cpu().numpy(), xb.grad.cpu().numpy(), rtol=2e-6, atol=1e-12)
    # deterministic: the same bits on a second evaluation
    assert torch.equal(trn._bce_mean(xa.detach(), t.to(DEV)), la.detach())
    with pytest.raises(ValueError):
        trn._bce_mean(xa, t.to(DEV)[..., :-1])


@pytest.mark.parametrize("B", [1, 5, 32, 70])
def test_fused_pose_loss_matches_torch(B):
    """_PoseLossFn == the reference's expression for gt_eye (train_icp_weights.py:193,197-200): torch.norm over the 1-vector
    xi_theta and the 2-vector xi_r, batch means, and the gradient w.r.t. T_pred for arbitrary upstream weights; a pair with a
    zero translation / rotation residual gets a zero gradient there (torch.norm's backward at zero)."""
    g = torch.Generator().manual_seed(B)
    T = torch.eye(4).repeat(B, 1, 1) + 0.3 * torch.randn(B, 4, 4, generator=g)
    T[0, 0, 3] = T[0, 1, 3] = 0.0
    T[0, 1, 0] = 0.0
    Ta = T.clone().to(DEV).requires_grad_(True)
    Tb = T.clone().to(DEV).requires_grad_(True)
    rot_a, trans_a = trn._PoseLossFn.apply(Ta)
    xi = Tb - torch.eye(4, device=DEV)
    rot_b = torch.norm(xi[:, 1, 0].unsqueeze(-1), dim=1).mean()
    trans_b = torch.norm(xi[:, 0:2, 3], dim=1).mean()
    np.testing.assert_allclose([rot_a.item(), trans_a.item()], [rot_b.item(), trans_b.item()], rtol=1e-6)
    (1.5 * rot_a + 0.25 * trans_a).backward()
    (1.5 * rot_b + 0.25 * trans_b).backward()
    np.testing.assert_allclose(Ta.grad.cpu().numpy(), Tb.grad.cpu().numpy(), rtol=1e-6, atol=1e-9)
    assert float(Ta.grad[0].abs().sum()) == 0.0
