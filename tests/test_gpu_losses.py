"""The trainer's loss functions on the GPU: every term of ``eval_training_loss`` against the reference's golden values, and the
single-launch loss kernels (csrc/mmk_loss.hip) against PyTorch's own operators."""
import os

import numpy as np
import pytest
import torch

from mm_masking_amd import train_icp_weights as trn

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


class _M:
    mean_all_pts = torch.tensor(40.0)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_product_training_loss_golden_full(golden_dir, tag):
    """The product's eval_training_loss with every term (rot, trans, fft / cfar / map-points BCE, num_pts) against the
    reference's golden values: gt_eye True (a) / False (b), default and non-default loss weights."""
    g = np.load(os.path.join(golden_dir, "losses.npz"), allow_pickle=False)
    lw = dict(zip([str(k) for k in g["lw_keys"]], g["lw_" + tag].tolist()))
    Bq = 5
    lmask = np.random.default_rng(int(g["seed_mask"])).uniform(0.01, 0.99, size=(Bq, 640, 640)).astype(np.float32)
    lfft = np.random.default_rng(int(g["seed_fft"])).uniform(0, 1, size=(Bq, 640, 640)).astype(np.float32)
    lcfar = (np.random.default_rng(int(g["seed_cfar"])).uniform(0, 1, size=(Bq, 640, 640)) > 0.95).astype(np.float32)
    Tp = torch.from_numpy(g["T_pred"]).to(DEV).requires_grad_(True)
    mask = torch.from_numpy(lmask).to(DEV).requires_grad_(True)
    m = _M()
    m.mean_all_pts = torch.tensor(40.0, device=DEV)
    loss, comp = trn.eval_training_loss(Tp, mask, torch.tensor(33.0, device=DEV), torch.from_numpy(g["T_gt"]).to(DEV),
                                        {"fft_data": torch.from_numpy(lfft), "fft_cfar": torch.from_numpy(lcfar)},
                                        {"pc": torch.from_numpy(g["pts"])}, m, loss_weights=lw, gt_eye=(tag == "a"), epoch=0)
    loss.backward()
    assert abs(float(loss) - float(g["loss_" + tag])) < 2e-5 * max(1.0, abs(float(g["loss_" + tag])))
    got = np.array([float(comp[k]) for k in ("rot", "trans", "fft", "mask_pts", "cfar", "num_pts")])
    np.testing.assert_allclose(got, g["comp_" + tag], rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(Tp.grad.cpu().numpy(), g["gT_" + tag], rtol=1e-4, atol=1e-7)
    assert abs(float(mask.grad.double().sum()) - float(g["gmask_sum_" + tag])) < 1e-3 * max(1e-3, abs(float(g["gmask_abs_" + tag])))
    assert abs(float(mask.grad.double().abs().sum()) / float(g["gmask_abs_" + tag]) - 1.0) < 1e-4 if float(g["gmask_abs_" + tag]) > 0 else True
    v = trn.eval_validation_loss(Tp.detach(), torch.from_numpy(g["T_gt"]).to(DEV), gt_eye=(tag == "a"))
    np.testing.assert_allclose(v.cpu().numpy(), g["val_eye" if tag == "a" else "val_gt"], rtol=1e-5)


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
