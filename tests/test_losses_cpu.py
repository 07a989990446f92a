"""CPU-only checks of the trainer's loss functions (product and oracle twin) against the reference's golden values
(tests/golden/losses.npz, made by importing /root/reference/mm_masking/train_icp_weights.py: tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest
import torch

from mm_masking_amd import train_icp_weights as trn
from oracle import train_ref


class _M:
    mean_all_pts = torch.tensor(40.0)


@pytest.mark.parametrize("impl", ["product", "oracle"])
def test_validation_loss_golden(golden_dir, impl):
    g = np.load(os.path.join(golden_dir, "losses.npz"), allow_pickle=False)
    Tp, Tg = torch.from_numpy(g["T_pred"]), torch.from_numpy(g["T_gt"])
    f = trn.eval_validation_loss if impl == "product" else train_ref.eval_validation_loss
    np.testing.assert_allclose(f(Tp, Tg, gt_eye=True).numpy(), g["val_eye"], rtol=1e-6)
    np.testing.assert_allclose(f(Tp, Tg, gt_eye=False).numpy(), g["val_gt"], rtol=1e-6)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_product_training_loss_rot_trans_golden(golden_dir, tag):
    """eval_training_loss of the product (not its oracle twin) on the golden poses: gt_eye True (a) and
    False (b), default and non-default loss weights.  The mask terms need the BEV raster (a HIP kernel) and
    are covered by the GPU twin of this test; here they are switched off and the pose terms compared."""
    g = np.load(os.path.join(golden_dir, "losses.npz"), allow_pickle=False)
    lw = dict(zip([str(k) for k in g["lw_keys"]], g["lw_" + tag].tolist()))
    lw_pose = dict(lw, fft=0.0, mask_pts=0.0, cfar=0.0, num_pts=0.0)
    Tp = torch.from_numpy(g["T_pred"]).requires_grad_(True)
    mask = torch.full((5, 8, 8), 0.5)
    loss, comp = trn.eval_training_loss(Tp, mask, torch.tensor(33.0), torch.from_numpy(g["T_gt"]), {}, {}, _M(),
                                        loss_weights=lw_pose, gt_eye=(tag == "a"), epoch=0)
    want = g["comp_" + tag]
    np.testing.assert_allclose([float(comp["rot"]), float(comp["trans"])], want[:2], rtol=1e-6)
    loss.backward()
    # the pose terms are the only ones that reach T_pred
    np.testing.assert_allclose(Tp.grad.numpy(), g["gT_" + tag], rtol=1e-5, atol=1e-7)
