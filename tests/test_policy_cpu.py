"""Host-side logic of the policy mirror on the CPU (no HIP call): the nn.Module path of
``LearnICPWeightPolicy`` (input assembly, normalisation, encoder / twice-applied decoder, amax
normalisation, ``mask_only``) against the golden vectors of the reference module, and the checkpoint
helpers of the trainer.  CPU only."""
import os

import numpy as np
import pytest
import torch

from mm_masking_amd import train_icp_weights as trn
from mm_masking_amd.icp_weight_policy import LearnICPWeightPolicy


def _params(**over):
    p = trn.default_params(torch.device("cpu"))
    p.update({"dropout": 0.0, "amp_dtype": torch.float32, "unet_backend": "torch"})
    p.update(over)
    return p


@pytest.mark.parametrize("tag", ["a", "b"])
def test_module_path_matches_reference_golden(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, "unet.npz"), allow_pickle=False)
    over = {}
    if tag == "b":
        over = {"cfar_input": True, "range_input": True, "leaky": True, "normalize": ["standardize"], "log_transform": True}
    torch.manual_seed(1234)
    model = LearnICPWeightPolicy(_params(**over))
    model.train()
    names = [str(n) for n in g["names_" + tag]]
    assert list(model.state_dict().keys()) == names
    if tag == "b":
        model.range_mask = torch.from_numpy(g["range_b"])
    scan = {"fft_data": torch.from_numpy(g["x_" + tag]), "fft_cfar": torch.from_numpy(g["cfar_" + tag]),
            "raw_pc": torch.zeros(2, 4, 3)}
    m = model(scan, {"pc": torch.zeros(2, 4, 6)}, torch.eye(4).repeat(2, 1, 1), mask_only=True)
    np.testing.assert_allclose(m.detach().numpy(), g["mask_" + tag], atol=2e-6)
    (m * torch.from_numpy(g["gsel_" + tag])).sum().backward()
    grads = dict(model.named_parameters())
    ga = np.array([grads[k].grad.double().abs().sum().item() for k in names])
    np.testing.assert_allclose(ga, g["gabs_" + tag], rtol=2e-3, atol=1e-6)


def test_polar_module_path_matches_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "polar_net.npz"), allow_pickle=False)
    torch.manual_seed(1234)
    model = LearnICPWeightPolicy(_params(network_input_type="polar", network_output_type="polar"))
    model.train()
    x = torch.from_numpy(g["x_p"])
    scan = {"fft_data": x, "fft_cfar": torch.zeros_like(x), "raw_pc": torch.zeros(2, 4, 3)}
    m = model(scan, {"pc": torch.zeros(2, 4, 6)}, None, mask_only=True)
    np.testing.assert_allclose(m.detach().numpy(), g["mask_p"], atol=2e-6)


def test_checkpoint_helpers(tmp_path):
    p = _params()
    torch.manual_seed(3)
    a = LearnICPWeightPolicy(p)
    opt = trn.make_optimizer(a, p)
    for q in a.parameters():
        q.grad = torch.randn_like(q) * 1e-3
    opt.step()
    a._step = 17
    trn.save_checkpoint(str(tmp_path / "resume.pt"), a, opt, epoch=4, best_norm=0.25)
    torch.manual_seed(4)
    b = LearnICPWeightPolicy(p)
    opt_b = trn.make_optimizer(b, p)
    nxt, best = trn.load_checkpoint(str(tmp_path / "resume.pt"), b, opt_b)
    assert (nxt, best, b._step) == (5, 0.25, 17)
    for (k, va), vb in zip(a.state_dict().items(), b.state_dict().values()):
        assert torch.equal(va, vb), k
    sa, sb = opt.state_dict()["state"], opt_b.state_dict()["state"]
    assert sa.keys() == sb.keys() and all(torch.equal(sa[i]["exp_avg"], sb[i]["exp_avg"]) for i in sa)
    # a bare state_dict (the reference's best_policy.pt / epoch_N.pt) loads the parameters only
    torch.save(a.state_dict(), str(tmp_path / "epoch_0.pt"))
    c = LearnICPWeightPolicy(p)
    assert trn.load_checkpoint(str(tmp_path / "epoch_0.pt"), c) == (0, None)
    assert all(torch.equal(va, vc) for va, vc in zip(a.state_dict().values(), c.state_dict().values()))


def test_batch_norm_module_tree_matches_reference_keys(golden_dir):
    """params["batch_norm"]: Conv, ReLU, BN, Conv, ReLU, BN (icp_weight_policy.py:104-125) -- the state_dict keys
    (incl. running statistics) and the CPU mirror's mask equal the reference module's (unet_grads.npz, tag n)."""
    g = np.load(os.path.join(golden_dir, "unet_grads.npz"), allow_pickle=False)
    torch.manual_seed(1234)
    model = LearnICPWeightPolicy(_params(batch_norm=True))
    model.train()
    assert list(model.state_dict().keys()) == [str(k) for k in g["sd_names_n"]]
    xin = np.random.default_rng(99).uniform(0.01, 1, size=(2, 64, 64)).astype(np.float32)
    scan = {"fft_data": torch.from_numpy(xin), "fft_cfar": torch.zeros(2, 64, 64), "raw_pc": torch.zeros(2, 4, 3)}
    m = model(scan, {"pc": torch.zeros(2, 4, 6)}, None, mask_only=True)
    np.testing.assert_allclose(m.detach().numpy(), g["mask_n"], atol=2e-6)
