"""GPU twins of tests/test_round2_cpu.py: the Cartesian -> polar kernel, the Dataset's HIP-side work (CFAR cache,
augmentation + polar -> Cartesian), the product's loss functions with their mask terms, and the device guard."""
import os

import numpy as np
import pytest
import torch

from mm_masking_amd import _lib
from mm_masking_amd import icp_weight_dataset as ds
from mm_masking_amd import radar_utils as ru
from mm_masking_amd import train_icp_weights as trn
from oracle import radar_ref

from test_round2_cpu import _write_export, dataset_params

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def test_cart_to_polar_golden_and_full_size(golden_dir):
    g = np.load(os.path.join(golden_dir, "cart2polar.npz"), allow_pickle=False)
    out = ru.radar_cartesian_to_polar(torch.from_numpy(g["cart"]).to(DEV), torch.from_numpy(g["az"]).to(DEV), 0.0596,
                                      polar_pixel_shape=g["polar"].shape[1:])
    assert out.dtype == torch.float64 and out.is_cuda
    # The golden vectors were made on another host: torch's CPU sin / cos (SLEEF, dispatched per CPU model) differ
    # in the last bit between hosts, for the reference as for us -> a few ulp here; bit-exact against the oracle
    # evaluated on THIS host (below), which tests/test_round2_cpu.py pins bit-exactly to the reference.
    np.testing.assert_allclose(out.cpu().numpy(), g["polar"], rtol=0, atol=1e-12)
    assert np.array_equal(out.cpu().numpy(), radar_ref.radar_cartesian_to_polar(g["cart"], g["az"], 0.0596,
                                                                                polar_pixel_shape=g["polar"].shape[1:]))
    out2 = ru.radar_cartesian_to_polar(torch.from_numpy(g["cart2"]), torch.from_numpy(g["az2"]), 0.1, cart_resolution=0.3,
                                       polar_pixel_shape=(16, 120))
    assert not out2.is_cuda                                                            # CPU in -> CPU out, as every operator
    np.testing.assert_allclose(out2.numpy(), g["polar2"], rtol=0, atol=1e-12)
    assert np.array_equal(out2.numpy(), radar_ref.radar_cartesian_to_polar(g["cart2"], g["az2"], 0.1, cart_resolution=0.3,
                                                                           polar_pixel_shape=(16, 120)))
    with pytest.raises(RuntimeError, match=str(g["fp32_error"])):
        ru.radar_cartesian_to_polar(torch.zeros(1, 8, 8, device=DEV), torch.zeros(1, 4, device=DEV), 0.0596, polar_pixel_shape=(4, 10))
    # full size: 640 x 640 -> 400 x 3360, against the oracle
    rng = np.random.default_rng(8)
    cart = rng.random((2, 640, 640))
    az = np.sort(rng.uniform(0, 2 * np.pi, (2, 400)), axis=1)
    got = ru.radar_cartesian_to_polar(torch.from_numpy(cart).to(DEV), torch.from_numpy(az).to(DEV), 0.0596).cpu().numpy()
    assert got.shape == (2, 400, 3360)
    assert np.array_equal(got, radar_ref.radar_cartesian_to_polar(cart, az, 0.0596))
    # round trip polar -> Cartesian -> polar reproduces a smooth image inside the Cartesian footprint
    A, R = 400, 3360
    rr, aa = np.meshgrid(np.arange(R) * 0.0596, np.arange(A) * 2 * np.pi / A)
    pol = (0.5 + 0.4 * np.sin(rr / 9.0) * np.cos(3 * aa)).astype(np.float32)[None]
    azf = (np.arange(A) * 2 * np.pi / A).astype(np.float32)[None]
    c = ru.radar_polar_to_cartesian_diff(torch.from_numpy(pol).to(DEV), torch.from_numpy(azf).to(DEV), 0.0596)
    back = ru.radar_cartesian_to_polar(c.double(), torch.from_numpy(azf).to(DEV).double(), 0.0596).cpu().numpy()
    inner = slice(40, 1200)                    # ranges well inside the 76 m half-width of the 640-pixel image
    assert np.abs(back[0, :, inner] - pol[0, :, inner]).max() < 0.05


def test_dataset_item_cartesian_augment_and_cfar_cache(golden_dir, tmp_path):
    """The parts of ICPWeightDataset that run HIP kernels: the CFAR cache (cfar_mask) equals the one written by
    the reference's functions, and an augmented Cartesian item equals the reference's __getitem__ output."""
    g = np.load(os.path.join(golden_dir, "dataset_item.npz"), allow_pickle=False)
    pairs = _write_export(str(tmp_path), g, with_cfar=False)
    d = ds.ICPWeightDataset(pairs, dataset_params(network_input_type="cartesian", augment=True), dataset_type="train",
                            data_dir=str(tmp_path))
    assert d.augment is True
    for i in range(2):
        cached = ds.read_png_gray(d.loc_cfar_path_list[i])
        want = ds.read_png_gray(_bytes_to_file(tmp_path, g["cfar_png_%d" % i], "want_%d.png" % i))
        assert np.array_equal(cached, want)
    d.T_loc_init = torch.from_numpy(g["T_init"])
    for i in range(2):
        torch.manual_seed(500 + i)                     # the augmentation's yaw draw (icp_weight_dataset.py:430)
        it = d[i]
        pre = "c%d_" % i
        fd, fc = it["loc_data"]["fft_data"].numpy(), it["loc_data"]["fft_cfar"].numpy()
        assert list(fd.shape) == g[pre + "fft_shape"].tolist() == [640, 640]
        # (fp32 bilinear resampling: one ulp of a sampling coordinate times the pixel contrast, as in
        # test_gpu_radar.py::test_polar_to_cart_golden)
        np.testing.assert_allclose(fd[::9, ::9], g[pre + "fft_sub"], atol=2e-5)
        np.testing.assert_allclose(fc[::9, ::9], g[pre + "cfar_sub"], atol=2e-5)
        assert abs(fd.astype(np.float64).sum() - float(g[pre + "fft_sum"])) < 1e-4 * max(1.0, float(g[pre + "fft_sum"]))
        for key, val in (("raw_pc", it["loc_data"]["raw_pc"]), ("filtered_pc", it["loc_data"]["filtered_pc"]),
                         ("map_pc", it["map_data"]["pc"])):
            np.testing.assert_allclose(val.numpy(), g[pre + key], rtol=1e-5, atol=1e-6, err_msg=key)   # fp32 rotation, other host
    # and the batch feeds the policy's training step end to end
    from mm_masking_amd.icp_weight_policy import LearnICPWeightPolicy
    params = trn.default_params(DEV)
    params.update({"icp_type": "pt2pl", "icp_loss_fn": {"name": "huber", "metric": 1.0}, "max_iter": 3})
    batch = next(iter(torch.utils.data.DataLoader(d, batch_size=2, shuffle=False, num_workers=0)))
    torch.manual_seed(0)
    model = LearnICPWeightPolicy(params).to(DEV)
    model.train()
    opt = trn.make_optimizer(model, params)
    loss, _ = trn.train_step(model, batch, opt, trn.loss_weights_from(params), DEV)
    assert torch.isfinite(loss)


def _bytes_to_file(tmp_path, arr, name):
    path = os.path.join(str(tmp_path), name)
    arr.tofile(path)
    return path


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


def test_device_guard_refuses_foreign_device():
    """A tensor on another HIP device than the current one must not reach a kernel launch."""
    if torch.cuda.device_count() < 2:
        # one-GPU box: the guard itself, on a device index that is not the current one
        with pytest.raises(_lib.MmkError, match="current HIP device"):
            _lib.stream_ptr(torch.device("cuda", torch.cuda.current_device() + 1))
        return
    x = torch.zeros(1, 4, 400, device="cuda:1")
    with pytest.raises(_lib.MmkError, match="current HIP device"):
        ru.cfar_mask(x, 0.0596)
