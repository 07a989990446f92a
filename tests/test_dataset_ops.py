"""Dataset-side tensor ops against golden vectors captured from the reference's
ICPWeightDataset methods (tests/golden/make_golden.py); CPU only."""
import os

import numpy as np
import torch

from mm_masking_amd import icp_weight_dataset as ds
from mm_masking_amd import synthetic


def _g(golden_dir):
    return np.load(os.path.join(golden_dir, "dataset_ops.npz"), allow_pickle=False)


def test_augment_data(golden_dir):
    g = _g(golden_dir)
    t = lambda k: torch.from_numpy(g[k])
    out = ds.augment_data(t("scan_raw"), t("scan_filt"), t("map6"), t("az"), t("fft"), t("cfar"), angle=float(g["angle"]))
    for got, key in zip(out, ("aug_raw", "aug_filt", "aug_map", "aug_az", "aug_fft", "aug_cfar")):
        np.testing.assert_allclose(got.numpy(), g[key], atol=2e-5)
    assert float(out[3][0]) == float(out[3].min())          # rolled so that the smallest azimuth leads
    # inputs are not modified in place
    np.testing.assert_array_equal(t("scan_raw").numpy(), g["scan_raw"])


def test_filter_map(golden_dir):
    g = _g(golden_dir)
    pts, nrm, T = torch.from_numpy(g["pts"]), torch.from_numpy(g["nrm"]), torch.from_numpy(g["T_gt"])
    pa, na = ds.filter_map(pts, nrm, T, return_aligned=True)
    pb, nb = ds.filter_map(pts, nrm, T, return_aligned=False)
    np.testing.assert_allclose(pa.numpy(), g["fa_p"], atol=1e-6)
    np.testing.assert_allclose(na.numpy(), g["fa_n"], atol=1e-6)
    np.testing.assert_array_equal(pb.numpy(), g["fb_p"])
    np.testing.assert_array_equal(nb.numpy(), g["fb_n"])
    assert 0 < pa.shape[0] < pts.shape[0]
    pl, _ = ds.filter_map(pts, nrm, T, loc_sensor="lidar", map_sensor="lidar")
    assert pl.shape[0] == pts.shape[0]


def test_padding_and_T_init():
    pts = torch.rand(7, 3)
    padded = ds.pad_scan(pts, 10)
    assert padded.shape == (10, 3) and torch.equal(padded[7:], torch.zeros(3, 3))
    m = ds.pad_map(torch.rand(5, 3), torch.rand(5, 3), 8, 1000.0)
    assert m.shape == (8, 6) and torch.all(m[5:] == 1000.0)
    gen = torch.Generator().manual_seed(0)
    for _ in range(20):
        T = ds.sample_T_init("train", 2.0, 0.6, generator=gen).numpy()
        yaw = np.arctan2(T[1, 0], T[0, 0])
        assert abs(yaw) <= 0.6 + 1e-6 and np.allclose(T[2], [0, 0, 1, 0]) and np.allclose(T[3], [0, 0, 0, 1])
        assert np.linalg.norm(T[:2, 3]) <= 2.0 * np.sqrt(2) * 1.2
    Tv = ds.sample_T_init("test", 2.0, 0.6, np_rng=np.random.default_rng(0))
    assert Tv.shape == (4, 4)
    # translation-first convention: a pure translation xi maps to T[:3,3] = rho
    np.testing.assert_allclose(synthetic.se3_exp([1.0, -2.0, 0, 0, 0, 0])[:3, 3], [1.0, -2.0, 0.0])
