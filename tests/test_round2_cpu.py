"""CPU-only checks added in round 2: the oracle's Cartesian -> polar restatement against the reference's
golden vectors; the product's own host-side functions (losses, load_radar, the Dataset over a plain export)
against the golden vectors — not only their oracle twins; the drop-in import shims."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from mm_masking_amd import icp_weight_dataset as ds
from mm_masking_amd import radar_utils as ru
from mm_masking_amd import train_icp_weights as trn
from oracle import radar_ref, train_ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_cart_to_polar_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "cart2polar.npz"), allow_pickle=False)
    out = radar_ref.radar_cartesian_to_polar(g["cart"], g["az"], 0.0596, polar_pixel_shape=g["polar"].shape[1:])
    assert out.dtype == np.float64 and np.array_equal(out, g["polar"])            # bit-exact
    out2 = radar_ref.radar_cartesian_to_polar(g["cart2"], g["az2"], 0.1, cart_resolution=0.3, polar_pixel_shape=(16, 120))
    assert np.array_equal(out2, g["polar2"])
    assert (g["polar"] == 0).mean() > 0.2 and (g["polar"] != 0).mean() > 0.2       # both inside and outside the image
    with pytest.raises(RuntimeError, match=str(g["fp32_error"])):
        radar_ref.radar_cartesian_to_polar(g["cart"].astype(np.float32), g["az"], 0.0596, polar_pixel_shape=(24, 200))
    # the product raises the reference's error for anything but fp64 before touching a device
    with pytest.raises(RuntimeError, match=str(g["fp32_error"])):
        ru.radar_cartesian_to_polar(torch.zeros(1, 8, 8), torch.zeros(1, 4), 0.0596, polar_pixel_shape=(4, 10))


# ----------------------------------------------------------------------------- product twins of the oracle's functions
def test_product_load_radar_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "radar_load.npz"), allow_pickle=False)
    fft, az, ts = ru.load_radar(g["png"])
    assert fft.dtype == np.float32 and np.array_equal(fft, g["fft"])
    assert np.array_equal(az, g["az"]) and np.array_equal(ts, g["ts"])


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


# ----------------------------------------------------------------------------- Dataset over a plain export (8f.2)
def _write_export(root, g, with_cfar=True):
    """The files ICPWeightDataset reads, from the arrays of dataset_item.npz."""
    map_seq, loc_seq = "boreas-map", "boreas-loc"
    pdir = os.path.join(root, "vtr_export", "radar_lidar", map_seq, loc_seq)
    os.makedirs(os.path.join(pdir, "scan"))
    os.makedirs(os.path.join(pdir, "map"))
    os.makedirs(os.path.join(root, "vtr_data", loc_seq, "radar"))
    cdir = os.path.join(root, "cfar", loc_seq, "polar", "1.0_0.09")
    os.makedirs(cdir)
    np.savez(os.path.join(pdir, "index.npz"), loc_stamp=g["loc_stamp"], map_stamp=g["map_stamp"], T_gt=g["T_gt"],
             T_map_sensor_robot=g["T_map_sensor_robot"])
    for i, (ls, ms) in enumerate(zip(g["loc_stamp"], g["map_stamp"])):
        g["raw_%d" % i].tofile(os.path.join(pdir, "scan", "%d_raw.bin" % ls))
        g["filt_%d" % i].tofile(os.path.join(pdir, "scan", "%d_filt.bin" % ls))
        g["map_%d" % i].tofile(os.path.join(pdir, "map", "%d.bin" % ms))
        g["png_%d" % i].tofile(os.path.join(root, "vtr_data", loc_seq, "radar", "%d.png" % ls))
        if with_cfar:
            g["cfar_png_%d" % i].tofile(os.path.join(cdir, "%d.png" % ls))
    return [[map_seq, loc_seq]]


def dataset_params(**over):
    p = {"map_sensor": "lidar", "loc_sensor": "radar", "random": False, "num_train": -1, "num_val": -1, "augment": False,
         "float_type": torch.float32, "use_gt": False, "gt_eye": True, "pos_std": 2.0, "rot_std": 0.6, "a_thresh": 1.0,
         "b_thresh": 0.09, "network_input_type": "polar", "max_loc_pts": 40, "max_map_pts": 80}
    p.update(over)
    return p


def test_dataset_item_matches_reference_polar(golden_dir, tmp_path):
    """ICPWeightDataset.__getitem__ (polar network input, no augmentation: no HIP call) on the export written from
    the fixture, against the dictionaries the reference's __getitem__ / load_graph_data returned."""
    g = np.load(os.path.join(golden_dir, "dataset_item.npz"), allow_pickle=False)
    pairs = _write_export(str(tmp_path), g)
    d = ds.ICPWeightDataset(pairs, dataset_params(), dataset_type="train", data_dir=str(tmp_path))
    assert len(d) == 2 and d.target_pad_val == 1000.0 and d.augment is False
    # T_init is exp of a seeded uniform draw (icp_weight_dataset.py:261-277): planar, inside the envelope
    for T in d.T_loc_init:
        assert abs(float(T[0, 3])) <= 2.0 and abs(float(T[1, 3])) <= 2.0 and float(T[2, 3]) == 0.0
        assert abs(np.arctan2(float(T[1, 0]), float(T[0, 0]))) <= 0.6 + 1e-6
    d.T_loc_init = torch.from_numpy(g["T_init"])             # the fixture's initial guesses
    for i in range(2):
        it = d[i]
        pre = "p%d_" % i
        for key, val in (("raw_pc", it["loc_data"]["raw_pc"]), ("filtered_pc", it["loc_data"]["filtered_pc"]),
                         ("fft_sub", it["loc_data"]["fft_data"]), ("cfar_sub", it["loc_data"]["fft_cfar"]),
                         ("map_pc", it["map_data"]["pc"]), ("T_init", it["transforms"]["T_ml_init"]),
                         ("T_gt", it["transforms"]["T_ml_gt"])):
            assert val.dtype == torch.float32
            assert np.array_equal(val.numpy(), g[pre + key]), key
        assert [it["loc_data"]["timestamp"], it["map_data"]["timestamp"]] == g[pre + "stamps"].tolist()
        assert it["map_data"]["pc"].shape == (80, 6) and it["loc_data"]["raw_pc"].shape == (40, 3)
    # a DataLoader batches the items into the Row-D dictionary of SURVEY.md §8a
    batch = next(iter(torch.utils.data.DataLoader(d, batch_size=2, shuffle=False, num_workers=0)))
    assert batch["loc_data"]["fft_data"].shape == (2, 40, 336) and batch["map_data"]["pc"].shape == (2, 80, 6)
    assert batch["transforms"]["T_ml_init"].shape == (2, 4, 4)
    # padding sizes derived from the data when params does not fix them
    d2 = ds.ICPWeightDataset(pairs, dataset_params(max_loc_pts=0, max_map_pts=0, num_val=1), dataset_type="test",
                             data_dir=str(tmp_path))
    assert len(d2) == 1 and d2.max_loc_pts == 30 and 0 < d2.max_map_pts <= 90
    assert d.get_item_from_loc_timestamp(int(g["loc_stamp"][1]))["loc_data"]["timestamp"] == int(g["loc_stamp"][1])


def test_png_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (13, 29), dtype=np.uint8)
    ds.write_png_gray(str(tmp_path / "a.png"), img)
    assert np.array_equal(ds.read_png_gray(str(tmp_path / "a.png")), img)


# ----------------------------------------------------------------------------- drop-in shims
def test_dropin_modules_accept_the_reference_import_lines():
    """mm_masking_amd/dropin on sys.path: the reference's own import statements
    (icp_weight_policy.py:6-7, icp_weight_dataset.py:11-12, train_icp_weights.py:3,5,17) resolve."""
    code = "\n".join([
        "import sys",
        "sys.path.insert(0, %r)" % os.path.join(ROOT, "mm_masking_amd", "dropin"),
        "sys.path.insert(1, %r)" % ROOT,
        "from dICP.ICP import ICP",
        "from radar_utils import load_pc_from_file, cfar_mask, extract_pc, radar_polar_to_cartesian_diff, "
        "radar_cartesian_to_polar, radar_polar_to_cartesian, extract_weights, point_to_cart_idx, "
        "form_cart_range_angle_grid, form_polar_range_grid",
        "from radar_utils import load_radar, cfar_mask, extract_pc, load_pc_from_file, radar_cartesian_to_polar, "
        "radar_polar_to_cartesian_diff, extract_bev_from_pts, point_to_cart_idx",
        "from icp_weight_dataset import ICPWeightDataset",
        "from icp_weight_policy import LearnICPWeightPolicy",
        "from radar_utils import extract_bev_from_pts",
        "import mm_masking_amd.icp_weight_policy as p, mm_masking_amd.radar_utils as r",
        "assert LearnICPWeightPolicy is p.LearnICPWeightPolicy and cfar_mask is r.cfar_mask",
        "assert ICP.__module__ == 'mm_masking_amd.dICP.ICP' and ICPWeightDataset.__module__ == 'mm_masking_amd.icp_weight_dataset'",
        "icp = ICP(icp_type='pt2pt', config_path='../external/dICP/config/dICP_config.yaml')",
        "assert icp.target_pad_val == 1000.0",
    ])
    subprocess.check_call([sys.executable, "-c", code], cwd="/")
