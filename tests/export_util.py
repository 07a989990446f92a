"""Writes a full-size synthetic plain-file export for ``ICPWeightDataset`` (the layout its docstring describes):
Navtech-format PNG rows (8 timestamp bytes, 2 encoder bytes, 1 pad byte, 3360 power bytes per azimuth:
/root/reference/mm_masking/radar_utils.py:20-27), scan clouds and a lidar map per sample.  Test infrastructure."""
import os

import numpy as np
import torch

from mm_masking_amd import icp_weight_dataset as ds
from mm_masking_amd import synthetic


def navtech_rows(fft, az):
    """(400,3360) power in [0,1] + azimuths (rad) -> (400,3371) uint8 rows."""
    A = fft.shape[0]
    rows = np.zeros((A, 11 + fft.shape[1]), np.uint8)
    ts = (1_600_000_000_000_000 + np.arange(A, dtype=np.int64) * 625)
    rows[:, :8] = np.frombuffer(ts.tobytes(), dtype=np.uint8).reshape(A, 8)
    enc = np.round(az.astype(np.float64) / (2 * np.pi / 5600)).astype(np.uint16)
    rows[:, 8:10] = np.frombuffer(enc.tobytes(), dtype=np.uint8).reshape(A, 2)
    rows[:, 11:] = np.round(fft * 255.0).astype(np.uint8)
    return rows


def write_synthetic_export(root, n, n_scan=4000, m_valid=20000, first=9000):
    """n samples under ``root``; returns loc_pairs.  Scan clouds: ``n_scan`` random points on the map (the loader's cost
    does not depend on where they are), 3 floats each; map: the synthetic lidar submap, xyz | normal."""
    map_seq, loc_seq = "boreas-map", "boreas-loc"
    pdir = os.path.join(root, "vtr_export", "radar_lidar", map_seq, loc_seq)
    os.makedirs(os.path.join(pdir, "scan"), exist_ok=True)
    os.makedirs(os.path.join(pdir, "map"), exist_ok=True)
    rdir = os.path.join(root, "vtr_data", loc_seq, "radar")
    os.makedirs(rdir, exist_ok=True)
    loc_stamp = np.arange(n, dtype=np.int64) + 1_700_000_000
    map_stamp = np.arange(n, dtype=np.int64) + 1_600_000_000
    rng = np.random.default_rng(5)
    for i in range(n):
        p = synthetic.make_pair(first + i, m_valid=m_valid, m_pad=m_valid)
        ds.write_png_gray(os.path.join(rdir, "%d.png" % loc_stamp[i]), navtech_rows(p["fft_polar"], p["azimuths"]))
        m = p["map_pc"][:m_valid].astype(np.float32)
        m.tofile(os.path.join(pdir, "map", "%d.bin" % map_stamp[i]))
        sel = rng.choice(m_valid, size=n_scan, replace=False)
        sc = np.ascontiguousarray(m[sel, :3] + rng.normal(0, 0.05, (n_scan, 3)).astype(np.float32) * np.array([1, 1, 0], np.float32))
        sc.tofile(os.path.join(pdir, "scan", "%d_raw.bin" % loc_stamp[i]))
        sc.tofile(os.path.join(pdir, "scan", "%d_filt.bin" % loc_stamp[i]))
    np.savez(os.path.join(pdir, "index.npz"), loc_stamp=loc_stamp, map_stamp=map_stamp,
             T_gt=np.tile(np.eye(4), (n, 1, 1)), T_map_sensor_robot=np.eye(4))
    return [[map_seq, loc_seq]]


def write_fixture_export(root, g, with_cfar=True):
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


def assert_same(a, b, path=""):
    if isinstance(a, dict):
        assert a.keys() == b.keys(), (path, a.keys(), b.keys())
        for k in a:
            assert_same(a[k], b[k], path + "/" + str(k))
    elif torch.is_tensor(a):
        assert a.dtype == b.dtype and torch.equal(a.cpu(), b.cpu()), path
    else:
        assert a == b, path
