"""Tensor-side pieces of the reference's ``ICPWeightDataset`` (mm_masking/icp_weight_dataset.py)
that define what enters the hot path: map filtering, padding to
batchable shapes, the random initial guess, and the rotation augmentation — plus ``ICPWeightDataset``, the reference's
Dataset over a plain-file export (Navtech PNG rows, .bin clouds) in place of the vtr3 pose graphs
/ pyboreas / ROS 2 bags, which are out of scope.  The tensor functions take and return plain
tensors so that the loader and the synthetic generator share them.  They are per-item host logic (the reference runs them in DataLoader workers) and use
ordinary tensor ops on whatever device the tensors live on.
"""
import math
import os

import numpy as np
import torch

from .synthetic import se3_exp


def filter_map(map_pts, map_norms, T_ml_gt, loc_sensor="radar", map_sensor="lidar", return_aligned=False,
               elevation_threshold=0.05, z_normal_threshold=0.9):
    """icp_weight_dataset.py:402-423: keep lidar map points within +-0.05 rad of elevation (seen from
    the ground-truth scan pose) whose normal is not (nearly) vertical."""
    pts_loc = (T_ml_gt[:3, :3] @ map_pts.T + T_ml_gt[:3, 3:4]).T
    nrm_loc = (T_ml_gt[:3, :3] @ map_norms.T).T
    elev = torch.abs(torch.atan2(pts_loc[:, 2], torch.sqrt(pts_loc[:, 0] * pts_loc[:, 0] + pts_loc[:, 1] * pts_loc[:, 1])))
    z_norm = torch.abs(nrm_loc[:, 2])
    if loc_sensor == "radar" and map_sensor == "lidar":
        valid = (elev <= elevation_threshold) & (z_norm <= z_normal_threshold)
    else:
        valid = torch.ones((pts_loc.shape[0],), dtype=torch.bool, device=map_pts.device)
    if return_aligned:
        return pts_loc[valid], nrm_loc[valid]
    return map_pts[valid], map_norms[valid]


def pad_scan(points, max_loc_pts, float_type=torch.float32):
    """icp_weight_dataset.py:377-381: zero rows up to ``max_loc_pts`` (they get weight 0 downstream)."""
    pad = torch.zeros((max_loc_pts - points.shape[0], 3), dtype=float_type, device=points.device)
    return torch.cat((points.to(float_type), pad), dim=0)


def pad_map(map_pts, map_norms, max_map_pts, target_pad_val, float_type=torch.float32):
    """icp_weight_dataset.py:394-398: (M,6) = xyz | normal, rows padded with ``target_pad_val``."""
    pad = target_pad_val * torch.ones((max_map_pts - map_pts.shape[0], 3), dtype=float_type, device=map_pts.device)
    return torch.cat((torch.cat((map_pts.to(float_type), pad), dim=0),
                      torch.cat((map_norms.to(float_type), pad), dim=0)), dim=1)


def sample_T_init(dataset_type="train", pos_std=2.0, rot_std=0.6, float_type=torch.float32, generator=None,
                  np_rng=None):
    """icp_weight_dataset.py:260-280: T_init = Exp(xi), xi = (x, y, 0, 0, 0, yaw); uniform in
    [-std, std] for training, normal(0, std) otherwise."""
    if dataset_type == "train":
        xi = 2 * torch.rand((6, 1), dtype=float_type, generator=generator) - 1
        xi[0:2] = pos_std * xi[0:2]
        xi[5] = rot_std * xi[5]
        xi[2:5] = 0.0
        xi = xi.reshape(6).double().numpy()
    else:
        rng = np_rng if np_rng is not None else np.random
        phi = rng.normal(0.0, rot_std)
        x = rng.normal(0.0, pos_std)
        y = rng.normal(0.0, pos_std)
        xi = np.array([x, y, 0.0, 0.0, 0.0, phi])
    return torch.tensor(se3_exp(xi), dtype=float_type)


def augment_data(scan_pc_raw, scan_pc_filt, map_pc, azimuths, fft_data, fft_cfar, float_type=torch.float32, angle=None):
    """icp_weight_dataset.py:425-452: one random yaw applied to the clouds (and normals) and, by
    shifting + rolling the azimuths, to the polar images.  ``angle`` (rad) fixes the draw."""
    if angle is None:
        angle = 2 * np.pi * torch.rand(1, dtype=float_type)
    angle = torch.as_tensor(angle, dtype=float_type).reshape(1)
    rot_mat = torch.tensor([[torch.cos(angle), -torch.sin(angle)], [torch.sin(angle), torch.cos(angle)]],
                           dtype=float_type, device=scan_pc_raw.device)
    scan_pc_raw, scan_pc_filt, map_pc = scan_pc_raw.clone(), scan_pc_filt.clone(), map_pc.clone()
    scan_pc_raw[:, :2] = torch.matmul(scan_pc_raw[:, :2], rot_mat)
    scan_pc_filt[:, :2] = torch.matmul(scan_pc_filt[:, :2], rot_mat)
    map_pc[:, :2] = torch.matmul(map_pc[:, :2], rot_mat)
    if map_pc.shape[1] == 6:
        map_pc[:, 3:5] = torch.matmul(map_pc[:, 3:5], rot_mat)
    azimuths = azimuths - angle.to(azimuths.device)
    azimuths = torch.where(azimuths < 0.0, azimuths + 2 * np.pi, azimuths)
    shift = -int(torch.argmin(azimuths).item())
    return (scan_pc_raw, scan_pc_filt, map_pc, torch.roll(azimuths, shift, dims=0), torch.roll(fft_data, shift, dims=0),
            torch.roll(fft_cfar, shift, dims=0))


# ----------------------------------------------------------------------------- the Dataset
def read_png_gray(path):
    """cv2.imread(path, cv2.IMREAD_GRAYSCALE) for the 8-bit single-channel PNGs of the Boreas radar
    folder and of the CFAR cache (icp_weight_dataset.py:186,336,342): the decoded bytes, (H,W) uint8."""
    from PIL import Image
    with Image.open(path) as im:
        if im.mode != "L":
            im = im.convert("L")
        return np.array(im, dtype=np.uint8)


def write_png_gray(path, img_u8):
    """cv2.imwrite of an 8-bit image (the CFAR cache, icp_weight_dataset.py:195)."""
    from PIL import Image
    Image.fromarray(np.ascontiguousarray(img_u8, dtype=np.uint8), mode="L").save(path, format="PNG")


def load_xyz_bin(path, cols=3):
    """float32 x ``cols`` per point (cols = 6: radar_utils.load_pc_from_file's layout)."""
    a = np.fromfile(path, dtype=np.float32)
    return a.reshape((len(a) // cols, cols))


class ICPWeightDataset(torch.utils.data.Dataset):
    """``ICPWeightDataset(loc_pairs, params, dataset_type)`` of mm_masking/icp_weight_dataset.py:28-362 over
    plain files.  Same constructor arguments and ``params`` keys (map_sensor, loc_sensor, random, num_train /
    num_val, augment, float_type, use_gt, gt_eye, pos_std, rot_std, a_thresh, b_thresh, network_input_type),
    same per-item dictionary (:357-362)::

        {'loc_data': {'raw_pc' (N,3), 'filtered_pc' (N,3), 'fft_data', 'fft_cfar', 'timestamp'},
         'map_data': {'pc' (M,6) padded with target_pad_val, 'timestamp'},
         'transforms': {'T_ml_init' (4,4), 'T_ml_gt' (4,4)}}

    What upstream pulls out of vtr3 pose graphs and pyboreas trajectory files (ROS 2 bags, not reproducible
    offline: SURVEY.md §2) is read from a one-off plain export instead::

        <data_dir>/vtr_data/<loc_seq>/radar/<loc_stamp>.png          Navtech polar scan, as in Boreas (:167)
        <data_dir>/cfar/<loc_seq>/polar/<a>_<b>/<loc_stamp>.png      CFAR cache, created when missing (:181-195)
        <data_dir>/vtr_export/<sensor_dir>/<map_seq>/<loc_seq>/index.npz
              loc_stamp (n) int64, map_stamp (n) int64, T_gt (n,4,4) = inv(T_loc) @ T_map (:215),
              T_map_sensor_robot (4,4) (:118-124)
        .../scan/<loc_stamp>_raw.bin, _filt.bin    float32 x 3 per point (extract_points_and_map's clouds)
        .../map/<map_stamp>.bin                    float32 x 6 per point, xyz | normal in the robot frame

    The per-item work is the reference's: PNG rows -> load_radar, zero / target_pad_val padding to the largest
    cloud of the set, map filtering by elevation and normal, rotation augmentation, polar -> Cartesian (HIP
    kernels through radar_utils; the CFAR cache is written with cfar_mask's HIP kernel).  Use it with
    ``DataLoader(..., num_workers=0)``: the items touch the GPU.
    """

    def __init__(self, loc_pairs, params=None, dataset_type="train", data_dir="../data"):
        from .dICP.ICP import ICP
        map_sensor, loc_sensor = params["map_sensor"], params["loc_sensor"]
        if dataset_type == "train":
            num_samples = params["num_train"]
            self.augment = params["augment"]
        else:
            num_samples = params["num_val"]
            self.augment = False
        self.float_type = params["float_type"]
        self.map_sensor, self.loc_sensor = map_sensor, loc_sensor
        self.gt_eye = params["gt_eye"]
        self.network_input_type = params["network_input_type"]
        self.loc_pairs = loc_pairs
        self.a_thresh, self.b_thresh = params["a_thresh"], params["b_thresh"]
        self.target_pad_val = ICP(icp_type="pt2pt", config_path="../external/dICP/config/dICP_config.yaml").target_pad_val
        if not params["random"]:
            np.random.seed(99)
            torch.manual_seed(99)
        if map_sensor == "lidar" and loc_sensor == "radar":
            sensor_dir_name = "radar_lidar"
        elif map_sensor == "radar" and loc_sensor == "radar":
            sensor_dir_name = "radar"
        elif map_sensor == "lidar" and loc_sensor == "lidar":
            sensor_dir_name = "lidar"
        else:
            raise ValueError("Invalid sensor combination")
        self.data_dir = data_dir
        self.polar_res = 0.0596
        self.samples = []                 # (pair index, loc_stamp, map_stamp)
        self.pair_dirs, self.T_map_sensor_robot = [], []
        self.loc_radar_path_list, self.loc_cfar_path_list = [], []
        T_gt, T_init = [], []
        self.max_loc_pts = int(params.get("max_loc_pts", 0))
        self.max_map_pts = int(params.get("max_map_pts", 0))
        scan_max = self.max_loc_pts == 0 or self.max_map_pts == 0
        radar_in = not (map_sensor == "lidar" and loc_sensor == "lidar")
        for pair_idx, (map_seq, loc_seq) in enumerate(loc_pairs):
            pdir = os.path.join(data_dir, "vtr_export", sensor_dir_name, map_seq, loc_seq)
            idx = np.load(os.path.join(pdir, "index.npz"), allow_pickle=False)
            self.pair_dirs.append(pdir)
            T_msr = torch.from_numpy(np.asarray(idx["T_map_sensor_robot"], dtype=np.float64)).type(self.float_type)
            self.T_map_sensor_robot.append(T_msr)
            for ii in range(len(idx["loc_stamp"])):
                loc_stamp, map_stamp = int(idx["loc_stamp"][ii]), int(idx["map_stamp"][ii])
                radar_path, cfar_path = 0, 0
                if radar_in:
                    radar_path = os.path.join(data_dir, "vtr_data", loc_seq, "radar", "%d.png" % loc_stamp)
                    if not os.path.exists(radar_path):
                        continue                                                   # :171-172
                    cfar_dir = os.path.join(data_dir, "cfar", loc_seq, "polar", "%s_%s" % (self.a_thresh, self.b_thresh))
                    os.makedirs(cfar_dir, exist_ok=True)
                    cfar_path = os.path.join(cfar_dir, "%d.png" % loc_stamp)
                    if not os.path.exists(cfar_path):
                        self._write_cfar(radar_path, cfar_path)
                T_gt_idx = torch.tensor(np.asarray(idx["T_gt"][ii]), dtype=self.float_type)
                if scan_max:
                    raw, _, mp, mn = self._read_clouds(pdir, loc_stamp, map_stamp)
                    mps, mns = self._to_sensor_frame(torch.from_numpy(mp), torch.from_numpy(mn), T_msr)
                    kept, _ = filter_map(mps, mns, T_gt_idx, loc_sensor, map_sensor)
                    self.max_loc_pts = max(self.max_loc_pts, raw.shape[0])
                    self.max_map_pts = max(self.max_map_pts, kept.shape[0])
                if params["use_gt"]:                                               # :248-252
                    T_init_idx = torch.eye(4, dtype=self.float_type) if self.gt_eye else T_gt_idx.clone()
                else:
                    T_rand = sample_T_init(dataset_type, params["pos_std"], params["rot_std"], self.float_type)
                    T_init_idx = T_rand if self.gt_eye else (T_rand.double() @ T_gt_idx.double()).type(self.float_type)
                self.samples.append((pair_idx, loc_stamp, map_stamp))
                T_gt.append(T_gt_idx)
                T_init.append(T_init_idx)
                self.loc_radar_path_list.append(radar_path)
                self.loc_cfar_path_list.append(cfar_path)
                if num_samples > 0 and len(self.samples) >= num_samples:
                    break
        if not self.samples:
            raise ValueError("ICPWeightDataset: no usable sample under %s" % data_dir)
        self.T_loc_gt = torch.stack(T_gt)
        self.T_loc_init = torch.stack(T_init)

    # -- helpers
    def _write_cfar(self, radar_path, cfar_path):
        """icp_weight_dataset.py:185-195: hard GO-CFAR of the polar scan, cached as an 8-bit PNG."""
        from . import radar_utils as ru
        fft, _, _ = ru.load_radar(read_png_gray(radar_path))
        fft = torch.tensor(fft, dtype=self.float_type).unsqueeze(0)
        cfar = ru.cfar_mask(fft, self.polar_res, a_thresh=self.a_thresh, b_thresh=self.b_thresh, diff=False)
        write_png_gray(cfar_path, np.rint(255.0 * cfar.squeeze(0).cpu().numpy()).astype(np.uint8))

    @staticmethod
    def _read_clouds(pdir, loc_stamp, map_stamp):
        raw = load_xyz_bin(os.path.join(pdir, "scan", "%d_raw.bin" % loc_stamp), 3)
        filt = load_xyz_bin(os.path.join(pdir, "scan", "%d_filt.bin" % loc_stamp), 3)
        m = load_xyz_bin(os.path.join(pdir, "map", "%d.bin" % map_stamp), 6)
        return raw, filt, np.ascontiguousarray(m[:, :3]), np.ascontiguousarray(m[:, 3:6])

    @staticmethod
    def _to_sensor_frame(map_pts, map_norms, T):
        return (T[:3, :3] @ map_pts.T + T[:3, 3:4]).T, (T[:3, :3] @ map_norms.T).T       # :387-388

    def __len__(self):
        return len(self.samples)

    def load_graph_data(self, idx, T_ml_gt):
        """icp_weight_dataset.py:364-400 with the clouds read from the export instead of the pose graph."""
        pair_idx, loc_stamp, map_stamp = self.samples[idx]
        raw, filt, mp, mn = self._read_clouds(self.pair_dirs[pair_idx], loc_stamp, map_stamp)
        scan_pc_raw = pad_scan(torch.from_numpy(raw), self.max_loc_pts, self.float_type)
        scan_pc_filt = pad_scan(torch.from_numpy(filt), self.max_loc_pts, self.float_type)
        mps, mns = self._to_sensor_frame(torch.from_numpy(mp), torch.from_numpy(mn), self.T_map_sensor_robot[pair_idx])
        mps, mns = filter_map(mps, mns, T_ml_gt, self.loc_sensor, self.map_sensor, return_aligned=self.gt_eye)
        map_pc = pad_map(mps, mns, self.max_map_pts, self.target_pad_val, self.float_type)
        return scan_pc_raw, scan_pc_filt, map_pc, loc_stamp, map_stamp

    def __getitem__(self, index):
        """icp_weight_dataset.py:323-362."""
        from . import radar_utils as ru
        T_init = self.T_loc_init[index]
        T_ml_gt = self.T_loc_gt[index]
        scan_pc_raw, scan_pc_filt, map_pc, loc_stamp, map_stamp = self.load_graph_data(index, T_ml_gt)
        assert scan_pc_raw.shape == scan_pc_filt.shape, "Raw and filtered pointclouds dont match!"
        if not (self.map_sensor == "lidar" and self.loc_sensor == "lidar"):
            fft_data, azimuths, _ = ru.load_radar(read_png_gray(self.loc_radar_path_list[index]))
            fft_data = torch.tensor(fft_data, dtype=self.float_type)
            azimuths = torch.tensor(azimuths, dtype=self.float_type)
            fft_cfar = torch.tensor(read_png_gray(self.loc_cfar_path_list[index]), dtype=self.float_type) / 255.0
            if self.augment:
                scan_pc_raw, scan_pc_filt, map_pc, azimuths, fft_data, fft_cfar = augment_data(
                    scan_pc_raw, scan_pc_filt, map_pc, azimuths, fft_data, fft_cfar, self.float_type)
            if self.network_input_type == "cartesian":
                fft_data, fft_cfar = ru._polar_to_cart_pair(fft_data.unsqueeze(0), fft_cfar.unsqueeze(0),
                                                            azimuths.unsqueeze(0), self.polar_res)
                fft_data, fft_cfar = fft_data.squeeze(0), fft_cfar.squeeze(0)
        else:
            fft_data, fft_cfar = 0.0, 0.0
        loc_data = {"raw_pc": scan_pc_raw, "filtered_pc": scan_pc_filt, "fft_data": fft_data, "fft_cfar": fft_cfar,
                    "timestamp": loc_stamp}
        map_data = {"pc": map_pc, "timestamp": map_stamp}
        return {"loc_data": loc_data, "map_data": map_data, "transforms": {"T_ml_init": T_init, "T_ml_gt": T_ml_gt}}

    def get_item_from_loc_timestamp(self, loc_stamp_req):
        """icp_weight_dataset.py:454-495."""
        index = [i for i, s in enumerate(self.samples) if s[1] == int(loc_stamp_req)]
        assert index != [], "loc_stamp_req not found in dataset"
        return self[index[0]]
