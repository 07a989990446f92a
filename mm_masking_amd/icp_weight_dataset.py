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
import threading

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
    kernels through radar_utils; the CFAR cache is written with cfar_mask's HIP kernel).

    Two item modes:
      * default (``params["batched_prepare"]`` absent / False): ``__getitem__`` returns the reference's finished item -- for a
        Cartesian network that is one batch-1 polar -> Cartesian launch per item on the GPU, so such a Dataset is used with
        ``DataLoader(..., num_workers=0)``;
      * ``params["batched_prepare"] = True`` (what keeps a ~10 ms training step fed; the reference runs its per-item work in
        4 DataLoader workers, train_icp_weights.py:454-455): ``__getitem__`` is CPU-ONLY and safe for ``num_workers=4`` --
        decoded PNG rows (from a decoded-byte cache beside the CFAR cache: PNG inflate is ~10 ms per scan), load_radar's
        azimuths, clouds, padding, map filtering, augmentation -- and leaves the images as uint8 polar rows
        (``fft_u8``, ``cfar_u8``, ``azimuths`` in ``loc_data``); ``finish_batch`` then does bytes / 255 and ONE batched
        polar -> Cartesian launch for the whole batch on the device.  ``DeviceLoader`` wraps both steps and stages batch
        i + 1 on a side stream while batch i trains.  Finished batches are equal to ``default_collate`` of the default
        mode's items (tests/test_loader_cpu.py, tests/test_gpu_loader.py).
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
        self._native_info = {}
        self.batched_prepare = bool(params.get("batched_prepare", False))
        self.decoded_cache = bool(params.get("decoded_cache", self.batched_prepare))
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

    def _cloud_sources(self, index):
        """The export's cloud files of sample ``index`` (what _read_clouds opens)."""
        pair_idx, loc_stamp, map_stamp = self.samples[index]
        pdir = self.pair_dirs[pair_idx]
        return [os.path.join(pdir, "scan", "%d_raw.bin" % loc_stamp), os.path.join(pdir, "scan", "%d_filt.bin" % loc_stamp),
                os.path.join(pdir, "map", "%d.bin" % map_stamp)]

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
        if self.batched_prepare and not (self.map_sensor == "lidar" and self.loc_sensor == "lidar"):
            return self._cpu_item(index, T_init, T_ml_gt, scan_pc_raw, scan_pc_filt, map_pc, loc_stamp, map_stamp)
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

    # -- worker-safe items (params["batched_prepare"])
    def _decoded(self, png_path):
        """Decoded rows of an 8-bit PNG, through a raw-byte cache next to the file (``<name>.png.u8`` = uint32 height, uint32
        width, then the rows): inflating a 400 x 3371 Navtech scan costs ~10 ms of a worker's time, reading 1.3 MB from
        the page cache ~0.2 ms.  Written once (atomically: several workers may race for the same file)."""
        if not self.decoded_cache:
            return read_png_gray(png_path)
        cpath = png_path + ".u8"
        try:
            if os.path.getmtime(cpath) >= os.path.getmtime(png_path):
                raw = np.memmap(cpath, dtype=np.uint8, mode="r")      # mapped, not read: the one copy is the caller's slice
                h, w = np.frombuffer(raw[:8].tobytes(), dtype=np.uint32)
                if raw.size == 8 + int(h) * int(w):
                    return raw[8:].reshape(int(h), int(w))
        except OSError:
            pass
        img = read_png_gray(png_path)
        tmp = "%s.%d.tmp" % (cpath, os.getpid())
        try:
            with open(tmp, "wb") as f:
                f.write(np.array(img.shape, dtype=np.uint32).tobytes())
                f.write(np.ascontiguousarray(img).tobytes())
            os.replace(tmp, cpath)
        except OSError:
            pass                      # read-only data directory: decode every time
        return img

    def _cpu_item(self, index, T_init, T_ml_gt, scan_pc_raw, scan_pc_filt, map_pc, loc_stamp, map_stamp):
        """The CPU-only part of __getitem__ (icp_weight_dataset.py:336-348): no float image, no GPU call."""
        raw = self._decoded(self.loc_radar_path_list[index])
        # load_radar (radar_utils.py:20-27) without the float image: azimuths from the encoder bytes, power bytes kept as uint8
        azimuths = torch.tensor(np.frombuffer(raw[:, 8:10].tobytes(), dtype=np.uint16) * (2 * np.pi / 5600), dtype=self.float_type)
        fft_u8 = torch.from_numpy(np.array(raw[:, 11:], dtype=np.uint8, order="C"))
        cfar_u8 = torch.from_numpy(np.array(self._decoded(self.loc_cfar_path_list[index]), dtype=np.uint8, order="C"))
        if self.augment:
            scan_pc_raw, scan_pc_filt, map_pc, azimuths, fft_u8, cfar_u8 = augment_data(
                scan_pc_raw, scan_pc_filt, map_pc, azimuths, fft_u8, cfar_u8, self.float_type)
        loc_data = {"raw_pc": scan_pc_raw, "filtered_pc": scan_pc_filt, "fft_u8": fft_u8, "cfar_u8": cfar_u8,
                    "azimuths": azimuths, "timestamp": loc_stamp}
        return {"loc_data": loc_data, "map_data": {"pc": map_pc, "timestamp": map_stamp},
                "transforms": {"T_ml_init": T_init, "T_ml_gt": T_ml_gt}}

    # -- native item path (DeviceLoader, thread mode): bytes go from the page cache into the batch's pinned buffers in C
    def _prepared_clouds(self, index):
        """Path of the sample's prepared clouds -- raw (N,3) | filtered (N,3) | map (M,6) fp32, padded, the map filtered
        and in the sensor frame exactly as load_graph_data returns them (they depend on the sample alone, not on the
        epoch) -- written on first use next to the export."""
        pair_idx, loc_stamp, map_stamp = self.samples[index]
        pdir = os.path.join(self.pair_dirs[pair_idx], "prepared")
        # the file name carries a digest of EVERYTHING load_graph_data's result depends on besides the sample's stamps: padding
        # sizes and value, sensors (filter_map), gt_eye, float type, the ground-truth pose, the sensor-robot transform, and
        # size + mtime of the source cloud files -- a changed configuration or re-exported data gets a new file instead of
        # stale clouds
        import hashlib
        h = hashlib.sha1()
        h.update(repr((self.max_loc_pts, self.max_map_pts, float(self.target_pad_val), self.loc_sensor, self.map_sensor, bool(self.gt_eye),
                       str(self.float_type))).encode())
        h.update(self.T_loc_gt[index].to(torch.float64).contiguous().numpy().tobytes())
        h.update(self.T_map_sensor_robot[pair_idx].to(torch.float64).contiguous().numpy().tobytes())
        for src in self._cloud_sources(index):
            try:
                st = os.stat(src)
                h.update(("%s:%d:%d" % (os.path.basename(src), st.st_size, st.st_mtime_ns)).encode())
            except OSError:
                h.update(("%s:missing" % os.path.basename(src)).encode())
        path = os.path.join(pdir, "%d_%d_%s.f32" % (loc_stamp, map_stamp, h.hexdigest()[:16]))
        if not os.path.exists(path):
            raw, filt, mp, _, _ = self.load_graph_data(index, self.T_loc_gt[index])
            os.makedirs(pdir, exist_ok=True)
            tmp = "%s.%d.%d.tmp" % (path, os.getpid(), threading.get_ident())
            with open(tmp, "wb") as f:
                for t in (raw, filt, mp):
                    f.write(t.to(torch.float32).contiguous().numpy().tobytes())
            os.replace(tmp, path)
        return path

    def native_item_spec(self):
        """Shapes / dtypes of one item of the native path (what DeviceLoader allocates its pinned batch buffers from)."""
        raw = self._decoded(self.loc_radar_path_list[0])
        A, R = raw.shape[0], raw.shape[1] - 11
        f = self.float_type
        return {"loc_data": {"raw_pc": ((self.max_loc_pts, 3), f), "filtered_pc": ((self.max_loc_pts, 3), f),
                             "fft_u8": ((A, R), torch.uint8), "cfar_u8": ((A, R), torch.uint8), "azimuths": ((A,), f),
                             "aug_cs": ((2,), f), "timestamp": ((), torch.int64)},
                "map_data": {"pc": ((self.max_map_pts, 6), f), "timestamp": ((), torch.int64)},
                "transforms": {"T_ml_init": ((4, 4), f), "T_ml_gt": ((4, 4), f)}}

    def fill_batch(self, indices, bufs, threads=4, generator=None):
        """Items ``indices`` written into rows 0.. of the batch buffers ``bufs`` (native_item_spec's layout): the same items as
        ``__getitem__`` of the ``batched_prepare`` mode, except that the augmentation's rotation of the clouds is left to the
        device (``aug_cs`` = (cos, sin) of the drawn yaw; finish_batch applies it).  ONE interpreter thread: the per-item share
        that needs Python (paths and caches resolved once per sample, the 400 encoder counts -> azimuths in numpy, the
        augmentation's yaw drawn in item order -- reproducible under a seed, which a pool of worker threads is not) runs
        here; every byte of the tensors is then moved by one mmk_host_read_rows_batch call whose ``threads`` C threads share
        the jobs (page cache -> pinned memory, column cut and azimuth roll on the way; GIL released for the whole batch).
        ``generator``: the torch.Generator the yaw is drawn from (DeviceLoader passes its own, so that the draws do not
        interleave with whatever the training thread takes from the global generator; None = the global one)."""
        from . import _lib
        L = _lib.lib()
        rd = L.mmk_host_read_rows
        loc, mp, tr = bufs["loc_data"], bufs["map_data"], bufs["transforms"]
        A, R = loc["fft_u8"].shape[1], loc["fft_u8"].shape[2]
        nb_scan, nb_map = self.max_loc_pts * 12, self.max_map_pts * 24
        nb = len(indices)
        jobs = (_lib.ReadJob * (5 * nb))()
        keep = []                                      # (the path bytes must outlive the calls)
        az_all, cs_all = loc["azimuths"].numpy(), loc["aug_cs"].numpy()
        # ---- pass 1 (one library call for the batch): the 2-byte encoder column of every item's radar rows
        enc = np.empty((nb, A, 2), np.uint8)
        infos = []
        for j, index in enumerate(indices):
            info = self._native_info.get(index)
            if info is None:
                assert self.float_type == torch.float32, "native loader path: float32 items"
                rpath, cpath = self.loc_radar_path_list[index], self.loc_cfar_path_list[index]
                for png in (rpath, cpath):
                    if not os.path.exists(png + ".u8") or os.path.getmtime(png + ".u8") < os.path.getmtime(png):
                        self._decoded(png)
                info = ((rpath + ".u8").encode(), (cpath + ".u8").encode(), self._prepared_clouds(index).encode())
                self._native_info[index] = info
            infos.append(info)
            keep.append(info)
            q = jobs[j]
            q.path, q.header_bytes, q.rows, q.row_bytes, q.col0, q.ncols, q.roll, q.dst = info[0], 8, A, R + 11, 8, 2, 0, enc[j].ctypes.data
        if L.mmk_host_read_rows_batch(jobs, nb, int(threads)) != 0:
            raise _lib.MmkError(L.mmk_last_error().decode())
        # ---- the interpreter's share, vectorised over the batch: encoder counts -> azimuths, the augmentation's yaw (drawn in
        # item order from one call), the roll that brings the smallest azimuth to row 0 (augment_data, icp_weight_dataset.py:425-452)
        az = (enc.reshape(nb, A * 2).view(np.uint16).reshape(nb, A) * (2 * np.pi / 5600)).astype(np.float32)
        shifts = np.zeros(nb, np.int64)
        cs_all[:nb, 0], cs_all[:nb, 1] = 1.0, 0.0
        if self.augment:
            angles = 2 * np.pi * torch.rand(nb, dtype=self.float_type, generator=generator)
            cs_all[:nb, 0], cs_all[:nb, 1] = torch.cos(angles).numpy(), torch.sin(angles).numpy()
            az = az - angles.numpy().astype(np.float32)[:, None]
            az = np.where(az < 0.0, az + np.float32(2 * np.pi), az)
            shifts = -np.argmin(az, axis=1)
            rows = (np.arange(A)[None, :] - shifts[:, None]) % A          # np.roll(az[j], shifts[j]) for every j
            az = np.take_along_axis(az, rows, axis=1)
        az_all[:nb] = az
        # ---- pass 2: every byte of the tensors
        n = 0
        ts_loc, ts_map = loc["timestamp"].numpy(), mp["timestamp"].numpy()
        Ti, Tg = tr["T_ml_init"].numpy(), tr["T_ml_gt"].numpy()
        strides = [(t.data_ptr(), t.stride(0) * t.element_size()) for t in (loc["fft_u8"], loc["cfar_u8"], loc["raw_pc"], loc["filtered_pc"], mp["pc"])]
        for j, index in enumerate(indices):
            rfile, cfile, prep = infos[j]
            shift = int(shifts[j])
            for k, (path, hdr, rows_, rb, c0, nc, roll) in enumerate((
                    (rfile, 8, A, R + 11, 11, R, shift),
                    (cfile, 8, A, R, 0, R, shift),
                    (prep, 0, 1, nb_scan, 0, nb_scan, 0),
                    (prep, nb_scan, 1, nb_scan, 0, nb_scan, 0),
                    (prep, 2 * nb_scan, 1, nb_map, 0, nb_map, 0))):
                q = jobs[n]
                q.path, q.header_bytes, q.rows, q.row_bytes, q.col0, q.ncols, q.roll = path, hdr, rows_, rb, c0, nc, roll
                q.dst = strides[k][0] + j * strides[k][1]
                n += 1
            _, loc_stamp, map_stamp = self.samples[index]
            ts_loc[j] = loc_stamp
            ts_map[j] = map_stamp
            Ti[j] = self.T_loc_init[index].numpy()
            Tg[j] = self.T_loc_gt[index].numpy()
        if L.mmk_host_read_rows_batch(jobs, n, int(threads)) != 0:
            raise _lib.MmkError(L.mmk_last_error().decode())
        del keep

    def get_item_from_loc_timestamp(self, loc_stamp_req):
        """icp_weight_dataset.py:454-495."""
        index = [i for i, s in enumerate(self.samples) if s[1] == int(loc_stamp_req)]
        assert index != [], "loc_stamp_req not found in dataset"
        return self[index[0]]


_LUT = {}


def _unit_lut(device):
    """bytes / 255 as load_radar computes it (radar_utils.py:26: numpy fp32 division), 256 entries, on ``device``."""
    key = str(device)
    if key not in _LUT:
        _LUT[key] = torch.from_numpy(np.divide(np.arange(256, dtype=np.uint8), 255.0, dtype=np.float32)).to(device)
    return _LUT[key]


def _bytes_to_unit(u8, device, non_blocking):
    u8 = u8.to(device, non_blocking=non_blocking).contiguous()
    if u8.is_cuda:
        from . import _lib
        out = torch.empty(u8.shape, dtype=torch.float32, device=u8.device)
        _lib.check(_lib.lib().mmk_u8_to_float(_lib.ptr(u8), _lib.ptr(_unit_lut(u8.device)), u8.numel(), _lib.ptr(out),
                                              _lib.stream_ptr(u8.device)))
        return out
    return _unit_lut(u8.device)[u8.long()]


def finish_batch(batch, device, network_input_type="cartesian", float_type=torch.float32, polar_res=0.0596, non_blocking=True):
    """Device half of the worker-safe loader: a collated batch of ``batched_prepare`` items -> the reference's batch
    dictionary (icp_weight_dataset.py:357-362) with every tensor on ``device``.  bytes / 255 is load_radar's fp32 division
    (radar_utils.py:26) and the CFAR cache's (icp_weight_dataset.py:343), taken from a 256-entry table the host computed
    with that very division (mmk_u8_to_float); the polar -> Cartesian resampling (icp_weight_dataset.py:351-352) is ONE
    batched launch instead of one per item; ``aug_cs`` (native item path) is the augmentation's rotation of the clouds
    (icp_weight_dataset.py:435-443), applied here as [x y] @ [[c, -s], [s, c]]."""
    from . import radar_utils as ru
    device = torch.device(device)
    loc = batch["loc_data"]
    to = lambda v: v.to(device, non_blocking=non_blocking) if torch.is_tensor(v) else v       # noqa: E731
    if "fft_u8" not in loc:                      # already a finished (default-mode) batch: move it
        return {k: {kk: to(vv) for kk, vv in v.items()} for k, v in batch.items()}
    assert float_type == torch.float32
    fft = _bytes_to_unit(loc["fft_u8"], device, non_blocking)
    cfar = _bytes_to_unit(loc["cfar_u8"], device, non_blocking)
    az = to(loc["azimuths"])
    if network_input_type == "cartesian":
        fft, cfar = ru._polar_to_cart_pair(fft, cfar, az.contiguous(), polar_res)
    raw_pc, filt_pc, map_pc = to(loc["raw_pc"]), to(loc["filtered_pc"]), to(batch["map_data"]["pc"])
    if "aug_cs" in loc:
        cs = to(loc["aug_cs"])
        c, s = cs[:, 0].view(-1, 1), cs[:, 1].view(-1, 1)

        def rot(t, k):
            x, y = t[:, :, k].clone(), t[:, :, k + 1].clone()
            t[:, :, k] = x * c + y * s
            t[:, :, k + 1] = y * c - x * s
        rot(raw_pc, 0), rot(filt_pc, 0), rot(map_pc, 0)
        if map_pc.shape[2] == 6:
            rot(map_pc, 3)
    stamp = lambda v: v.clone() if torch.is_tensor(v) else v          # noqa: E731  (host tensors: not views of a re-used buffer)
    return {"loc_data": {"raw_pc": raw_pc, "filtered_pc": filt_pc, "fft_data": fft, "fft_cfar": cfar,
                         "timestamp": stamp(loc["timestamp"])},
            "map_data": {"pc": map_pc, "timestamp": stamp(batch["map_data"]["timestamp"])},
            "transforms": {k: to(v) for k, v in batch["transforms"].items()}}


class DeviceLoader:
    """``DataLoader(dataset, num_workers=4)`` (train_icp_weights.py:454-455) + ``finish_batch``: iterating yields the reference's
    batch dictionaries on the device.  The per-item CPU work (``__getitem__`` of a ``batched_prepare`` Dataset) runs in
    ``num_workers`` parallel workers; the main thread stages batch i + 1 (H2D copies, bytes / 255, one polar -> Cartesian
    launch) on a side stream while the caller trains on batch i, and hands it over with a stream wait -- no host
    synchronisation.

    ``mode="threads"`` (default): the workers are threads of this process; every tensor of an item is written into the
    batch's pinned host buffers by ONE C call (mmk_host_read_rows through ctypes, GIL released: page cache -> pinned memory,
    column cut and augmentation roll on the way; clouds from a prepared-cloud cache), the interpreter only draws the yaw
    and decodes the 400 encoder counts.  A batch of 32 is ~105 MB; worker PROCESSES have to push it through shared memory
    and the main process's unpickling (measured 0.45 k items/s against the threads' rate in gpurun_out/r03_loader.json).
    ``mode="processes"``: torch's DataLoader with ``num_workers`` worker processes running the CPU-only ``__getitem__`` and
    pinned collation, as upstream.  Same batches either way, up to the rounding of the augmentation's cloud rotation,
    which the thread mode leaves to the device (tests/test_loader_cpu.py, tests/test_gpu_loader.py)."""

    def __init__(self, dataset, batch_size, device, num_workers=4, shuffle=False, drop_last=False, prefetch_factor=2,
                 persistent_workers=True, mode="threads", passes=1, seed=None):
        if not getattr(dataset, "batched_prepare", False):
            raise ValueError("DeviceLoader needs a Dataset built with params['batched_prepare'] = True (CPU-only items)")
        if mode not in ("threads", "processes"):
            raise ValueError("mode must be 'threads' or 'processes'")
        if passes != 1 and (mode != "threads" or num_workers == 0):
            raise ValueError("passes > 1 (several passes over the data in ONE iteration, the pipeline kept full across them) "
                             "is a feature of the thread mode")
        self.passes = int(passes)
        self.dataset, self.device, self.mode = dataset, torch.device(device), mode
        self.batch_size, self.shuffle, self.drop_last, self.num_workers = int(batch_size), shuffle, drop_last, int(num_workers)
        self.loader = None
        if mode == "processes" or num_workers == 0:
            kw = {}
            if num_workers > 0:
                kw = {"prefetch_factor": prefetch_factor, "persistent_workers": persistent_workers}
            self.loader = torch.utils.data.DataLoader(dataset, batch_size=batch_size, shuffle=shuffle, num_workers=num_workers,
                                                      drop_last=drop_last, pin_memory=self.device.type == "cuda", **kw)
        self._side = None
        self._bufs = {}            # (slot, batch length) -> pinned batch buffers
        # the loader's own random stream (shuffle order, augmentation yaws), seeded once from the global seed: the producer
        # thread's draws do not interleave with the training thread's draws from the global generator, so a seeded run gives
        # the same batches whatever the timing
        self._gen = torch.Generator()
        self._gen.manual_seed(int(torch.initial_seed()) if seed is None else int(seed))

    def __len__(self):
        n = len(self.dataset)
        return self.passes * (n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size)

    # -- thread mode: items written in place into pinned batch buffers
    def _batches(self):
        n = len(self.dataset)
        for _ in range(self.passes):
            order = torch.randperm(n, generator=self._gen).tolist() if self.shuffle else list(range(n))
            for i in range(0, n, self.batch_size):
                idx = order[i:i + self.batch_size]
                if len(idx) < self.batch_size and self.drop_last:
                    break
                yield idx

    def _assemble(self, slot, idx):
        key = (slot, len(idx))
        bufs = self._bufs.get(key)
        if bufs is None:
            pin = self.device.type == "cuda"
            bufs = {grp: {k: torch.empty((len(idx),) + tuple(shape), dtype=dt, pin_memory=pin and len(shape) > 0)
                          for k, (shape, dt) in d.items()} for grp, d in self.dataset.native_item_spec().items()}
            self._bufs[key] = bufs
        # one interpreter thread, num_workers C threads (fill_batch); the Python worker pool of the first cut made the
        # training thread wait for the interpreter lock
        self.dataset.fill_batch(idx, bufs, threads=self.num_workers, generator=self._gen)
        return bufs

    N_SLOTS = 4          # pinned batch buffers: one being filled, up to two queued, one being copied to the device

    def _cpu_batches(self):
        """Host batches, in order.  Thread mode: ONE producer thread assembles batch i + 1, i + 2 into free pinned slots
        (``fill_batch``: the copying is done by ``num_workers`` C threads inside one library call) while the caller's thread
        enqueues the step of batch i; ``_release`` hands a slot back once the side stream has copied it to the device."""
        if self.loader is not None:
            yield from self.loader
            return
        import queue
        import sys
        q = queue.Queue(maxsize=2)
        self._free = threading.Semaphore(self.N_SLOTS)
        stop = threading.Event()
        # The producer runs interpreter code too (a few hundred microseconds per batch); with CPython's default 5 ms switch
        # interval the training thread, which needs the interpreter lock ~300 times per step, can wait up to 5 ms for it
        # whenever the producer happens to hold it -- longer than the launch queue stays fed.  0.5 ms hands the lock over
        # quickly; restored when the iteration ends.
        old_switch = sys.getswitchinterval()
        sys.setswitchinterval(min(old_switch, 5e-4))

        def produce():
            try:
                for n, idx in enumerate(self._batches()):
                    while not self._free.acquire(timeout=0.2):
                        if stop.is_set():
                            return
                    if stop.is_set():
                        return
                    q.put(self._assemble(n % self.N_SLOTS, idx))
                q.put(None)
            except BaseException as e:                # noqa: BLE001  (handed to the consumer)
                q.put(e)
        th = threading.Thread(target=produce, name="mmk-loader-producer", daemon=True)
        th.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    return
                if isinstance(item, BaseException):
                    raise item
                yield item
        finally:
            sys.setswitchinterval(old_switch)
            stop.set()
            while th.is_alive():                      # unblock a producer waiting on a full queue
                try:
                    q.get(timeout=0.05)
                except queue.Empty:
                    pass
            th.join()

    def _release(self):
        if self.loader is None and getattr(self, "_free", None) is not None:
            self._free.release()

    def _stage(self, cpu_batch):
        ds = self.dataset
        if self.device.type != "cuda":
            out = finish_batch(cpu_batch, self.device, ds.network_input_type, ds.float_type, ds.polar_res)
            if self.loader is None:            # the slot buffers are re-used: hand out copies on a CPU device
                out = {g: {k: (v.clone() if torch.is_tensor(v) else v) for k, v in d.items()} for g, d in out.items()}
                self._release()
            return out, None
        if self._side is None:
            self._side = torch.cuda.Stream(self.device)
        with torch.cuda.stream(self._side):
            out = finish_batch(cpu_batch, self.device, ds.network_input_type, ds.float_type, ds.polar_res)
            ev = torch.cuda.Event()
            ev.record(self._side)
        return out, ev

    def __iter__(self):
        it = iter(self._cpu_batches())
        try:
            staged = self._stage(next(it))
        except StopIteration:
            return
        while staged is not None:
            batch, ev = staged
            if ev is not None:
                cur = torch.cuda.current_stream(self.device)
                cur.wait_event(ev)
                for grp in batch.values():                 # memory allocated on the side stream, consumed on the caller's
                    for v in grp.values():
                        if torch.is_tensor(v) and v.is_cuda:
                            v.record_stream(cur)
            # fetch + stage the next batch BEFORE handing this one over: its copies overlap the caller's step.  A pinned slot
            # goes back to the producer once the side stream's copies out of it have completed (host-side wait on the
            # staging event: those copies were enqueued a whole step ago).
            try:
                nxt = next(it)
            except StopIteration:
                nxt = None
            if ev is not None:
                ev.synchronize()
                self._release()
            staged = self._stage(nxt) if nxt is not None else None
            yield batch
