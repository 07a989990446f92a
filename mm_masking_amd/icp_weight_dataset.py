"""Tensor-side pieces of the reference's ``ICPWeightDataset`` (mm_masking/icp_weight_dataset.py)
that define what enters the hot path: map filtering, padding to
batchable shapes, the random initial guess, and the rotation augmentation.  The file I/O around
them (Boreas PNGs, vtr3 pose graphs, pyboreas, ROS2 bags) is out of scope; these functions take
and return plain tensors so that a loader for real data, or the synthetic generator, can share
them.  They are per-item host logic (the reference runs them in DataLoader workers) and use
ordinary tensor ops on whatever device the tensors live on.
"""
import math

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
