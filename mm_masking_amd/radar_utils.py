"""Radar operators of the hot path — host-side mirror of the reference module
``mm_masking/radar_utils.py`` (same function names, positional order and
argument meaning) with the arithmetic running in the hand-written gfx950
kernels of libmmk_hip.so (csrc/mmk_radar.hip) through the C ABI.

There is no CPU fallback: every kernel-backed function needs a HIP device and
the built library, and raises otherwise.  Tensors handed over on the CPU (the
reference runs these functions inside DataLoader workers,
icp_weight_dataset.py:336-352) are moved to the current HIP device and the
result is returned on the caller's device.

Reference line numbers are given per function (radar_utils.py:<lines>).
"""
import ctypes

import numpy as np
import torch

from . import _lib

__all__ = ["load_pc_from_file", "load_radar", "cfar_mask", "extract_pc", "extract_pc_padded", "extract_weights",
           "extract_bev_from_pts", "mean_peaks_parallel_fast", "pol_2_cart", "radar_polar_to_cartesian",
           "radar_polar_to_cartesian_diff", "radar_cartesian_to_polar", "point_to_cart_idx",
           "form_cart_range_angle_grid", "form_polar_range_grid"]


def _hip_device(t):
    if t.is_cuda:
        return t.device
    if not torch.cuda.is_available():
        raise _lib.MmkError("radar_utils needs an MI355X/HIP device: the operators are HIP kernels with no CPU path")
    return torch.device("cuda", torch.cuda.current_device())


def _back(out, like):
    return out if like.is_cuda else out.to(like.device)


_ws_cache = {}


def _workspace(nbytes, device):
    key = (device.index, torch.cuda.current_stream(device).cuda_stream, "radar")
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


# ----------------------------------------------------------------------------- host-only helpers
def load_pc_from_file(file_path, to_type=None, to_device="cpu"):
    """radar_utils.py:10-18: float32 x 6 per point."""
    pc = np.fromfile(file_path, dtype=np.float32)
    pc = torch.from_numpy(pc.reshape((len(pc) // 6, 6))).to(to_device)
    return pc if to_type is None else pc.type(to_type)


def load_radar(raw_img):
    """radar_utils.py:20-27: Navtech PNG rows -> (fft f32 (A,R), azimuths f64 (A,), timestamps i64 (A,))."""
    raw = np.asarray(raw_img)
    timestamps = np.frombuffer(raw[:, :8].tobytes(), dtype=np.int64) * 1000
    azimuths = np.frombuffer(raw[:, 8:10].tobytes(), dtype=np.uint16) * (2 * np.pi / 5600)
    fft_data = np.divide(raw[:, 11:], 255.0, dtype=np.float32)
    return fft_data, azimuths, timestamps


def form_cart_range_angle_grid(cart_resolution=0.2384, cart_pixel_width=640, dtype=None, device="cpu"):
    """radar_utils.py:399-419.  Evaluated with the same PyTorch CPU ops as the
    reference (so the constant grid is bit-identical to the reference's on the
    same host), then moved to ``device``."""
    if (cart_pixel_width % 2) == 0:
        cart_min_range = (cart_pixel_width / 2 - 0.5) * cart_resolution
    else:
        cart_min_range = cart_pixel_width / 2 * cart_resolution
    kw = {} if dtype is None else {"dtype": dtype}
    coords = torch.linspace(-cart_min_range, cart_min_range, cart_pixel_width, **kw)
    Y, X = torch.meshgrid(coords, -1 * coords, indexing="xy")
    sample_range = torch.sqrt(Y * Y + X * X)
    sample_angle = torch.arctan2(Y, X)
    sample_angle = sample_angle + torch.where(sample_angle < 0, 2.0 * torch.pi, 0.0)
    return sample_range.to(device), sample_angle.to(device)


def form_polar_range_grid(polar_resolution=0.2384, polar_pixel_shape=(400, 3360), dtype=None, device="cpu"):
    """radar_utils.py:421-438."""
    polar_range = (polar_pixel_shape[1] - 1) * polar_resolution
    kw = {} if dtype is None else {"dtype": dtype}
    range_coords = torch.linspace(0.0, polar_range, polar_pixel_shape[1], **kw).to(device)
    return range_coords.unsqueeze(0).expand(polar_pixel_shape[0], -1)


_grid_cache = {}


def _device_grids(width, device):
    key = (width, device.type, device.index)
    g = _grid_cache.get(key)
    if g is None:
        # the reference ignores cart_resolution when it builds the grid (radar_utils.py:276)
        r, a = form_cart_range_angle_grid(cart_pixel_width=width, dtype=torch.float32)
        g = (r.contiguous().to(device), a.contiguous().to(device))
        _grid_cache[key] = g
    return g


def point_to_cart_idx(pc, cart_resolution=0.2384, cart_pixel_width=640, min_to_plus_1=False):
    """radar_utils.py:374-397 (tiny elementwise host logic; the kernels fuse it)."""
    grid_pc_u = -pc[:, :, 0] / cart_resolution
    grid_pc_v = pc[:, :, 1] / cart_resolution
    if min_to_plus_1:
        grid_pc = torch.stack((grid_pc_v, grid_pc_u), axis=2)
        return grid_pc / (cart_pixel_width - 1) * 2
    grid_pc = torch.stack((grid_pc_u, grid_pc_v), axis=2)
    return grid_pc + cart_pixel_width / 2


def mean_peaks_parallel_fast(arr, diff, steep_fact):
    """radar_utils.py:167-185.  Stand-alone elementwise form kept for API parity;
    ``extract_pc`` does not call it (the marker rule is fused into the HIP
    extraction kernels)."""
    res = torch.zeros_like(arr)
    zero_detect = (1 - torch.tanh(steep_fact * arr)) if diff else (arr == 0)
    res[:, :, :-1] = arr[:, :, :-1] * zero_detect[:, :, 1:] + arr[:, :, 1:] * zero_detect[:, :, :-1]
    return res


def pol_2_cart(pointcloud):
    """radar_utils.py:187-195."""
    rho, phi = pointcloud[:, 0], pointcloud[:, 1]
    return torch.stack((rho * torch.cos(phi), rho * torch.sin(phi), torch.zeros_like(rho)), axis=1)


def radar_polar_to_cartesian(*args, **kwargs):
    """radar_utils.py:197-256 (cv2.remap based).  Never called on the reference's
    train path: outside the hot-path scope."""
    raise NotImplementedError("radar_polar_to_cartesian (cv2) is dead code upstream and out of scope; "
                              "use radar_polar_to_cartesian_diff")


def radar_cartesian_to_polar(cart, azimuths, radar_resolution, cart_resolution=0.2384, polar_pixel_shape=(400, 3360)):
    """radar_utils.py:338-372.  (B,H,W) fp64 + (B,A) -> (B,A,R) fp64, bit-identical to the reference.
    As upstream, only an fp64 image is accepted: the reference casts its sampling grid to double (:370)
    and ``F.grid_sample`` raises ``RuntimeError`` on the dtype mismatch for anything else — same error here.
    sin / cos of the azimuths and the range coordinates are formed on the host with the reference's own
    torch CPU calls (B*A + R numbers); products, divisions and the bilinear gather run in the HIP kernel."""
    if cart.dtype != torch.float64:
        raise RuntimeError("expected scalar type Float but found Double" if cart.dtype == torch.float32 else
                           "expected scalar type %s but found Double" % str(cart.dtype).replace("torch.", "").capitalize())
    dev = _hip_device(cart)
    x = cart.detach().to(dev).contiguous()
    B, H, W = x.shape
    A, R = int(polar_pixel_shape[0]), int(polar_pixel_shape[1])
    if azimuths.shape != (B, A):
        raise ValueError("azimuths must be (B, %d) (got %s)" % (A, tuple(azimuths.shape)))
    az = azimuths.detach().to(device="cpu", dtype=torch.float64)
    rc = form_polar_range_grid(polar_resolution=radar_resolution, polar_pixel_shape=polar_pixel_shape, dtype=torch.float64,
                               device="cpu")[0]
    s_az, c_az = torch.sin(az).to(dev).contiguous(), torch.cos(az).to(dev).contiguous()
    rc = rc.contiguous().to(dev)
    out = torch.empty(B, A, R, dtype=torch.float64, device=dev)
    _lib.check(_lib.lib().mmk_cart_to_polar(_lib.ptr(x), _lib.ptr(s_az), _lib.ptr(c_az), _lib.ptr(rc), B, A, R, H, W,
                                            float(cart_resolution), _lib.ptr(out), _lib.stream_ptr(dev)))
    return _back(out, cart)


# ----------------------------------------------------------------------------- R2
def cfar_cols(n_range, res, width=101, minr=2.0, maxr=80.0, guard=5):
    """Window half width and column range of radar_utils.py:34-39."""
    width = width + 1 if width % 2 == 0 else width
    w2 = width // 2
    mincol = max(0, int(minr / res + w2 + guard + 1))
    maxcol = min(n_range, int(maxr / res - w2 - guard))
    return w2, mincol, maxcol


def cfar_mask(raw_scans, res, width=101, minr=2.0, maxr=80.0, guard=5,
              a_thresh=1.0, b_thresh=0.09, diff=True, steep_fact=10.0):
    """GO-CFAR mask, radar_utils.py:29-69.  (B,A,R) fp32 -> (B,A,R) fp32."""
    assert raw_scans.ndim == 3, "raw_scans must be 3D"
    dev = _hip_device(raw_scans)
    x = _lib.dev_f32(raw_scans, dev)
    B, A, R = x.shape
    w2, mincol, maxcol = cfar_cols(R, res, width, minr, maxr, guard)
    out = torch.empty_like(x)
    _lib.check(_lib.lib().mmk_cfar_mask(_lib.ptr(x), B, A, R, w2, guard, mincol, max(mincol, maxcol),
                                        float(a_thresh), float(b_thresh), 1 if diff else 0, float(steep_fact),
                                        _lib.ptr(out), _lib.stream_ptr(dev)))
    return _back(out, raw_scans)


# ----------------------------------------------------------------------------- R3 + R4
def extract_pc_padded(thres_mask, res, azimuth_angles, azimuth_times, max_pts, T_ab=None, diff=True,
                      steep_fact=10.0):
    """Batched form of ``extract_pc``: zero-padded (B,max_pts,3) cloud in the
    reference's azimuth-major order plus the per-item point count (int32 (B,)),
    with no host synchronisation.  This is the layout the dataset hands to the
    policy (icp_weight_dataset.py:379-381)."""
    dev = _hip_device(thres_mask)
    m = _lib.dev_f32(thres_mask, dev)
    az = _lib.dev_f32(azimuth_angles, dev)
    tm = _lib.dev_f32(azimuth_times, dev) if azimuth_times is not None else None
    Tab = _lib.dev_f32(T_ab, dev).reshape(-1, 16) if T_ab is not None else None
    B, A, R = m.shape
    L = _lib.lib()
    nbytes = L.mmk_extract_peaks_workspace_bytes(B, A, R, int(max_pts))
    ws = _workspace(nbytes, dev)
    pc = torch.empty(B, int(max_pts), 3, dtype=torch.float32, device=dev)
    cnt = torch.empty(B, dtype=torch.int32, device=dev)
    _lib.check(L.mmk_extract_peaks(_lib.ptr(m), B, A, R, float(res), _lib.ptr(az), _lib.ptr(tm), _lib.ptr(Tab),
                                   1 if diff else 0, float(steep_fact), int(max_pts), _lib.ptr(pc), _lib.ptr(cnt),
                                   _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev)))
    return pc, cnt


def extract_pc(thres_mask, res, azimuth_angles, azimuth_times, T_ab=None, diff=True, steep_fact=10.0):
    """radar_utils.py:71-106: Python list of ragged (n_i,3) clouds (one host sync
    to read the counts, as the reference's ``nonzero`` implies)."""
    B, A, R = thres_mask.shape
    cap = (A * (R - 1) + 1) // 2
    cap = min(cap, 1 << 20)
    pc, cnt = extract_pc_padded(thres_mask, res, azimuth_angles, azimuth_times, cap, T_ab=T_ab, diff=diff,
                                steep_fact=steep_fact)
    counts = cnt.cpu().tolist()
    return [_back(pc[b, :min(n, cap)].clone(), thres_mask) for b, n in enumerate(counts)]


# ----------------------------------------------------------------------------- R5
def radar_polar_to_cartesian_diff(fft_data, azimuths, radar_resolution, cart_resolution=0.2384, cart_pixel_width=640,
                                  interpolate_crossover=True, fix_wobble=True):
    """radar_utils.py:258-336.  (B,A,R) + (B,A) -> (B,W,W).  As upstream, the
    pixel grid is built with the default 0.2384 m resolution whatever
    ``cart_resolution`` says (radar_utils.py:276)."""
    dev = _hip_device(fft_data)
    x = _lib.dev_f32(fft_data, dev)
    az = _lib.dev_f32(azimuths, dev)
    B, A, R = x.shape
    W = int(cart_pixel_width)
    rg, ag = _device_grids(W, dev)
    out = torch.empty(B, W, W, dtype=torch.float32, device=dev)
    _lib.check(_lib.lib().mmk_polar_to_cart(_lib.ptr(x), _lib.ptr(az), _lib.ptr(rg), _lib.ptr(ag), B, A, R, W,
                                            float(radar_resolution), 1 if interpolate_crossover else 0,
                                            1 if fix_wobble else 0, _lib.ptr(out), _lib.stream_ptr(dev)))
    return _back(out, fft_data)


def _polar_to_cart_pair(img_a, img_b, azimuths, radar_resolution, cart_pixel_width=640):
    """radar_polar_to_cartesian_diff of two images that share their azimuths (the FFT and the CFAR
    image of a scan, icp_weight_dataset.py:350-352) in one launch: same values as two calls."""
    dev = _hip_device(img_a)
    a, b = _lib.dev_f32(img_a, dev), _lib.dev_f32(img_b, dev)
    az = _lib.dev_f32(azimuths, dev)
    assert a.shape == b.shape
    B, A, R = a.shape
    W = int(cart_pixel_width)
    rg, ag = _device_grids(W, dev)
    out_a = torch.empty(B, W, W, dtype=torch.float32, device=dev)
    out_b = torch.empty(B, W, W, dtype=torch.float32, device=dev)
    _lib.check(_lib.lib().mmk_polar_to_cart_pair(_lib.ptr(a), _lib.ptr(b), _lib.ptr(az), _lib.ptr(rg), _lib.ptr(ag), B, A, R, W,
                                                 float(radar_resolution), 1, 1, _lib.ptr(out_a), _lib.ptr(out_b),
                                                 _lib.stream_ptr(dev)))
    return _back(out_a, img_a), _back(out_b, img_b)


# ----------------------------------------------------------------------------- R9
class _SampleWeights(torch.autograd.Function):
    """Bilinear gather of the mask at the scan points; backward is the
    scatter-add into the 4 taps (what autograd does for F.grid_sample at
    radar_utils.py:126)."""

    @staticmethod
    def forward(ctx, mask, pc, cart_resolution, cart_pixel_width=640):
        B, H, W = mask.shape
        N, cols = pc.shape[1], pc.shape[2]
        out = torch.empty(B, N, dtype=torch.float32, device=mask.device)
        _lib.check(_lib.lib().mmk_sample_weights_fwd(_lib.ptr(mask, torch.float32, "mask"), _lib.ptr(pc), B, N, cols,
                                                     H, W, int(cart_pixel_width), float(cart_resolution), _lib.ptr(out),
                                                     _lib.stream_ptr(mask.device)))
        ctx.save_for_backward(pc)
        ctx.shape = (B, H, W)
        ctx.cres = float(cart_resolution)
        ctx.cw = int(cart_pixel_width)
        return out

    @staticmethod
    def backward(ctx, gw):
        (pc,) = ctx.saved_tensors
        B, H, W = ctx.shape
        gw = gw.contiguous().float()
        gmask = torch.empty(B, H, W, dtype=torch.float32, device=gw.device)
        nb = int(_lib.lib().mmk_sample_weights_bwd_ws_bytes(B, pc.shape[1]))
        ws = _workspace(nb, gw.device)
        _lib.check(_lib.lib().mmk_sample_weights_bwd(_lib.ptr(gw), _lib.ptr(pc), B, pc.shape[1], pc.shape[2], H, W,
                                                     ctx.cw, ctx.cres, _lib.ptr(gmask), _lib.ptr(ws), nb, _lib.stream_ptr(gw.device)))
        return gmask, None, None, None


class _WeightStats(torch.autograd.Function):
    """The statistics of extract_weights (radar_utils.py:130-138) in one pass (mmk_weight_stats).
    Only diff_mean_num_non0 carries a gradient (d/dw of sum(0.5 tanh(5w) + 0.5) / B over real points)."""

    @staticmethod
    def forward(ctx, weights, pc):
        B, N = weights.shape
        part = torch.empty(B * 8, dtype=torch.float32, device=weights.device)
        out = torch.empty(8, dtype=torch.float32, device=weights.device)
        w = weights.contiguous()
        _lib.check(_lib.lib().mmk_weight_stats(_lib.ptr(w), _lib.ptr(pc), B, N, pc.shape[2], _lib.ptr(part), _lib.ptr(out),
                                               _lib.stream_ptr(weights.device)))
        ctx.save_for_backward(w, pc)
        diff = out[0].clone()
        ctx.mark_non_differentiable(out)
        return diff, out

    @staticmethod
    def backward(ctx, g_diff, _g_stats):
        if g_diff is None:
            return None, None
        w, pc = ctx.saved_tensors
        real = ~((pc[:, :, 0] == 0.0) & (pc[:, :, 1] == 0.0))
        t = torch.tanh(5 * w)
        return g_diff * (2.5 / w.shape[0]) * (1 - t * t) * real, None


def _extract_weights_stats(mask, scan_pc):
    """extract_weights plus the raw statistics vector of mmk_weight_stats (the policy takes its
    mean_all_pts from it)."""
    dev = _hip_device(mask)
    m = mask if (mask.is_cuda and mask.dtype == torch.float32 and mask.is_contiguous()) else \
        mask.to(device=dev, dtype=torch.float32).contiguous()
    pc = _lib.dev_f32(scan_pc, dev)
    # point_to_cart_idx's defaults (0.2384 m, 640 px) whatever the mask's shape: radar_utils.py:112
    weights = _SampleWeights.apply(m, pc, 0.2384, 640)
    diff_mean_num_non0, st = _WeightStats.apply(weights, pc)
    if not mask.is_cuda:
        weights = weights.to(mask.device)
    return (weights, diff_mean_num_non0, st[1], st[2], st[3], st[4]), st


def extract_weights(mask, scan_pc):
    """radar_utils.py:108-140 -> (weights (B,N), diff_mean_num_non0, mean_num_non0,
    mean_w, max_w, min_w).  The statistics are the reference's, formed over the real points in one
    fused pass (no boolean indexing, no host sync)."""
    return _extract_weights_stats(mask, scan_pc)[0]


# ----------------------------------------------------------------------------- R10
def extract_bev_from_pts(pc, cart_pixel_width=640):
    """radar_utils.py:142-165.  (B,M,>=2) -> (B,W,W) binary image."""
    dev = _hip_device(pc)
    p = _lib.dev_f32(pc, dev)
    B, M, cols = p.shape
    W = int(cart_pixel_width)
    bev = torch.empty(B, W, W, dtype=torch.float32, device=dev)
    _lib.check(_lib.lib().mmk_bev_raster(_lib.ptr(p), B, M, cols, W, 0.2384, _lib.ptr(bev), _lib.stream_ptr(dev)))
    return _back(bev.to(pc.dtype) if pc.dtype.is_floating_point else bev, pc)
