"""Synthetic scan-pair generator honouring the reference dataset's tensor contract
(mm_masking/icp_weight_dataset.py:357-362 and SURVEY.md §8a row D / §8d):

  polar radar power (A=400, R=3360) fp32 in [0,1] + azimuths quantised to the
  5600-count encoder (radar_utils.py:20-27), a lidar submap (M,6) = xyz|normal
  padded with ``target_pad_val`` rows (icp_weight_dataset.py:395-398), T_gt = I
  (gt_eye, train_icp_weights.py:366) and T_init = Exp(xi), xi = (x,y,0,0,0,yaw),
  x,y ~ U(-2,2) m, yaw ~ U(-0.6,0.6) rad (icp_weight_dataset.py:261-267).

The world is 2-D: random wall segments, a few closed polygons and a few hundred
pole-like scatterers inside a 150 m square.  The radar image holds a Gaussian
blob where each azimuth ray meets one of its first walls or a pole,
range-decaying Rayleigh speckle, and a few ghost returns that have no lidar
counterpart (what the mask network should learn to down-weight).  Pure
numpy on the host; seed = 1234 + global pair index so that every rank of a
data-parallel job draws its own pairs of the same stream.
"""
import numpy as np
import torch

POLAR_RES = 0.0596
# Scene density: "survey" = SURVEY.md §8d's workload (3 500-5 000 valid scan points of the 5 120 padded rows after
# GO-CFAR + peak extraction: measured 3 798-4 893, mean 4 270 over pairs 0..63); "sparse" = the scenes of rounds 1-2
# (2 450-2 970 valid points), kept for labelled side measurements.
DENSITY = {"survey": {"walls": (240, 281), "hits": 13, "poles": (550, 701)},
           "sparse": {"walls": (120, 181), "hits": 8, "poles": (250, 451)}}
N_AZ = 400
N_RANGE = 3360
ENCODER = 5600
BASE_SEED = 1234


def se3_exp(xi):
    """Exp of xi = (rho(3), phi(3)) -> 4x4 (translation first, as pylgmath's
    Transformation(xi_ab=...) used at icp_weight_dataset.py:275)."""
    xi = np.asarray(xi, dtype=np.float64).reshape(6)
    rho, phi = xi[:3], xi[3:]
    th = np.linalg.norm(phi)
    K = np.array([[0, -phi[2], phi[1]], [phi[2], 0, -phi[0]], [-phi[1], phi[0], 0]])
    if th < 1e-8:
        R = np.eye(3) + K
        V = np.eye(3) + 0.5 * K
    else:
        A = np.sin(th) / th
        Bc = (1 - np.cos(th)) / th ** 2
        C = (th - np.sin(th)) / th ** 3
        R = np.eye(3) + A * K + Bc * K @ K
        V = np.eye(3) + Bc * K + C * K @ K
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = V @ rho
    return T


def _walls(rng, half=75.0, count=(240, 281)):
    segs = []
    for _ in range(rng.integers(*count)):
        c = rng.uniform(-half, half, 2)
        ang = rng.uniform(0, np.pi)
        L = rng.uniform(4.0, 30.0)
        d = np.array([np.cos(ang), np.sin(ang)]) * L / 2
        segs.append(np.concatenate([c - d, c + d]))
    for _ in range(rng.integers(2, 5)):
        c = rng.uniform(-half * 0.8, half * 0.8, 2)
        k = rng.integers(3, 7)
        rad = rng.uniform(4.0, 12.0)
        a0 = rng.uniform(0, 2 * np.pi)
        pts = np.stack([c + rad * np.array([np.cos(a0 + 2 * np.pi * i / k), np.sin(a0 + 2 * np.pi * i / k)])
                        for i in range(k)])
        for i in range(k):
            segs.append(np.concatenate([pts[i], pts[(i + 1) % k]]))
    segs = np.clip(np.array(segs), -half, half)
    # keep walls away from the sensor at the origin
    mid = 0.5 * (segs[:, :2] + segs[:, 2:])
    return segs[np.linalg.norm(mid, axis=1) > 6.0]


def _poles(rng, half=75.0, count=(550, 701)):
    """Compact scatterers (poles, trunks): (K,2) positions, at least 6 m from the sensor."""
    k = rng.integers(*count)
    r = rng.uniform(6.0, half, k)
    a = rng.uniform(0, 2 * np.pi, k)
    return np.stack([r * np.cos(a), r * np.sin(a)], axis=1)


def _lidar(rng, segs, poles, m_valid, m_pad, pad_val):
    lens = np.linalg.norm(segs[:, 2:] - segs[:, :2], axis=1)
    m_pole = min(m_valid // 8, 12 * len(poles))
    m_wall = m_valid - m_pole
    which = rng.choice(len(segs), size=m_wall, p=lens / lens.sum())
    s = rng.uniform(0, 1, m_wall)
    p0, p1 = segs[which, :2], segs[which, 2:]
    xy = p0 + s[:, None] * (p1 - p0) + rng.normal(0, 0.02, (m_wall, 2))
    t = (p1 - p0) / lens[which, None]
    n = np.stack([-t[:, 1], t[:, 0]], axis=1)
    # normals face the sensor, as estimated lidar normals do
    flip = np.sum(n * xy, axis=1) > 0
    n[flip] *= -1
    pw = rng.integers(0, len(poles), m_pole)
    pxy = poles[pw] + rng.normal(0, 0.06, (m_pole, 2))
    pn = -pxy / np.linalg.norm(pxy, axis=1, keepdims=True)
    xy = np.concatenate([xy, pxy])
    n = np.concatenate([n, pn])
    perm = rng.permutation(m_valid)
    xy, n = xy[perm], n[perm]
    pc = np.full((m_pad, 6), pad_val, dtype=np.float32)
    pc[:m_valid, 0:2] = xy
    pc[:m_valid, 2] = 0.0
    pc[:m_valid, 3:5] = n
    pc[:m_valid, 5] = 0.0
    return pc


def _radar(rng, segs, poles, wobble=True, hits=13):
    counts = np.round(np.arange(N_AZ) * (ENCODER / N_AZ)).astype(np.int64)
    if wobble:
        counts = np.sort(np.clip(counts + rng.integers(-1, 2, N_AZ), 0, ENCODER - 1))
        az = (counts * (2 * np.pi / ENCODER)).astype(np.float32)
    d = np.stack([np.cos(az), np.sin(az)], axis=1).astype(np.float64)          # (A,2)
    p0, e = segs[:, :2], segs[:, 2:] - segs[:, :2]                               # (S,2)
    # ray o + t d hits p0 + s e :  t = cross(p0, e) / cross(d, e), s = cross(p0, d) / cross(d, e)
    den = d[:, None, 0] * e[None, :, 1] - d[:, None, 1] * e[None, :, 0]
    den = np.where(np.abs(den) < 1e-9, np.nan, den)
    t = (p0[None, :, 0] * e[None, :, 1] - p0[None, :, 1] * e[None, :, 0]) / den
    s = (p0[None, :, 0] * d[:, None, 1] - p0[None, :, 1] * d[:, None, 0]) / den
    hit = (t > 3.0) & (s >= 0) & (s <= 1) & (t < (N_RANGE - 40) * POLAR_RES)
    t = np.where(hit, t, np.inf)
    t.sort(axis=1)
    cols = np.arange(N_RANGE, dtype=np.float32)
    decay = np.exp(-cols / 2500.0).astype(np.float32)
    img = (rng.rayleigh(0.04, size=(N_AZ, N_RANGE)).astype(np.float32)) * decay[None, :]

    def blob(a, rng_m, amp, sigma=2.0):
        c = rng_m / POLAR_RES
        lo, hi = int(max(0, c - 8)), int(min(N_RANGE, c + 9))
        img[a, lo:hi] += amp * np.exp(-0.5 * ((cols[lo:hi] - c) / sigma) ** 2)

    for a in range(N_AZ):
        amp = 1.0
        for h in range(min(hits, t.shape[1])):    # the beam partially penetrates: the first `hits` walls
            if not np.isfinite(t[a, h]):
                break
            blob(a, t[a, h], rng.uniform(0.4, 0.9) * amp)
            amp *= 0.9
    pr = np.linalg.norm(poles, axis=1)
    pa = np.mod(np.arctan2(poles[:, 1], poles[:, 0]), 2 * np.pi)
    pk = np.searchsorted(az, pa.astype(np.float32)) % N_AZ
    for k in range(len(poles)):                # a pole lights up one or two neighbouring azimuths
        if pr[k] < (N_RANGE - 40) * POLAR_RES:
            blob(int(pk[k]), pr[k], rng.uniform(0.4, 0.9))
            if rng.uniform() < 0.5:
                blob(int((pk[k] + 1) % N_AZ), pr[k], rng.uniform(0.3, 0.7))
    for _ in range(rng.integers(3, 9)):        # ghosts / multipath: no lidar counterpart
        a0 = rng.integers(0, N_AZ)
        r = rng.uniform(8.0, 70.0)
        for da in range(rng.integers(3, 7)):
            blob((a0 + da) % N_AZ, r + rng.normal(0, 0.05), rng.uniform(0.5, 0.9))
    times = (np.arange(N_AZ, dtype=np.float32) * 625.0).astype(np.float32)
    return np.clip(img, 0.0, 1.0).astype(np.float32), az, times


def make_pair(index, m_valid=20000, m_pad=20480, pad_val=1000.0, dataset_type="train", pos_std=2.0, rot_std=0.6,
              wobble=True, density="survey", dim=2):
    """One synthetic scan pair (numpy).  ``dim=3`` (the SE(3) / 6x6 variant of SURVEY.md §8d config 3): the map points get
    a height z ~ U(-1.5, 1.5) m and 40 % of the normals are tilted out of the plane -- drawn from a stream of their own,
    so that everything else equals the dim-2 pair of the same index -- which makes z, roll and pitch observable for the
    planar radar scan (z = 0) under the point-to-plane residual."""
    rng = np.random.default_rng(BASE_SEED + int(index))
    dens = DENSITY[density]
    segs = _walls(rng, count=dens["walls"])
    poles = _poles(rng, count=dens["poles"])
    map_pc = _lidar(rng, segs, poles, m_valid, m_pad, pad_val)
    fft, az, times = _radar(rng, segs, poles, wobble=wobble, hits=dens["hits"])
    if dim == 3:
        r3 = np.random.default_rng([BASE_SEED + int(index), 3])
        map_pc[:m_valid, 2] = r3.uniform(-1.5, 1.5, m_valid)
        tilt = r3.uniform(-0.8, 0.8, m_valid) * (r3.uniform(0, 1, m_valid) < 0.4)
        map_pc[:m_valid, 3:5] *= np.cos(tilt)[:, None].astype(np.float32)
        map_pc[:m_valid, 5] = np.sin(tilt)
    if dataset_type == "train":
        xi = np.zeros(6)
        xi[0:2] = pos_std * rng.uniform(-1, 1, 2)
        xi[5] = rot_std * rng.uniform(-1, 1)
    else:
        xi = np.array([rng.normal(0, pos_std), rng.normal(0, pos_std), 0, 0, 0, rng.normal(0, rot_std)])
    return {"fft_polar": fft, "azimuths": az, "az_times": times, "map_pc": map_pc,
            "T_init": se3_exp(xi).astype(np.float32), "T_gt": np.eye(4, dtype=np.float32)}


def make_batch(indices, device="cpu", **kw):
    """Stack pairs into the batch layout of the reference DataLoader."""
    items = [make_pair(i, **kw) for i in indices]
    out = {k: torch.from_numpy(np.stack([it[k] for it in items])) for k in items[0]}
    return {k: v.to(device) for k, v in out.items()}


def simple_cloud_pair(seed, n, m, dim=2, noise=0.01, extent=40.0, pad_n=0, pad_m=0, pad_val=1000.0, with_normals=True,
                      yaw=0.2, trans=(0.8, -0.5, 0.0)):
    """Small structured cloud pair for parity tests (config 1 style): target = walls,
    source = a sub-sample moved by the inverse of a known transform."""
    rng = np.random.default_rng(seed)
    nseg = 12
    c = rng.uniform(-extent, extent, (nseg, 2))
    ang = rng.uniform(0, np.pi, nseg)
    L = rng.uniform(10, 30, nseg)
    dvec = np.stack([np.cos(ang), np.sin(ang)], 1)
    which = rng.integers(0, nseg, m)
    s = rng.uniform(-0.5, 0.5, m)
    xy = c[which] + (s * L[which])[:, None] * dvec[which]
    nrm = np.stack([-dvec[which, 1], dvec[which, 0]], 1)
    z = rng.uniform(-1.5, 1.5, m) if dim == 3 else np.zeros(m)
    tgt = np.zeros((m + pad_m, 6), dtype=np.float32)
    tgt[:m, 0:2] = xy
    tgt[:m, 2] = z
    tgt[:m, 3:5] = nrm
    if dim == 3:
        # tilt a third of the normals out of plane so that z and roll/pitch are observable
        tilt = rng.uniform(-0.8, 0.8, m) * (rng.uniform(0, 1, m) < 0.4)
        nz = np.sin(tilt)
        tgt[:m, 3:5] *= np.cos(tilt)[:, None]
        tgt[:m, 5] = nz
    tgt[m:, :] = pad_val
    sel = rng.choice(m, size=n, replace=(n > m))
    pts = tgt[sel, :3].astype(np.float64) + rng.normal(0, noise, (n, 3)) * np.array([1, 1, 1 if dim == 3 else 0])
    xi = np.array([trans[0], trans[1], trans[2] if dim == 3 else 0.0, 0.0, 0.0, yaw])
    if dim == 3:
        xi[3:5] = [0.03, -0.02]
    T_true = se3_exp(xi)                        # maps source -> target
    Tinv = np.linalg.inv(T_true)
    src = np.zeros((n + pad_n, 3), dtype=np.float32)
    src[:n] = (pts @ Tinv[:3, :3].T + Tinv[:3, 3]).astype(np.float32)
    if not with_normals:
        tgt = tgt[:, :3].copy()
    return src, tgt, T_true.astype(np.float32)
