"""Known-answer cases for the dICP boundary, independent of oracle/dicp_ref.py and of the HIP kernels: the
geometry and the expected numbers are built here from first principles (numpy fp64, scipy's matrix
exponential, finite differences) and both implementations are checked against them — the oracle in
tests/test_oracle_dicp.py (CPU), the kernels in tests/test_gpu_icp.py.  "Parity unpinned" stays (the reference
ships no dICP vectors), but the specification of DESIGN.md §3 gets anchors that do not come from our own code.

Contract being anchored: dICP.ICP.ICP(...).icp(source, target, T_init=, weight=, trim_dist=, loss_fn=, dim=)
-> {'T'} as called at /root/reference/mm_masking/icp_weight_policy.py:277-288.
"""
import numpy as np
from scipy.linalg import expm


def twist_matrix(xi):
    """4x4 se(3) element of xi = (rho, phi), translation first (pylgmath convention,
    icp_weight_dataset.py:275)."""
    r, p = xi[:3], xi[3:]
    X = np.zeros((4, 4))
    X[:3, :3] = [[0, -p[2], p[1]], [p[2], 0, -p[0]], [-p[1], p[0], 0]]
    X[:3, 3] = r
    return X


def exp_se3(xi):
    return expm(twist_matrix(np.asarray(xi, dtype=np.float64)))


def jittered_grid_pair(dim, seed=0, n_side=11, pitch=6.0, with_normals=True):
    """Noise-free copy clouds: well separated points (jittered grid, `pitch` metres apart), target = T_true
    applied to the source in fp64 and rounded to fp32.  Random unit normals (diverse directions) make
    point-to-plane observable in every direction.  -> src (N,3) f32, tgt (N,6) f32, T_true (4,4) f64."""
    rng = np.random.default_rng(seed)
    g = (np.arange(n_side) - (n_side - 1) / 2) * pitch
    X, Y = np.meshgrid(g, g)
    src = np.zeros((n_side * n_side, 3))
    src[:, 0] = X.ravel() + rng.uniform(-1.0, 1.0, src.shape[0])
    src[:, 1] = Y.ravel() + rng.uniform(-1.0, 1.0, src.shape[0])
    if dim == 3:
        src[:, 2] = rng.uniform(-2.0, 2.0, src.shape[0])
        xi = np.array([1.2, -0.8, 0.3, 0.04, -0.05, 0.15])
    else:
        xi = np.array([1.2, -0.8, 0.0, 0.0, 0.0, 0.15])
    src = src.astype(np.float32)
    T_true = exp_se3(xi)
    q = src.astype(np.float64) @ T_true[:3, :3].T + T_true[:3, 3]
    nrm = rng.normal(size=(src.shape[0], 3))
    if dim == 2:
        nrm[:, 2] = 0.0
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    tgt = np.concatenate([q, nrm], axis=1).astype(np.float32)
    perm = rng.permutation(src.shape[0])             # correspondences are not the identity permutation
    return src, np.ascontiguousarray(tgt[perm]), T_true


def pose_errors(T, T_true, dim):
    """(translation error in metres, rotation error in radians) between two poses."""
    T, T_true = np.asarray(T, np.float64), np.asarray(T_true, np.float64)
    dt = np.linalg.norm(T[:dim, 3] - T_true[:dim, 3])
    R = T[:3, :3] @ T_true[:3, :3].T
    # sin(angle) from the antisymmetric part (arccos of the trace loses half the digits near zero)
    ang = np.arcsin(np.clip(np.linalg.norm(R - R.T) / (2.0 * np.sqrt(2.0)), 0.0, 1.0))
    return dt, ang


def gauss_newton_step(src, tgt, T0, weight, icp_type, dim, loss=None, k=1.0, trim=5.0, fd_eps=1e-6):
    """One iteration computed from the definitions alone, in fp64:
      correspondences by exhaustive argmin of the squared distance (lowest index on ties),
      residuals e_i = q_i - p_i (pt2pt, first `dim` components) or n_i . (q_i - p_i) (pt2pl),
      Jacobian of p_i(delta) = Exp(delta) T0 s_i by central finite differences of scipy's expm,
      trim gate d_i < trim, robust weight (Cauchy 1 / (1 + r^2/k^2), Huber min(1, k/r)),
      delta = argmin sum_i w_i |e_i - J_i delta|^2,   T1 = Exp(delta) T0.
    delta lives in the planar subgroup (x, y, theta) for dim 2 and in se(3) for dim 3."""
    s = np.asarray(src, np.float64)
    t = np.asarray(tgt, np.float64)
    T0 = np.asarray(T0, np.float64)
    w_in = np.ones(len(s)) if weight is None else np.asarray(weight, np.float64)
    axes = [0, 1, 5] if dim == 2 else [0, 1, 2, 3, 4, 5]

    def moved(delta_small):
        xi = np.zeros(6)
        xi[axes] = delta_small
        Tn = exp_se3(xi) @ T0
        return s @ Tn[:3, :3].T + Tn[:3, 3]

    p = moved(np.zeros(len(axes)))
    d2 = ((p[:, None, :dim] - t[None, :, :dim]) ** 2).sum(-1)
    idx = d2.argmin(1)
    q = t[idx, :3]
    n = t[idx, 3:6] if t.shape[1] >= 6 else None
    rows_J, rows_e, rows_w = [], [], []
    G = np.zeros((len(s), 3, len(axes)))
    for a in range(len(axes)):
        h = np.zeros(len(axes))
        h[a] = fd_eps
        G[:, :, a] = (moved(h) - moved(-h)) / (2 * fd_eps)
    for i in range(len(s)):
        ev = (q[i] - p[i])[:dim]
        dist = np.sqrt((ev ** 2).sum())
        keep = 1.0 if dist < trim else 0.0
        if icp_type == "pt2pl":
            e = np.array([n[i, :dim] @ ev])
            J = (n[i, :dim] @ G[i, :dim, :])[None, :]
        else:
            e, J = ev, G[i, :dim, :]
        r = np.sqrt((e ** 2).sum())
        if loss == "cauchy":
            rho = 1.0 / (1.0 + (r / k) ** 2)
        elif loss == "huber":
            rho = 1.0 if r <= k else k / r
        else:
            rho = 1.0
        rows_J.append(J), rows_e.append(e), rows_w.append(np.full(len(e), w_in[i] * keep * rho))
    J, e, w = np.concatenate(rows_J), np.concatenate(rows_e), np.concatenate(rows_w)
    A = J.T @ (w[:, None] * J)
    b = J.T @ (w * e)
    delta = np.linalg.solve(A, b)
    xi = np.zeros(6)
    xi[axes] = delta
    return exp_se3(xi) @ T0, delta, idx


def two_group_lines(r2, n1=8, n2=6, r1=0.25, spacing=12.0, n0=4):
    """Point-to-plane problem with a closed-form first step.  Source points on the line y = 0 at x positions
    symmetric about 0; each has exactly one target point: groups 1 and 2 straight above (normal (0, 1)) at
    heights r1 and r2, group 0 (n0 points, normal (1, 0)) at the source point itself — zero residual, it only
    makes x observable.  `spacing` > any trim distance used, so the nearest target is the point's own twin
    and its distance is exactly the height.  With identity T_init the rows are J = (0, 1, x_i) for groups
    1 / 2 and (1, 0, 0) for group 0; sum w x_i = 0 per group decouples theta, and
        delta_y = (n1 g(r1) r1 + n2 g(r2) r2) / (n1 g(r1) + n2 g(r2)),   delta_x = delta_theta = 0,
    g = trim gate x robust weight.  -> src (N,3) f32, tgt (N,6) f32 (groups 1, 2, 0 in this order), (n1, n2, r1)."""
    assert n1 % 2 == 0 and n2 % 2 == 0 and n0 % 2 == 0
    # one lattice of +-(j + 1/2) * spacing; every +- pair belongs to one group, so every group is symmetric
    half = [(j + 0.5) * spacing for j in range((n1 + n2 + n0) // 2)]
    h1, h2, h0 = half[:n1 // 2], half[n1 // 2:(n1 + n2) // 2], half[(n1 + n2) // 2:]
    xs = np.array([v for grp in (h1, h2, h0) for h in grp for v in (-h, h)])
    n = n1 + n2 + n0
    src = np.zeros((n, 3), np.float32)
    src[:, 0] = xs
    tgt = np.zeros((n, 6), np.float32)
    tgt[:, 0] = xs
    tgt[:n1, 1] = r1
    tgt[n1:n1 + n2, 1] = r2
    tgt[:n1 + n2, 4] = 1.0
    tgt[n1 + n2:, 3] = 1.0
    return src, tgt, (n1, n2, r1)


def two_group_delta_y(r2, n1, n2, r1, loss, k, trim):
    def g(r):
        keep = 1.0 if np.float32(r) * np.float32(r) < np.float32(trim) * np.float32(trim) else 0.0
        if loss == "huber":
            rho = 1.0 if r <= k else k / r
        elif loss == "cauchy":
            rho = 1.0 / (1.0 + (r / k) ** 2)
        else:
            rho = 1.0
        return keep * rho
    r1f, r2f = float(np.float32(r1)), float(np.float32(r2))
    return (n1 * g(r1f) * r1f + n2 * g(r2f) * r2f) / (n1 * g(r1f) + n2 * g(r2f))
