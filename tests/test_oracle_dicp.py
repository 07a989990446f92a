"""CPU-only tests that pin the dICP oracle (oracle/dicp_ref.py) by known-answer
geometry, invariances and finite differences — the reference holds no vectors for
this boundary ("parity unpinned")."""
import numpy as np
import torch

from mm_masking_amd import synthetic
from oracle import _clib, dicp_ref


def _pair_batch(B, n, m, dim, pad_n=0, pad_m=0, seed=0):
    S, Tg, Tt = [], [], []
    for b in range(B):
        s, t, T = synthetic.simple_cloud_pair(seed + b, n, m, dim=dim, pad_n=pad_n, pad_m=pad_m,
                                              yaw=0.02 + 0.01 * b, trans=(0.6, -0.4 + 0.1 * b, 0.1))
        S.append(s), Tg.append(t), Tt.append(T)
    return np.stack(S), np.stack(Tg), np.stack(Tt)


def test_config1_plumbing_pt2pt_5_iters():
    """BASELINE.json configs[0]: single 1024-pt 2-D cloud pair, CPU dICP point-to-point, 5 iterations."""
    rng = np.random.default_rng(0)
    N = 1024
    src = np.zeros((1, N, 3), np.float32)
    ang = rng.uniform(0, 2 * np.pi, N)
    rad = 20 + 3 * np.sin(5 * ang)                       # a closed, non-symmetric outline
    src[0, :, 0], src[0, :, 1] = rad * np.cos(ang), rad * np.sin(ang)
    T_true = synthetic.se3_exp([0.3, -0.2, 0, 0, 0, 0.03])
    tgt = np.zeros((1, N, 3), np.float32)
    tgt[0] = (src[0].astype(np.float64) @ T_true[:3, :3].T + T_true[:3, 3] + rng.normal(0, 0.01, (N, 3)) * [1, 1, 0])
    icp = dicp_ref.ICPRef("pt2pt", differentiable=False, max_iterations=5, tolerance=1e-9)
    out = icp.icp(torch.from_numpy(src), torch.from_numpy(tgt), T_init=torch.eye(4).unsqueeze(0), trim_dist=5.0,
                  loss_fn={"name": "cauchy", "metric": 1.0}, dim=2)
    T = out["T"][0].numpy()
    assert out["num_iter"] == 5 and set(out.keys()) >= {"T"}
    # point-to-point slides along the outline: after 5 steps most of the offset is gone ...
    assert np.abs(T[:2, 3] - T_true[:2, 3]).max() < 0.03 and abs(T[1, 0] - T_true[1, 0]) < 0.015
    # ... and it converges to the known transform when left running
    icp = dicp_ref.ICPRef("pt2pt", differentiable=False, max_iterations=60, tolerance=1e-7)
    T = icp.icp(torch.from_numpy(src), torch.from_numpy(tgt), T_init=torch.eye(4).unsqueeze(0), trim_dist=5.0,
                loss_fn={"name": "cauchy", "metric": 1.0}, dim=2)["T"][0].numpy()
    assert np.abs(T[:2, 3] - T_true[:2, 3]).max() < 5e-3 and abs(T[1, 0] - T_true[1, 0]) < 1e-3
    assert np.array_equal(T[2], [0, 0, 1, 0]) and np.array_equal(T[3], [0, 0, 0, 1])


def test_nn_oracle_ties_and_transform():
    t = np.array([[[0, 0], [1, 0], [1, 0], [5, 5]]], np.float32)
    p = np.array([[[0.9, 0.0], [0.5, 0.0], [4.0, 4.0]]], np.float32)
    idx, d2 = _clib.nn_search(p, t)
    assert idx.tolist() == [[1, 0, 3]]                      # exact tie at 0.5 -> lowest index
    np.testing.assert_allclose(d2[0], [0.01, 0.25, 2.0], rtol=1e-6)
    T = synthetic.se3_exp([1, 2, 3, 0.1, -0.2, 0.3]).astype(np.float32)[None]
    s = np.random.default_rng(1).normal(size=(1, 50, 3)).astype(np.float32)
    p3 = _clib.transform(s, T, 3)
    np.testing.assert_allclose(p3[0], s[0] @ T[0, :3, :3].T + T[0, :3, 3], atol=1e-6)
    pt = torch.stack(dicp_ref.transform_points(torch.from_numpy(s), torch.from_numpy(T), 3), -1).numpy()
    np.testing.assert_array_equal(pt, p3)                   # torch restatement == C restatement, bit for bit


def test_se_exp_properties():
    d3 = torch.tensor([[0.3, -0.2, 0.1, 0.05, -0.4, 0.7], [1e-7, 0, 0, 1e-6, 0, 0], [0, 0, 0, 0, 0, 0]], dtype=torch.float64)
    E = dicp_ref.se_exp(d3, 3)
    for b in range(3):
        np.testing.assert_allclose(E[b].numpy(), synthetic.se3_exp(d3[b].numpy()), atol=1e-12)
        R = E[b, :3, :3]
        np.testing.assert_allclose((R @ R.T).numpy(), np.eye(3), atol=1e-12)
    d2 = torch.tensor([[0.5, -1.0, 0.6], [0.1, 0.2, 1e-7]], dtype=torch.float64)
    E2 = dicp_ref.se_exp(d2, 2)
    for b in range(2):
        want = synthetic.se3_exp([d2[b, 0], d2[b, 1], 0, 0, 0, d2[b, 2]])
        np.testing.assert_allclose(E2[b].numpy(), want, atol=1e-12)


def test_invariances_and_padding():
    B, n, m = 2, 500, 1200
    src, tgt, T_true = _pair_batch(B, n, m, 2, seed=31)
    s, t = torch.from_numpy(src), torch.from_numpy(tgt)
    w = torch.rand(B, n) + 0.1
    icp = dicp_ref.ICPRef("pt2pl", differentiable=False, max_iterations=8, tolerance=1e-9)
    kw = dict(trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0}, dim=2)
    T1 = icp.icp(s, t, T_init=torch.eye(4).repeat(B, 1, 1), weight=w, **kw)["T"]
    T2 = icp.icp(s, t, T_init=torch.eye(4).repeat(B, 1, 1), weight=4.0 * w, **kw)["T"]
    assert (T1 - T2).abs().max().item() < 1e-5
    sp = torch.cat([s, torch.zeros(B, 50, 3)], 1)
    wp = torch.cat([w, torch.zeros(B, 50)], 1)
    tp = torch.cat([t, torch.full((B, 77, 6), icp.target_pad_val)], 1)
    T3 = icp.icp(sp, tp, T_init=torch.eye(4).repeat(B, 1, 1), weight=wp, **kw)["T"]
    assert (T1 - T3).abs().max().item() < 1e-6
    assert np.abs(T1.numpy()[:, :2, 3] - T_true[:, :2, 3]).max() < 0.05
    assert np.abs(T1.numpy()[:, 1, 0] - T_true[:, 1, 0]).max() < 5e-3
    # dim 3, point-to-plane recovers the full SE(3) offset
    src3, tgt3, T3true = _pair_batch(1, 1500, 3000, 3, seed=8)
    icp3 = dicp_ref.ICPRef("pt2pl", differentiable=False, max_iterations=15, tolerance=1e-9)
    T3d = icp3.icp(torch.from_numpy(src3), torch.from_numpy(tgt3), trim_dist=5.0, loss_fn={"name": "cauchy", "metric": 1.0},
                   dim=3)["T"].numpy()
    assert np.abs(T3d - T3true).max() < 0.05


def test_oracle_gradient_finite_difference():
    """Pins the oracle's own gradient (fp64 restatement, fixed correspondences)."""
    B, n, m = 1, 120, 400
    src, tgt, _ = _pair_batch(B, n, m, 2, seed=5)
    K = 3
    ref32 = dicp_ref.ICPRef("pt2pl", differentiable=True, max_iterations=K, tolerance=1e-12)
    base = ref32.icp(torch.from_numpy(src), torch.from_numpy(tgt), T_init=torch.eye(4).repeat(B, 1, 1),
                     weight=torch.ones(B, n), trim_dist=5.0, loss_fn={"name": "cauchy", "metric": 1.0}, dim=2)
    fixed = base["hist"]["idx"]

    def f(wv):
        o = ref32.icp(torch.from_numpy(src), torch.from_numpy(tgt), T_init=torch.eye(4).repeat(B, 1, 1), weight=wv,
                      trim_dist=5.0, loss_fn={"name": "cauchy", "metric": 1.0}, dim=2, dtype=torch.float64,
                      fixed_idx=fixed)["T"]
        return o[:, 0, 3].sum() + 2.0 * o[:, 1, 0].sum()

    w = torch.ones(B, n, dtype=torch.float64, requires_grad=True)
    f(w).backward()
    g = w.grad.clone()
    eps = 1e-6
    for i in (0, 17, 63, 119):
        wp = w.detach().clone(); wp[0, i] += eps
        wm = w.detach().clone(); wm[0, i] -= eps
        fd = (f(wp) - f(wm)).item() / (2 * eps)
        assert abs(fd - g[0, i].item()) <= 1e-5 * max(1e-3, abs(fd)) + 1e-9


# ----------------------------------------------------------------------------- known answers built outside dicp_ref
import pytest  # noqa: E402

import dicp_kat as kat  # noqa: E402


def _run_ref(icp_type, src, tgt, T0, dim, K, loss=None, trim=50.0, weight=None, tol=1e-12):
    ref = dicp_ref.ICPRef(icp_type, differentiable=False, max_iterations=K, tolerance=tol)
    lf = None if loss is None else {"name": loss, "metric": 1.0}
    w = None if weight is None else torch.from_numpy(np.asarray(weight, np.float32))[None]
    return ref.icp(torch.from_numpy(src)[None], torch.from_numpy(tgt)[None], T_init=torch.from_numpy(np.asarray(T0, np.float32))[None],
                   weight=w, trim_dist=trim, loss_fn=lf, dim=dim)


@pytest.mark.parametrize("dim", [2, 3])
@pytest.mark.parametrize("icp_type", ["pt2pt", "pt2pl"])
def test_kat_noise_free_copy_recovers_transform(icp_type, dim):
    """Noise-free copy clouds, offset inside the reference's perturbation envelope: the transform is recovered
    to north_star's gate, 1e-3 m / 1e-4 rad (in fact to fp32 rounding of the clouds)."""
    src, tgt, T_true = kat.jittered_grid_pair(dim, seed=3 + dim)
    out = _run_ref(icp_type, src, tgt, np.eye(4), dim, K=30)
    dt, da = kat.pose_errors(out["T"][0].numpy(), T_true, dim)
    assert dt <= 1e-3 and da <= 1e-4, (dt, da)
    assert dt <= 2e-5 and da <= 2e-6, (dt, da)


@pytest.mark.parametrize("icp_type,dim,loss", [("pt2pt", 2, None), ("pt2pl", 2, "huber"), ("pt2pt", 3, "cauchy"), ("pt2pl", 3, None),
                                               ("pt2pl", 2, "cauchy"), ("pt2pt", 2, "huber")])
def test_kat_one_gauss_newton_step_from_definitions(icp_type, dim, loss):
    """One iteration against a step computed from the definitions (finite-difference Jacobian of scipy's
    matrix exponential, exhaustive argmin, numpy solve): independent of dicp_ref's closed forms."""
    src, tgt, _ = kat.jittered_grid_pair(dim, seed=11)
    rng = np.random.default_rng(5)
    w = rng.uniform(0.2, 1.0, len(src)).astype(np.float32)
    T0 = kat.exp_se3([0.3, -0.2, 0.1 if dim == 3 else 0, 0.01 if dim == 3 else 0, 0, 0.03]).astype(np.float32)
    T1, delta, idx = kat.gauss_newton_step(src, tgt, T0, w, icp_type, dim, loss=loss, k=1.0, trim=5.0)
    out = _run_ref(icp_type, src, tgt, T0, dim, K=1, loss=loss, trim=5.0, weight=w)
    assert np.array_equal(out["hist"]["idx"][0][0].numpy(), idx)
    got = out["hist"]["delta"][0][0].numpy()
    np.testing.assert_allclose(got, delta, rtol=2e-5, atol=2e-6)          # fp32 per-point terms vs fp64
    np.testing.assert_allclose(out["T"][0].numpy(), T1, atol=3e-6)


@pytest.mark.parametrize("r2,loss", [(1.0 - 1e-3, "huber"), (1.0, "huber"), (1.0 + 1e-3, "huber"), (3.0, "huber"),
                                     (0.7, "cauchy"), (2.0, None)])
def test_kat_huber_kink_closed_form(r2, loss):
    """Robust weights at and around the Huber kink r = k: the first step of the two-group line problem has
    the closed form of dicp_kat.two_group_delta_y."""
    src, tgt, (n1, n2, r1) = kat.two_group_lines(r2)
    out = _run_ref("pt2pl", src, tgt, np.eye(4), 2, K=1, loss=loss, trim=5.0)
    d = out["hist"]["delta"][0][0].numpy()
    want = kat.two_group_delta_y(r2, n1, n2, r1, loss, 1.0, 5.0)
    assert abs(d[1] - want) <= 2e-7 * max(1.0, abs(want)) and abs(d[0]) < 1e-9 and abs(d[2]) < 1e-9
    if loss == "huber" and r2 > 1.0:       # the kink moved the answer (a plain least-squares step differs)
        assert abs(want - kat.two_group_delta_y(r2, n1, n2, r1, None, 1.0, 5.0)) > 1e-4


@pytest.mark.parametrize("r2,kept", [(5.0 - 1e-3, True), (5.0, False), (5.0 + 1e-3, False)])
def test_kat_trim_boundary_closed_form(r2, kept):
    """Hard trim gate d < trim_dist (5.0 at icp_weight_policy.py:279): a correspondence at exactly the trim
    distance is dropped, one just inside is kept."""
    src, tgt, (n1, n2, r1) = kat.two_group_lines(r2)
    out = _run_ref("pt2pl", src, tgt, np.eye(4), 2, K=1, loss=None, trim=5.0)
    d = out["hist"]["delta"][0][0].numpy()
    want = kat.two_group_delta_y(r2, n1, n2, r1, None, 1.0, 5.0)
    assert abs(d[1] - want) <= 2e-7 * max(1.0, abs(want))
    assert (abs(want - r1) < 1e-7) == (not kept)
