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
