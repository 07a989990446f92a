"""Parity of the HIP differentiable ICP (through the C ABI) with the oracle.
Bit-exact for correspondence indices; pose within 1e-3 m / 1e-4 rad (BASELINE.json
north_star; in practice ~1e-6); gradients within 2e-3 relative of autograd through
the CPU restatement."""
import ctypes

import numpy as np
import pytest
import torch

from mm_masking_amd import _lib, synthetic
from mm_masking_amd.dICP.ICP import ICP
from oracle import _clib, dicp_ref

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


@pytest.fixture(params=["brute", "grid"], autouse=True)
def nn_engine(request):
    """Every ICP test runs with both nearest-neighbour engines: the exhaustive scan and the exact
    uniform-grid search must produce identical correspondences."""
    ICP.NN_SEARCH_OVERRIDE = request.param
    yield request.param
    ICP.NN_SEARCH_OVERRIDE = None


def _nn_gpu(src, tgt, T, dim):
    L = _lib.lib()
    B, N, _ = src.shape
    M = tgt.shape[1]
    s = torch.from_numpy(src).to(DEV)
    t = torch.from_numpy(tgt).to(DEV)
    Tt = torch.from_numpy(T.reshape(B, 16)).to(DEV)
    Mpad = L.mmk_nn_padded_m(M)
    planar = torch.empty(B, dim, Mpad, dtype=torch.float32, device=DEV)
    _lib.check(L.mmk_pack_target(_lib.ptr(t), B, M, t.shape[2], dim, _lib.ptr(planar), _lib.stream_ptr(DEV)))
    ws = torch.empty(L.mmk_nn_workspace_bytes(B, N, M, dim), dtype=torch.uint8, device=DEV)
    idx = torch.empty(B, N, dtype=torch.int32, device=DEV)
    d2 = torch.empty(B, N, dtype=torch.float32, device=DEV)
    _lib.check(L.mmk_nn_search(_lib.ptr(s), _lib.ptr(planar), _lib.ptr(Tt), B, N, M, dim, _lib.ptr(idx), _lib.ptr(d2),
                               _lib.ptr(ws), ws.numel(), _lib.stream_ptr(DEV)))
    torch.cuda.synchronize()
    return idx.cpu().numpy(), d2.cpu().numpy()


def _rand_T(rng, B, dim):
    T = np.tile(np.eye(4, dtype=np.float32), (B, 1, 1))
    for b in range(B):
        xi = np.zeros(6)
        xi[:2] = rng.uniform(-2, 2, 2)
        xi[5] = rng.uniform(-0.6, 0.6)
        if dim == 3:
            xi[2] = rng.uniform(-0.5, 0.5)
            xi[3:5] = rng.uniform(-0.1, 0.1, 2)
        T[b] = synthetic.se3_exp(xi).astype(np.float32)
    return T


@pytest.mark.parametrize("dim", [2, 3])
@pytest.mark.parametrize("B,N,M", [(1, 1, 1), (2, 513, 1025), (3, 1000, 3000), (1, 1024, 1024), (9, 300, 2049)])
def test_nn_bit_exact(dim, B, N, M):
    rng = np.random.default_rng(B * 1000 + N + M + dim)
    src = rng.uniform(-60, 60, (B, N, 3)).astype(np.float32)
    tgt = rng.uniform(-60, 60, (B, M, 6)).astype(np.float32)
    # exact duplicates in the target: ties must resolve to the lowest index
    if M > 10:
        tgt[:, M // 2] = tgt[:, 3]
        tgt[:, M - 1] = tgt[:, 7]
        src[:, 0] = tgt[:, 3, :3]
    T = _rand_T(rng, B, dim)
    idx, d2 = _nn_gpu(src, tgt, T, dim)
    p = _clib.transform(src, T, dim)
    idx_ref, d2_ref = _clib.nn_search(p, np.ascontiguousarray(tgt[:, :, :dim]))
    np.testing.assert_array_equal(idx, idx_ref)
    np.testing.assert_array_equal(d2, d2_ref)


def test_nn_padded_targets_and_sources():
    rng = np.random.default_rng(5)
    B, N, M = 2, 700, 1500
    src = np.zeros((B, N, 3), np.float32)
    src[:, :500] = rng.uniform(-50, 50, (B, 500, 3))
    tgt = np.full((B, M, 6), 1000.0, np.float32)
    tgt[:, :1200] = rng.uniform(-50, 50, (B, 1200, 6))
    T = _rand_T(rng, B, 2)
    idx, d2 = _nn_gpu(src, tgt, T, 2)
    idx_ref, d2_ref = _clib.nn_search(_clib.transform(src, T, 2), np.ascontiguousarray(tgt[:, :, :2]))
    np.testing.assert_array_equal(idx, idx_ref)
    np.testing.assert_array_equal(d2, d2_ref)
    assert idx.max() < 1200


def _pair_batch(B, n, m, dim, pad_n=0, pad_m=0, seed=0, with_normals=True):
    S, Tg, Tt = [], [], []
    for b in range(B):
        s, t, T = synthetic.simple_cloud_pair(seed + b, n, m, dim=dim, pad_n=pad_n, pad_m=pad_m,
                                              with_normals=with_normals, yaw=0.02 + 0.01 * b, trans=(0.6, -0.4 + 0.1 * b, 0.1))
        S.append(s), Tg.append(t), Tt.append(T)
    return np.stack(S), np.stack(Tg), np.stack(Tt)


CASES = [("pt2pt", "cauchy", 2), ("pt2pl", "huber", 2), ("pt2pt", "huber", 3), ("pt2pl", "cauchy", 3),
         ("pt2pl", None, 2)]


@pytest.mark.parametrize("icp_type,loss,dim", CASES)
def test_icp_forward_matches_oracle(icp_type, loss, dim):
    B, n, m = 3, 1500, 4000
    src, tgt, T_true = _pair_batch(B, n, m, dim, pad_n=100, pad_m=96, seed=10 * dim)
    rng = np.random.default_rng(1)
    w = rng.uniform(0.1, 1.0, (B, n + 100)).astype(np.float32)
    w[:, n:] = 0.0
    loss_fn = None if loss is None else {"name": loss, "metric": 1.0}
    K = 12
    ref = dicp_ref.ICPRef(icp_type, differentiable=False, max_iterations=K, tolerance=1e-9)
    out = ref.icp(torch.from_numpy(src), torch.from_numpy(tgt), T_init=torch.eye(4).repeat(B, 1, 1),
                  weight=torch.from_numpy(w), trim_dist=5.0, loss_fn=loss_fn, dim=dim)
    # run the GPU with save_state to read every iteration's correspondences
    icp = ICP(icp_type=icp_type, differentiable=True, max_iterations=K, tolerance=1e-9)
    wt = torch.from_numpy(w).to(DEV).requires_grad_(True)
    T = icp.icp(torch.from_numpy(src).to(DEV), torch.from_numpy(tgt).to(DEV), T_init=torch.eye(4, device=DEV).repeat(B, 1, 1),
                weight=wt, trim_dist=5.0, loss_fn=loss_fn, dim=dim)["T"]
    saved = T.grad_fn.saved_tensors
    idx_hist, T_hist = saved[3].cpu().numpy(), saved[4].cpu().numpy().reshape(K + 1, B, 4, 4)
    for k in range(out["num_iter"]):
        act = out["hist"]["active"][k].numpy()
        np.testing.assert_array_equal(idx_hist[k][act], out["hist"]["idx"][k].numpy()[act], err_msg="iteration %d" % k)
        np.testing.assert_allclose(T_hist[k + 1], out["hist"]["T"][k + 1].numpy(), atol=2e-6)
    Tg = T.detach().cpu().numpy()
    np.testing.assert_allclose(Tg, out["T"].numpy(), atol=2e-6)
    # known answer: the clouds are copies related by T_true (1 cm noise); point-to-plane
    # recovers it, point-to-point at least closes most of the initial gap in K steps
    err = np.abs(Tg[:, :dim, 3] - T_true[:, :dim, 3]).max()
    err0 = np.abs(T_true[:, :dim, 3]).max()
    if icp_type == "pt2pl":
        assert err < 0.05 and np.abs(Tg[:, 1, 0] - T_true[:, 1, 0]).max() < 5e-3
    else:
        assert err < 0.5 * err0


def test_icp_zero_rows_anywhere_match_oracle():
    """The scan visits identical all-zero rows once (mmk_icp.hip: src_zero_scan_kernel / src_units_kernel).  Zero rows need
    not be trailing padding: a whole 512-row block of zeros in the middle of a scan, zero rows at the front, a scan of
    nothing but zeros and one without any -- correspondences of every row and every iteration equal the oracle's."""
    B, n, m, dim = 4, 2048, 3000, 2
    src, tgt, _ = _pair_batch(B, n, m, dim, pad_n=0, pad_m=40, seed=91)
    src[0, 512:1024] = 0.0                 # a zero block in the middle, real rows behind it
    src[0, 100:110] = 0.0                  # ... and the first zero row in an earlier, mixed block
    src[1, :600] = 0.0                     # zeros at the front (the first block is all zero and holds the representative)
    src[2] = 0.0                           # nothing but zeros
    rng = np.random.default_rng(4)
    w = rng.uniform(0.1, 1.0, (B, n)).astype(np.float32)
    K = 6
    loss_fn = {"name": "huber", "metric": 1.0}
    ref = dicp_ref.ICPRef("pt2pl", differentiable=False, max_iterations=K, tolerance=1e-9)
    out = ref.icp(torch.from_numpy(src), torch.from_numpy(tgt), T_init=torch.eye(4).repeat(B, 1, 1),
                  weight=torch.from_numpy(w), trim_dist=5.0, loss_fn=loss_fn, dim=dim)
    icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=K, tolerance=1e-9)
    wt = torch.from_numpy(w).to(DEV).requires_grad_(True)
    T = icp.icp(torch.from_numpy(src).to(DEV), torch.from_numpy(tgt).to(DEV), T_init=torch.eye(4, device=DEV).repeat(B, 1, 1),
                weight=wt, trim_dist=5.0, loss_fn=loss_fn, dim=dim)["T"]
    saved = T.grad_fn.saved_tensors
    idx_hist, T_hist = saved[3].cpu().numpy(), saved[4].cpu().numpy().reshape(K + 1, B, 4, 4)
    for k in range(out["num_iter"]):
        act = out["hist"]["active"][k].numpy()
        np.testing.assert_array_equal(idx_hist[k][act], out["hist"]["idx"][k].numpy()[act], err_msg="iteration %d" % k)
        np.testing.assert_allclose(T_hist[k + 1], out["hist"]["T"][k + 1].numpy(), atol=2e-6)


@pytest.mark.parametrize("icp_type,loss,dim", CASES[:4])
def test_icp_backward_matches_autograd(icp_type, loss, dim):
    B, n, m = 2, 900, 2500
    src, tgt, _ = _pair_batch(B, n, m, dim, pad_n=60, seed=77 + dim)
    rng = np.random.default_rng(2)
    w0 = rng.uniform(0.2, 1.0, (B, n + 60)).astype(np.float32)
    w0[:, n:] = 0.0
    loss_fn = {"name": loss, "metric": 0.5}
    K = 5
    G = torch.from_numpy(rng.normal(size=(B, 4, 4)).astype(np.float32))
    T0 = torch.from_numpy(_rand_T(rng, B, dim) * 0 + np.eye(4, dtype=np.float32))
    T0[:, 0, 3] += 0.2

    wr = torch.from_numpy(w0).requires_grad_(True)
    T0r = T0.clone().requires_grad_(True)
    ref = dicp_ref.ICPRef(icp_type, differentiable=True, max_iterations=K, tolerance=1e-9)
    out = ref.icp(torch.from_numpy(src), torch.from_numpy(tgt), T_init=T0r, weight=wr, trim_dist=3.0,
                  loss_fn=loss_fn, dim=dim)
    (out["T"] * G).sum().backward()

    wg = torch.from_numpy(w0).to(DEV).requires_grad_(True)
    T0g = T0.clone().to(DEV).requires_grad_(True)
    icp = ICP(icp_type=icp_type, differentiable=True, max_iterations=K, tolerance=1e-9)
    T = icp.icp(torch.from_numpy(src).to(DEV), torch.from_numpy(tgt).to(DEV), T_init=T0g, weight=wg, trim_dist=3.0,
                loss_fn=loss_fn, dim=dim)["T"]
    (T * G.to(DEV)).sum().backward()
    np.testing.assert_allclose(T.detach().cpu().numpy(), out["T"].detach().numpy(), atol=2e-6)
    gw, gw_ref = wg.grad.cpu().numpy(), wr.grad.numpy()
    scale = np.abs(gw_ref).max()
    assert scale > 0
    assert np.abs(gw - gw_ref).max() <= 2e-3 * scale, (np.abs(gw - gw_ref).max(), scale)
    gT, gT_ref = T0g.grad.cpu().numpy(), T0r.grad.numpy()
    assert np.abs(gT - gT_ref).max() <= 2e-3 * np.abs(gT_ref).max()


def test_icp_invariances():
    B, n, m = 2, 800, 2000
    src, tgt, _ = _pair_batch(B, n, m, 2, seed=31)
    s, t = torch.from_numpy(src).to(DEV), torch.from_numpy(tgt).to(DEV)
    w = torch.rand(B, n, device=DEV) + 0.1
    icp = ICP("pt2pl", differentiable=False, max_iterations=6, tolerance=1e-9)
    kw = dict(trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0}, dim=2)
    T1 = icp.icp(s, t, T_init=torch.eye(4, device=DEV).repeat(B, 1, 1), weight=w, **kw)["T"]
    # weight scaling leaves the Gauss-Newton step unchanged
    T2 = icp.icp(s, t, T_init=torch.eye(4, device=DEV).repeat(B, 1, 1), weight=4.0 * w, **kw)["T"]
    assert (T1 - T2).abs().max().item() < 1e-5
    # zero-weight padded source rows and target_pad_val target rows change nothing
    sp = torch.cat([s, torch.zeros(B, 50, 3, device=DEV)], 1)
    wp = torch.cat([w, torch.zeros(B, 50, device=DEV)], 1)
    tp = torch.cat([t, torch.full((B, 77, 6), icp.target_pad_val, device=DEV)], 1)
    T3 = icp.icp(sp, tp, T_init=torch.eye(4, device=DEV).repeat(B, 1, 1), weight=wp, **kw)["T"]
    assert (T1 - T3).abs().max().item() < 1e-6
    # run-to-run determinism (fixed reduction order, no float atomics in the ICP)
    T4 = icp.icp(s, t, T_init=torch.eye(4, device=DEV).repeat(B, 1, 1), weight=w, **kw)["T"]
    assert torch.equal(T1, T4)
    # returns a dict with key 'T' on the caller's device, also for CPU inputs
    T5 = icp.icp(s.cpu(), t.cpu(), T_init=torch.eye(4).repeat(B, 1, 1), weight=w.cpu(), **kw)["T"]
    assert T5.device.type == "cpu" and (T5 - T1.cpu()).abs().max().item() == 0.0


def test_icp_tolerance_freezes_pairs_and_early_exit():
    B, n, m = 2, 600, 1500
    src, tgt, T_true = _pair_batch(B, n, m, 2, seed=3)
    src[1] = tgt[1, :n, :3]              # pair 1 starts converged (exact copy)
    s, t = torch.from_numpy(src).to(DEV), torch.from_numpy(tgt).to(DEV)
    icp = ICP("pt2pt", differentiable=False, max_iterations=50, tolerance=1e-5)
    T = icp.icp(s, t, T_init=torch.eye(4, device=DEV).repeat(B, 1, 1), trim_dist=5.0,
                loss_fn={"name": "cauchy", "metric": 1.0}, dim=2)["T"]
    assert icp.last_iterations < 50                     # stopped by the polled flags
    act = icp.last_state["active"].cpu().numpy()
    assert act[0].tolist() == [1, 1] and act[1, 1] == 0   # pair 1 froze after its first (zero) step
    ref = dicp_ref.ICPRef("pt2pt", differentiable=False, max_iterations=50, tolerance=1e-5)
    out = ref.icp(torch.from_numpy(src), torch.from_numpy(tgt), T_init=torch.eye(4).repeat(B, 1, 1), trim_dist=5.0,
                  loss_fn={"name": "cauchy", "metric": 1.0}, dim=2)
    np.testing.assert_allclose(T.cpu().numpy(), out["T"].numpy(), atol=2e-6)
    assert np.abs(T.cpu().numpy()[0, :2, 3] - T_true[0, :2, 3]).max() < 0.05


def test_grid_nn_far_queries_and_clamped_targets():
    """Grid engine corner cases: queries far outside the grid / far from every target (exhaustive
    fallback), targets outside the grid (clamped to border cells), exact duplicates (lowest index)."""
    rng = np.random.default_rng(3)
    B, N, M = 2, 600, 3000
    tgt = np.zeros((B, M, 6), np.float32)
    tgt[:, :, :2] = rng.uniform(-60, 60, (B, M, 2))
    tgt[:, :50, :2] = rng.uniform(150, 400, (B, 50, 2))          # beyond the 256 m grid
    tgt[:, 2900:, :] = 1000.0                                    # target_pad_val rows
    tgt[:, 100, :2] = tgt[:, 7, :2]                              # duplicate point
    src = np.zeros((B, N, 3), np.float32)
    src[:, :, :2] = rng.uniform(-70, 70, (B, N, 2))
    src[:, :40, :2] = rng.uniform(200, 500, (B, 40, 2))          # far queries
    src[:, 40:60, :2] = rng.uniform(-500, -130, (B, 20, 2))
    src[:, 60, :2] = tgt[:, 7, :2]
    T0 = torch.eye(4).repeat(B, 1, 1)
    kw = dict(trim_dist=5.0, loss_fn={"name": "cauchy", "metric": 1.0}, dim=2)
    ref = dicp_ref.ICPRef("pt2pt", differentiable=False, max_iterations=2, tolerance=1e-9)
    out = ref.icp(torch.from_numpy(src), torch.from_numpy(tgt), T_init=T0, **kw)
    icp = ICP("pt2pt", differentiable=True, max_iterations=2, tolerance=1e-9)
    w = torch.ones(B, N, device=DEV, requires_grad=True)
    T = icp.icp(torch.from_numpy(src).to(DEV), torch.from_numpy(tgt).to(DEV), T_init=T0.to(DEV), weight=w, **kw)["T"]
    idx = T.grad_fn.saved_tensors[3].cpu().numpy()
    for k in range(2):
        np.testing.assert_array_equal(idx[k], out["hist"]["idx"][k].numpy())
    assert idx[0][0, 60] == 7
    np.testing.assert_allclose(T.detach().cpu().numpy(), out["T"].numpy(), atol=2e-6)


def test_icp_error_behaviour():
    icp = ICP("pt2pl", differentiable=False, max_iterations=2)
    s = torch.zeros(1, 8, 3, device=DEV)
    with pytest.raises(ValueError):
        icp.icp(s, torch.zeros(1, 8, 3, device=DEV), dim=2)            # pt2pl without normals
    with pytest.raises(ValueError):
        icp.icp(s, torch.zeros(1, 8, 6, device=DEV), loss_fn={"name": "tukey", "metric": 1.0}, dim=2)
    with pytest.raises(ValueError):
        icp.icp(s, torch.zeros(1, 8, 6, device=DEV), dim=4)
    p = _lib.IcpParams(B=1, N=8, M=8, tgt_cols=6, dim=2, icp_type=1, loss=2, loss_k=1.0, trim_dist=5.0, tolerance=0.0,
                       max_iter=2, save_state=0, check_every=0, nn_method=0)
    L = _lib.lib()
    rc = L.mmk_icp_forward(ctypes.byref(p), *([ctypes.c_void_p(8)] * 10), ctypes.c_void_p(0), 0, None, ctypes.c_void_p(0))
    assert rc == -3 and b"workspace" in L.mmk_last_error()
    # degenerate geometry (all weights zero): delta = 0, pose unchanged
    T0 = torch.eye(4, device=DEV).repeat(1, 1, 1)
    T0[0, 0, 3] = 0.3
    T = ICP("pt2pt", differentiable=False, max_iterations=3).icp(
        torch.rand(1, 16, 3, device=DEV), torch.rand(1, 16, 6, device=DEV), T_init=T0, weight=torch.zeros(1, 16, device=DEV),
        dim=2)["T"]
    assert torch.equal(T, T0)


def test_config5_50k_fp32_gate():
    """BASELINE.json configs[4]: 50k-point lidar submap, fp32 end to end, pose within
    1e-3 m / 1e-4 rad of the CPU restatement, correspondences bit-exact."""
    B = 2
    raws = [synthetic.make_pair(100 + b, m_valid=50000, m_pad=50176, pos_std=0.5, rot_std=0.03) for b in range(B)]
    tgt = np.stack([r["map_pc"] for r in raws])
    rng = np.random.default_rng(0)
    # scan = noisy sub-sample of the map seen from the perturbed pose
    src = np.zeros((B, 5120, 3), np.float32)
    T0 = np.stack([r["T_init"] for r in raws])
    for b in range(B):
        sel = rng.choice(50000, 4500, replace=False)
        src[b, :4500, :2] = tgt[b, sel, :2] + rng.normal(0, 0.03, (4500, 2))
    w = np.ones((B, 5120), np.float32)
    w[:, 4500:] = 0
    K = 10
    kw = dict(trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0}, dim=2)
    ref = dicp_ref.ICPRef("pt2pl", differentiable=False, max_iterations=K, tolerance=1e-5)
    out = ref.icp(torch.from_numpy(src), torch.from_numpy(tgt), T_init=torch.from_numpy(T0), weight=torch.from_numpy(w), **kw)
    icp = ICP("pt2pl", differentiable=True, max_iterations=K, tolerance=1e-5)
    wt = torch.from_numpy(w).to(DEV).requires_grad_(True)
    T = icp.icp(torch.from_numpy(src).to(DEV), torch.from_numpy(tgt).to(DEV), T_init=torch.from_numpy(T0).to(DEV), weight=wt, **kw)["T"]
    idx_hist = T.grad_fn.saved_tensors[3].cpu().numpy()
    for k in range(out["num_iter"]):
        act = out["hist"]["active"][k].numpy()          # frozen pairs skip the NN kernel
        np.testing.assert_array_equal(idx_hist[k][act], out["hist"]["idx"][k].numpy()[act])
    Tg, Tr = T.detach().cpu().numpy(), out["T"].numpy()
    assert np.abs(Tg[:, :2, 3] - Tr[:, :2, 3]).max() <= 1e-3
    assert np.abs(np.arctan2(Tg[:, 1, 0], Tg[:, 0, 0]) - np.arctan2(Tr[:, 1, 0], Tr[:, 0, 0])).max() <= 1e-4
    # and the ICP actually localises: back to identity within a few cm
    assert np.abs(Tg[:, :2, 3]).max() < 0.1 and np.abs(Tg[:, 1, 0]).max() < 5e-3


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_config5_50k_through_the_policy_with_override_mask(mode):
    """BASELINE.json configs[4] as SURVEY.md 8d writes it: B = 4, M = 50 000 valid / 50 176 padded, fp32 end to end, the U-Net
    bypassed with ``override_mask=ones`` -- i.e. the ``generate_baseline`` call path
    (/root/reference/mm_masking/train_icp_weights.py:298-319 -> icp_weight_policy.py:136,188-189,277-288): through
    ``LearnICPWeightPolicy.forward`` in eval mode (ICP_alg_inference: 50 iterations, tolerance 1e-5) and in train mode
    (ICP_alg: 10 iterations), under no_grad as upstream.  Correspondences bit-exact, pose within 1e-3 m / 1e-4 rad of the CPU
    restatement fed the oracle's own extract_weights of the same mask."""
    from mm_masking_amd import train_icp_weights as trn
    from mm_masking_amd.icp_weight_policy import LearnICPWeightPolicy
    from oracle import radar_ref, train_ref
    B, MV, MP, N, NV = 4, 50000, 50176, 5120, 4500
    raws = [synthetic.make_pair(100 + b, m_valid=MV, m_pad=MP, pos_std=0.5, rot_std=0.03) for b in range(B)]
    tgt = np.stack([r["map_pc"] for r in raws])
    assert tgt.shape == (B, MP, 6)
    rng = np.random.default_rng(0)
    src = np.zeros((B, N, 3), np.float32)
    T0 = np.stack([r["T_init"] for r in raws])
    for b in range(B):
        sel = rng.choice(MV, NV, replace=False)
        src[b, :NV, :2] = tgt[b, sel, :2] + rng.normal(0, 0.03, (NV, 2))
    lf = {"name": "huber", "metric": 1.0}
    params = trn.default_params(DEV)
    params.update({"icp_type": "pt2pl", "icp_loss_fn": lf, "max_iter": 10})
    torch.manual_seed(0)
    model = LearnICPWeightPolicy(params).to(DEV)
    assert model.ICP_alg.nn_search == model.ICP_alg_inference.nn_search == ICP.NN_SEARCH_OVERRIDE
    model.train() if mode == "train" else model.eval()
    ones = torch.ones(B, 640, 640)
    zeros = torch.zeros(B, 640, 640)
    scan = {"fft_data": zeros, "fft_cfar": zeros, "raw_pc": torch.from_numpy(src), "filtered_pc": torch.from_numpy(src)}
    batch = {"loc_data": scan, "map_data": {"pc": torch.from_numpy(tgt)},
             "transforms": {"T_ml_init": torch.from_numpy(T0), "T_ml_gt": torch.eye(4).repeat(B, 1, 1)}}
    with torch.no_grad():
        T, mask, _ = model(scan, batch["map_data"], torch.from_numpy(T0).to(DEV), override_mask=ones)
    alg = model.ICP_alg if mode == "train" else model.ICP_alg_inference
    idx_gpu = alg.last_state["idx"][0].cpu().numpy()
    assert torch.equal(mask.cpu(), ones)                        # ones / amax(ones)
    # the oracle on the same inputs: its own bilinear sampling of the same mask, then the CPU dICP
    w_ref = radar_ref.extract_weights(ones.numpy(), src)[0]
    # (1 inside the image, less on its border rows / columns and outside: the map reaches beyond the 76 m half-width; 0 on zero rows)
    assert (w_ref[:, NV:] == 0).all() and (w_ref[:, :NV] == 1).mean() > 0.9
    K = 10 if mode == "train" else 50
    ref = dicp_ref.ICPRef("pt2pl", differentiable=False, max_iterations=K, tolerance=1e-5)
    out = ref.icp(torch.from_numpy(src), torch.from_numpy(tgt), T_init=torch.from_numpy(T0), weight=torch.from_numpy(w_ref),
                  trim_dist=5.0, loss_fn=lf, dim=2)
    # a pair that froze keeps the correspondences of its last active iteration (frozen pairs skip the search)
    act = np.stack([a.numpy() for a in out["hist"]["active"]])           # (iterations, B)
    for b in range(B):
        k_last = int(np.nonzero(act[:, b])[0].max())
        np.testing.assert_array_equal(idx_gpu[b], out["hist"]["idx"][k_last][b].numpy(), err_msg="pair %d" % b)
    Tg, Tr = T.cpu().numpy(), out["T"].numpy()
    assert np.abs(Tg[:, :2, 3] - Tr[:, :2, 3]).max() <= 1e-3
    assert np.abs(np.arctan2(Tg[:, 1, 0], Tg[:, 0, 0]) - np.arctan2(Tr[:, 1, 0], Tr[:, 0, 0])).max() <= 1e-4
    assert np.abs(Tg[:, :2, 3]).max() < 0.1 and np.abs(Tg[:, 1, 0]).max() < 5e-3          # and it localises
    if mode == "eval":
        assert alg.last_iterations < 50 and out["num_iter"] < 50      # both stopped on the tolerance
    # the same through generate_baseline itself (the "ones" branch: every mask-loss weight 0)
    li, lo = trn.generate_baseline(model, [batch], baseline_type="val" if mode == "eval" else "train", device=DEV,
                                   loss_weights={"icp": 1.0, "icp_rot": 1.0, "icp_trans": 1.0, "fft": 0.0, "mask_pts": 0.0, "cfar": 0.0,
                                                 "num_pts": 0.0})
    eye = torch.eye(4).repeat(B, 1, 1)
    if mode == "eval":
        want_i = float(train_ref.eval_validation_loss(torch.from_numpy(T0), eye)[0])
        want_o = float(train_ref.eval_validation_loss(out["T"], eye)[0])
    else:
        lw = dict(train_ref.DEFAULT_LOSS_WEIGHTS, mask_pts=0.0)
        want_i = float(train_ref.eval_training_loss(torch.from_numpy(T0), ones, None, eye, zeros, None, None, None, lw)[0])
        want_o = float(train_ref.eval_training_loss(out["T"], ones, None, eye, zeros, None, None, None, lw)[0])
    assert abs(li - want_i) <= 1e-6 * max(1.0, abs(want_i)) and abs(lo - want_o) <= 2e-5 + 1e-3 * abs(want_o), (li, want_i, lo, want_o)


# ----------------------------------------------------------------------------- known answers built outside our code
import dicp_kat as kat  # noqa: E402


def _run_hip(icp_type, src, tgt, T0, dim, K, loss=None, trim=50.0, weight=None, tol=1e-12):
    icp = ICP(icp_type, differentiable=False, max_iterations=K, tolerance=tol)
    icp.check_every = 0 if K < 8 else icp.check_every
    lf = None if loss is None else {"name": loss, "metric": 1.0}
    w = None if weight is None else torch.from_numpy(np.asarray(weight, np.float32))[None].to(DEV)
    T = icp.icp(torch.from_numpy(src)[None].to(DEV), torch.from_numpy(tgt)[None].to(DEV),
                T_init=torch.from_numpy(np.asarray(T0, np.float32))[None].to(DEV), weight=w, trim_dist=trim, loss_fn=lf, dim=dim)["T"]
    return T[0].cpu().numpy(), icp.last_state


@pytest.mark.parametrize("dim", [2, 3])
@pytest.mark.parametrize("icp_type", ["pt2pt", "pt2pl"])
def test_kat_noise_free_copy_recovers_transform(icp_type, dim):
    """Noise-free copy clouds (tests/dicp_kat.py), offset inside the reference's perturbation envelope: the HIP ICP
    recovers the transform to north_star's gate 1e-3 m / 1e-4 rad (in fact to fp32 rounding of the clouds)."""
    src, tgt, T_true = kat.jittered_grid_pair(dim, seed=3 + dim)
    T, _ = _run_hip(icp_type, src, tgt, np.eye(4), dim, K=30)
    dt, da = kat.pose_errors(T, T_true, dim)
    assert dt <= 1e-3 and da <= 1e-4, (dt, da)
    assert dt <= 2e-5 and da <= 2e-6, (dt, da)


@pytest.mark.parametrize("icp_type,dim,loss", [("pt2pt", 2, None), ("pt2pl", 2, "huber"), ("pt2pt", 3, "cauchy"), ("pt2pl", 3, None),
                                               ("pt2pl", 2, "cauchy"), ("pt2pt", 2, "huber")])
def test_kat_one_gauss_newton_step_from_definitions(icp_type, dim, loss):
    """One HIP iteration against a step computed from the definitions alone (finite-difference Jacobian of
    scipy's matrix exponential, exhaustive argmin, numpy solve)."""
    src, tgt, _ = kat.jittered_grid_pair(dim, seed=11)
    rng = np.random.default_rng(5)
    w = rng.uniform(0.2, 1.0, len(src)).astype(np.float32)
    T0 = kat.exp_se3([0.3, -0.2, 0.1 if dim == 3 else 0, 0.01 if dim == 3 else 0, 0, 0.03]).astype(np.float32)
    T1, delta, idx = kat.gauss_newton_step(src, tgt, T0, w, icp_type, dim, loss=loss, k=1.0, trim=5.0)
    T, st = _run_hip(icp_type, src, tgt, T0, dim, K=1, loss=loss, trim=5.0, weight=w)
    assert np.array_equal(st["idx"][0, 0].cpu().numpy(), idx)
    got = st["delta"][0, 0].cpu().numpy()[:len(delta)]
    np.testing.assert_allclose(got, delta, rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(T, T1, atol=3e-6)


@pytest.mark.parametrize("r2,loss", [(1.0 - 1e-3, "huber"), (1.0, "huber"), (1.0 + 1e-3, "huber"), (3.0, "huber"),
                                     (0.7, "cauchy"), (2.0, None)])
def test_kat_huber_kink_closed_form(r2, loss):
    src, tgt, (n1, n2, r1) = kat.two_group_lines(r2)
    _, st = _run_hip("pt2pl", src, tgt, np.eye(4), 2, K=1, loss=loss, trim=5.0)
    d = st["delta"][0, 0].cpu().numpy()
    want = kat.two_group_delta_y(r2, n1, n2, r1, loss, 1.0, 5.0)
    assert abs(d[1] - want) <= 2e-7 * max(1.0, abs(want)) and abs(d[0]) < 1e-9 and abs(d[2]) < 1e-9


@pytest.mark.parametrize("r2,kept", [(5.0 - 1e-3, True), (5.0, False), (5.0 + 1e-3, False)])
def test_kat_trim_boundary_closed_form(r2, kept):
    src, tgt, (n1, n2, r1) = kat.two_group_lines(r2)
    _, st = _run_hip("pt2pl", src, tgt, np.eye(4), 2, K=1, loss=None, trim=5.0)
    d = st["delta"][0, 0].cpu().numpy()
    want = kat.two_group_delta_y(r2, n1, n2, r1, None, 1.0, 5.0)
    assert abs(d[1] - want) <= 2e-7 * max(1.0, abs(want))
    assert (abs(want - r1) < 1e-7) == (not kept)


@pytest.mark.parametrize("dim,icp_type", [(2, "pt2pl"), (3, "pt2pt")])
def test_icp_stage_entries_equal_one_forward_iteration_and_flag_unarmed_keys(dim, icp_type):
    """mmk_nn_search -> keys -> mmk_icp_accumulate -> mmk_icp_solve_update == one iteration of mmk_icp_forward (bit for bit:
    same kernels), and a key that no search kernel wrote raises MMK_ICP_STATUS_UNARMED_KEY instead of passing as a
    correspondence."""
    import ctypes
    from mm_masking_amd import _lib
    import importlib
    icp_mod = importlib.import_module("mm_masking_amd.dICP.ICP")       # the module (the package re-exports the class under the same name)
    ICP = icp_mod.ICP
    L = _lib.lib()
    rng = np.random.default_rng(40 + dim)
    B, N, M = 3, 700, 2100
    tgt = rng.uniform(-40, 40, (B, M, 6)).astype(np.float32)
    nrm = rng.normal(size=(B, M, 3))
    if dim == 2:
        nrm[..., 2] = 0
    tgt[..., 3:] = (nrm / np.linalg.norm(nrm, axis=-1, keepdims=True)).astype(np.float32)
    if dim == 2:
        tgt[..., 2] = 0
    src = (tgt[:, rng.permutation(M)[:N], :3] + rng.normal(0, 0.05, (B, N, 3))).astype(np.float32)
    w = rng.uniform(0.1, 1, (B, N)).astype(np.float32)
    T0 = np.tile(np.eye(4, dtype=np.float32), (B, 1, 1))
    T0[:, 0, 3] = 0.4
    T0[:, 1, 3] = -0.3
    s, t, wt, Tt = (torch.from_numpy(a).to(DEV) for a in (src, tgt, w, T0.reshape(B, 16)))
    lf = {"name": "huber", "metric": 1.0}
    icp = ICP(icp_type, differentiable=False, max_iterations=1, tolerance=0.0)
    ICP.NN_SEARCH_OVERRIDE = None
    ref = icp.icp(s, t, T_init=Tt.view(B, 4, 4), weight=wt, trim_dist=5.0, loss_fn=lf, dim=dim)["T"]
    idx_ref = icp.last_state["idx"][0]
    icp_mod.check_errors(wait=True)                                   # the clean call raised nothing

    p = icp._params(B, N, M, 6, dim, lf, 5.0, save_state=False)
    Mpad = L.mmk_nn_padded_m(M)
    planar = torch.empty(B, dim, Mpad, dtype=torch.float32, device=DEV)
    _lib.check(L.mmk_pack_target(_lib.ptr(t), B, M, 6, dim, _lib.ptr(planar), _lib.stream_ptr(DEV)))
    ws = torch.empty(L.mmk_nn_workspace_bytes(B, N, M, dim), dtype=torch.uint8, device=DEV)
    idx = torch.empty(B, N, dtype=torch.int32, device=DEV)
    d2 = torch.empty(B, N, dtype=torch.float32, device=DEV)
    _lib.check(L.mmk_nn_search(_lib.ptr(s), _lib.ptr(planar), _lib.ptr(Tt), B, N, M, dim, _lib.ptr(idx), _lib.ptr(d2),
                               _lib.ptr(ws), ws.numel(), _lib.stream_ptr(DEV)))
    keys = (d2.view(torch.int32).to(torch.int64) << 32) | idx.to(torch.int64)
    n_part = L.mmk_icp_partials_count(ctypes.byref(p))
    assert n_part == B * ((N + 255) // 256) * (9 if dim == 2 else 27)

    def stage(keys_t):
        parts = torch.zeros(n_part, dtype=torch.float64, device=DEV)
        idx_out = torch.full((B, N), -7, dtype=torch.int32, device=DEV)
        status = torch.zeros(1, dtype=torch.int32, device=DEV)
        _lib.check(L.mmk_icp_accumulate(ctypes.byref(p), _lib.ptr(s), _lib.ptr(t), _lib.ptr(wt), _lib.ptr(Tt), _lib.ptr(keys_t),
                                        _lib.ptr(idx_out), _lib.ptr(parts), _lib.ptr(status), _lib.stream_ptr(DEV)))
        T1 = torch.empty(B, 16, dtype=torch.float32, device=DEV)
        delta = torch.empty(B, 6, dtype=torch.float64, device=DEV)
        A = torch.empty(B, 36, dtype=torch.float64, device=DEV)
        act_in = torch.ones(B, dtype=torch.int32, device=DEV)
        act_out = torch.empty(B, dtype=torch.int32, device=DEV)
        _lib.check(L.mmk_icp_solve_update(ctypes.byref(p), _lib.ptr(parts), _lib.ptr(Tt), _lib.ptr(T1), _lib.ptr(delta), _lib.ptr(A),
                                          _lib.ptr(act_in), _lib.ptr(act_out), _lib.stream_ptr(DEV)))
        torch.cuda.synchronize()
        return idx_out, T1.view(B, 4, 4), int(status.item())

    idx_out, T1, status = stage(keys)
    assert status == 0
    assert torch.equal(idx_out, idx_ref) and torch.equal(idx_out, idx)
    assert torch.equal(T1, ref)
    # one key nobody armed, one index outside the target: flagged, and clamped so that nothing is read out of bounds
    bad = keys.clone()
    bad[1, 5] = -1                                   # NN_KEY_INIT = ~0
    bad[2, 9] = (bad[2, 9] & ~0xffffffff) | (M + 3)
    idx_bad, _, status = stage(bad)
    assert status == icp_mod.ICP_STATUS_UNARMED_KEY
    assert int(idx_bad[1, 5]) == M - 1 and int(idx_bad[2, 9]) == M - 1
    same = torch.ones(B, N, dtype=torch.bool, device=DEV)
    same[1, 5] = same[2, 9] = False
    assert torch.equal(idx_bad[same], idx_ref[same])
    # the Python watch turns a raised flag into an error at the next check
    slot = torch.zeros(1, dtype=torch.int32).pin_memory()
    slot[0] = icp_mod.ICP_STATUS_UNARMED_KEY
    ev = torch.cuda.Event()
    ev.record()
    icp_mod._status.pending.append((ev, slot))
    with pytest.raises(_lib.MmkError, match="UNARMED_KEY"):
        icp_mod.check_errors(wait=True)
    icp_mod.check_errors(wait=True)                  # consumed: clean again
