"""Parity of the HIP radar operators (through the C ABI, via the radar_utils mirror)
with the golden vectors captured from the reference and with the numpy oracle."""
import os

import numpy as np
import pytest
import torch

from mm_masking_amd import radar_utils as ru
from mm_masking_amd import synthetic
from oracle import radar_ref as R

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def _g(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(DEV)


def _mask_close(raw, got, want, thres, band=1e-6):
    bad = got != want
    if bad.any():
        assert np.all(np.abs(raw[bad] - thres[bad]) < band)
    assert bad.mean() < 1e-4


def test_cfar_golden_and_oracle(golden_dir):
    g = _load(golden_dir, "radar_cfar.npz")
    raw = g["raw"]
    th = R.cfar_threshold(raw, 0.0596)
    hard = ru.cfar_mask(_g(raw), 0.0596, diff=False).cpu().numpy()
    _mask_close(raw, hard, g["hard"].astype(np.float32), th)
    soft = ru.cfar_mask(_g(raw), 0.0596, diff=True).cpu().numpy()
    d = np.abs(soft - g["soft"])
    assert (d > 0.5).mean() < 1e-4 and d[d <= 0.5].max() < 2e-5
    kw = {k: v for k, v in zip(g["kw2_keys"], g["kw2_vals"])}
    kw["width"], kw["guard"] = int(kw["width"]), int(kw["guard"])
    raw2 = g["raw2"]
    _mask_close(raw2, ru.cfar_mask(_g(raw2), 0.2, diff=False, **kw).cpu().numpy(), g["hard2"].astype(np.float32),
                R.cfar_threshold(raw2, 0.2, **kw))
    d2 = np.abs(ru.cfar_mask(_g(raw2), 0.2, diff=True, steep_fact=7.0, **kw).cpu().numpy() - g["soft2"])
    assert (d2 > 0.5).mean() < 1e-3 and d2[d2 <= 0.5].max() < 2e-5
    # CPU tensors in -> CPU tensors out (the reference runs this in DataLoader workers)
    out_cpu = ru.cfar_mask(torch.from_numpy(raw2), 0.2, diff=False, **kw)
    assert out_cpu.device.type == "cpu"
    with pytest.raises(AssertionError):
        ru.cfar_mask(torch.zeros(4, 4), 0.0596)


def test_cfar_full_size_vs_oracle():
    raw = synthetic.make_batch([3, 4])["fft_polar"].numpy()
    got = ru.cfar_mask(_g(raw), 0.0596, diff=False).cpu().numpy()
    want = R.cfar_mask(raw, 0.0596, diff=False)
    _mask_close(raw, got, want, R.cfar_threshold(raw, 0.0596))
    assert 500 < want[0].sum() < 40000


def test_extract_pc_golden(golden_dir):
    g = _load(golden_dir, "radar_peaks.npz")
    mask = g["mask"].astype(np.float32)
    pcs = ru.extract_pc(_g(mask), 0.0596, _g(g["az"]), _g(g["tm"]), diff=False)
    for got, want in zip(pcs, (g["pc0"], g["pc1"])):
        assert tuple(got.shape) == want.shape
        np.testing.assert_allclose(got.cpu().numpy(), want, atol=3e-5)
    pcT = ru.extract_pc(_g(mask), 0.0596, _g(g["az"]), _g(g["tm"]), T_ab=_g(g["T_ab"]), diff=False)
    for got, want in zip(pcT, (g["pcT0"], g["pcT1"])):
        np.testing.assert_allclose(got.cpu().numpy(), want, atol=4e-5)
    pcs_soft = ru.extract_pc(_g(g["soft_mask"]), 0.0596, _g(g["az"]), _g(g["tm"]), diff=True)
    for got, want in zip(pcs_soft, (g["pcs0"], g["pcs1"])):
        assert tuple(got.shape) == want.shape
        np.testing.assert_allclose(got.cpu().numpy(), want, atol=4e-5)
    # padded/batched variant: zero padding, counts, truncation
    pc, cnt = ru.extract_pc_padded(_g(mask), 0.0596, _g(g["az"]), _g(g["tm"]), 64, diff=False)
    assert cnt.cpu().tolist() == [g["pc0"].shape[0], g["pc1"].shape[0]]
    np.testing.assert_allclose(pc[0, :g["pc0"].shape[0]].cpu().numpy() if g["pc0"].shape[0] <= 64 else pc[0].cpu().numpy(),
                               g["pc0"][:64], atol=3e-5)
    big, _ = ru.extract_pc_padded(_g(mask), 0.0596, _g(g["az"]), _g(g["tm"]), 512, diff=False)
    n0 = g["pc0"].shape[0]
    assert torch.count_nonzero(big[0, n0:]).item() == 0
    # empty mask -> empty clouds
    empty = ru.extract_pc(torch.zeros(2, 8, 1400, device=DEV), 0.0596, _g(g["az"]), _g(g["tm"]), diff=False)
    assert [tuple(e.shape) for e in empty] == [(0, 3), (0, 3)]
    # the survey's known answer
    kat_mask = torch.zeros(1, 2, 1400, device=DEV)
    kat_mask[0, 1, 500:504] = 1.0
    kat = ru.extract_pc(kat_mask, 0.0596, torch.tensor([[0.1, 0.3]], device=DEV), torch.zeros(1, 2, device=DEV), diff=False)[0]
    np.testing.assert_allclose(kat.cpu().numpy(), g["kat"], atol=3e-5)


def test_extract_pc_full_size_vs_oracle():
    raw = synthetic.make_batch([5, 6])
    mask = R.cfar_mask(raw["fft_polar"].numpy(), 0.0596, diff=False)
    want = R.extract_pc(mask, 0.0596, raw["azimuths"].numpy(), raw["az_times"].numpy(), diff=False)
    pc, cnt = ru.extract_pc_padded(_g(mask), 0.0596, raw["azimuths"].to(DEV), raw["az_times"].to(DEV), 5120, diff=False)
    for b in range(2):
        n = want[b].shape[0]
        assert cnt[b].item() == n and 300 < n <= 5120
        np.testing.assert_allclose(pc[b, :n].cpu().numpy(), want[b], atol=5e-5)
        assert torch.count_nonzero(pc[b, n:]).item() == 0


def test_polar_to_cart_golden(golden_dir):
    g = _load(golden_dir, "radar_polar2cart.npz")
    pol = (g["pol"] / np.float32(255.0)).astype(np.float32)
    kw = dict(cart_resolution=0.9536, cart_pixel_width=160)
    got = ru.radar_polar_to_cartesian_diff(_g(pol), _g(g["az"]), 0.5, **kw).cpu().numpy()
    np.testing.assert_allclose(got, g["cart"], atol=2e-5)
    got = ru.radar_polar_to_cartesian_diff(_g(pol), _g(g["az"]), 0.5, fix_wobble=False, **kw).cpu().numpy()
    np.testing.assert_allclose(got, g["cart_nowob"], atol=2e-4)
    got = ru.radar_polar_to_cartesian_diff(_g(pol), _g(g["az"]), 0.5, interpolate_crossover=False, **kw).cpu().numpy()
    np.testing.assert_allclose(got, g["cart_nocross"], atol=2e-5)
    rb = np.random.default_rng(int(g["seed_b"]))
    pol_b = (rb.integers(0, 256, size=(1, 400, 3360), dtype=np.uint8) / np.float32(255.0)).astype(np.float32)
    cart_b = ru.radar_polar_to_cartesian_diff(_g(pol_b), _g(g["az_b"]), 0.0596).cpu().numpy()
    # white noise sampled at range bins up to ~1800: one fp32 ulp of the bin coordinate
    # (torch.linspace differs by an ulp between hosts) times the pixel contrast
    np.testing.assert_allclose(cart_b[:, ::8, ::8], g["cart_b_sub"], atol=5e-4)
    np.testing.assert_allclose(cart_b[0, 200], g["cart_b_row"], atol=5e-4)
    assert abs(cart_b.astype(np.float64).sum() - g["cart_b_sum"]) < 0.5


def test_extract_weights_golden(golden_dir):
    g = _load(golden_dir, "radar_points.npz")
    pts = g["pts"]
    mask = np.random.default_rng(int(g["seed_mask"])).uniform(0, 1, size=(2, 640, 640)).astype(np.float32)
    mt = _g(mask).requires_grad_(True)
    w, dmn, mn, mean_w, max_w, min_w = ru.extract_weights(mt, _g(pts))
    np.testing.assert_allclose(w.detach().cpu().numpy(), g["weights"], atol=2e-5)
    np.testing.assert_allclose([dmn.item(), mn.item(), mean_w.item(), max_w.item(), min_w.item()], g["stats"],
                               rtol=1e-5, atol=2e-5)
    (w * _g(g["grad_w"])).sum().backward()
    want = np.zeros_like(mask)
    i = g["grad_nz_idx"]
    want[i[0], i[1], i[2]] = g["grad_nz_val"]
    np.testing.assert_allclose(mt.grad.cpu().numpy(), want, atol=3e-5)
    # SURVEY.md §8a R9 known answers: pixel centre -> mask value, fake -> 0
    ones = torch.ones(1, 640, 640, device=DEV)
    pk = torch.tensor([[[0.1192, 0.1192, 0.0], [0.0, 0.0, 0.0], [500.0, 0.0, 0.0]]], device=DEV)
    wk = ru.extract_weights(ones, pk)[0].cpu().numpy()
    np.testing.assert_allclose(wk[0], [1.0, 0.0, 0.0], atol=1e-6)


@pytest.mark.parametrize("tag", ["s", "f"])
def test_extract_weights_non_square_mask(golden_dir, tag):
    """Masks that are not the 640 x 640 Cartesian grid — (40,96) and the polar (400,3360): points are
    normalised by the 640-pixel width, grid_sample stretches [-1,1] over the mask (radar_utils.py:112,126)."""
    g = _load(golden_dir, "polar_net.npz")
    B, H, W = (int(v) for v in g["ew_shape_" + tag])
    mask = np.random.default_rng(int(g["ew_seed_" + tag])).uniform(0, 1, size=(B, H, W)).astype(np.float32)
    mt = _g(mask).requires_grad_(True)
    w, dmn, mn, mean_w, max_w, min_w = ru.extract_weights(mt, _g(g["ew_pts_" + tag]))
    np.testing.assert_allclose(w.detach().cpu().numpy(), g["ew_w_" + tag], atol=2e-5)
    np.testing.assert_allclose([dmn.item(), mn.item(), mean_w.item(), max_w.item(), min_w.item()], g["ew_stats_" + tag],
                               rtol=1e-5, atol=2e-5)
    (w * _g(g["ew_gw_" + tag])).sum().backward()
    want = np.zeros_like(mask)
    i = g["ew_gidx_" + tag]
    want[i[0], i[1], i[2]] = g["ew_gval_" + tag]
    np.testing.assert_allclose(mt.grad.cpu().numpy(), want, atol=3e-5)


def test_weight_stats_fused_pass():
    """mmk_weight_stats against the plain tensor expressions of radar_utils.py:130-138 and
    icp_weight_policy.py:209-212, including the gradient of diff_mean_num_non0."""
    g = torch.Generator().manual_seed(4)
    B, N = 3, 700
    pc = (torch.rand(B, N, 3, generator=g) * 120 - 60)
    pc[:, 600:, :] = 0.0                    # padded rows
    pc[0, 5, 0] = 0.0                       # x == 0 only: real for the weights, not counted by mean_all_pts
    pc = pc.to(DEV)
    mask = torch.rand(B, 640, 640, generator=g).to(DEV).requires_grad_(True)
    (w, dmn, mn, mean_w, max_w, min_w), st = ru._extract_weights_stats(mask, pc)
    real = ~((pc[:, :, 0] == 0) & (pc[:, :, 1] == 0))
    wd = w.detach()
    assert abs(mn.item() - ((wd > 0.05) & real).sum().item() / B) < 1e-6 * N
    assert abs(mean_w.item() - wd[real].mean().item()) < 1e-6
    assert max_w.item() == wd[real].max().item() and min_w.item() == wd[real].min().item()
    assert abs(st[5].item() - ((pc[:, :, 0] != 0) & (pc[:, :, 1] != 0)).sum().item() / B) < 1e-6 * N
    assert st[6].item() == real.sum().item()
    dmn.backward()
    got = mask.grad.clone()
    mask.grad = None
    w2 = ru._SampleWeights.apply(mask, pc, 0.2384, 640)
    ref = ((0.5 * torch.tanh(5 * w2) + 0.5) * real).sum() / B
    assert abs(ref.item() - dmn.item()) < 1e-4 * max(1.0, abs(ref.item()))
    ref.backward()
    assert (got - mask.grad).abs().max().item() < 1e-6


def test_polar_to_cart_pair_equals_two_calls():
    """The paired resampling used by prepare_batch (FFT + CFAR image, shared coordinates) gives exactly the
    values of two separate radar_polar_to_cartesian_diff calls."""
    raw = synthetic.make_batch([0, 1], device=DEV, m_valid=500, m_pad=512)
    fft, az = raw["fft_polar"], raw["azimuths"]
    cf = ru.cfar_mask(fft, 0.0596, a_thresh=1.0, b_thresh=0.09, diff=False)
    a, b = ru._polar_to_cart_pair(fft, cf, az, 0.0596)
    assert torch.equal(a, ru.radar_polar_to_cartesian_diff(fft, az, 0.0596))
    assert torch.equal(b, ru.radar_polar_to_cartesian_diff(cf, az, 0.0596))


def test_bev_golden_and_point_idx(golden_dir):
    g = _load(golden_dir, "radar_points.npz")
    bev = ru.extract_bev_from_pts(_g(g["bev_pts"])).cpu().numpy()
    want = np.zeros_like(bev)
    j = g["bev_nz_idx"]
    want[j[0], j[1], j[2]] = 1.0
    np.testing.assert_array_equal(bev, want)
    np.testing.assert_allclose(ru.point_to_cart_idx(_g(g["pts"])).cpu().numpy(), g["idx_plain"], rtol=1e-6, atol=4e-5)
    np.testing.assert_allclose(ru.point_to_cart_idx(_g(g["pts"]), min_to_plus_1=True).cpu().numpy(), g["idx_norm"],
                               rtol=1e-6, atol=1e-7)
