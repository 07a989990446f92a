"""The oracle (oracle/) against the golden vectors generated from the reference's
own modules (tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import radar_ref as R
from oracle import train_ref, unet_ref


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def _assert_mask_close(raw, got, want, thres, band=1e-6):
    """Masks may differ only where the cell sits within `band` of its threshold
    (window-sum rounding order)."""
    bad = got != want
    if bad.any():
        assert np.all(np.abs(raw[bad] - thres[bad]) < band), "CFAR mask differs away from the threshold"
    assert bad.mean() < 1e-4


def test_cfar_hard_and_soft(golden_dir):
    g = _load(golden_dir, "radar_cfar.npz")
    raw = g["raw"]
    th = R.cfar_threshold(raw, 0.0596)
    hard = R.cfar_mask(raw, 0.0596, diff=False)
    _assert_mask_close(raw, hard, g["hard"].astype(np.float32), th)
    soft = R.cfar_mask(raw, 0.0596, diff=True)
    d = np.abs(soft - g["soft"])
    # soft mask jumps 0 <-> >0.99 at the hardshrink edge; elsewhere smooth
    jump = d > 0.5
    assert jump.mean() < 1e-4 and d[~jump].max() < 2e-5
    kw = {k: v for k, v in zip(g["kw2_keys"], g["kw2_vals"])}
    for k in ("width", "guard"):
        kw[k] = int(kw[k])
    raw2 = g["raw2"]
    th2 = R.cfar_threshold(raw2, 0.2, **kw)
    _assert_mask_close(raw2, R.cfar_mask(raw2, 0.2, diff=False, **kw), g["hard2"].astype(np.float32), th2)
    d2 = np.abs(R.cfar_mask(raw2, 0.2, diff=True, steep_fact=7.0, **kw) - g["soft2"])
    assert (d2 > 0.5).mean() < 1e-3 and d2[d2 <= 0.5].max() < 2e-5
    assert R.cfar_cols(3360, 0.0596)[1:] == (89, 1287)


def test_peaks_and_extract_pc(golden_dir):
    g = _load(golden_dir, "radar_peaks.npz")
    mask = g["mask"].astype(np.float32)
    rng = (np.float32(0.0596) * np.arange(mask.shape[2]).astype(np.float32))
    np.testing.assert_array_equal(R.mean_peaks_parallel_fast(mask * rng, False, 10.0), g["peaks_hard"])
    pcs = R.extract_pc(mask, 0.0596, g["az"], g["tm"], diff=False)
    for got, want in zip(pcs, (g["pc0"], g["pc1"])):
        assert got.shape == want.shape and want.shape[0] > 10
        np.testing.assert_allclose(got, want, rtol=0, atol=2e-5)
    pcT = R.extract_pc(mask, 0.0596, g["az"], g["tm"], T_ab=g["T_ab"], diff=False)
    for got, want in zip(pcT, (g["pcT0"], g["pcT1"])):
        np.testing.assert_allclose(got, want, rtol=0, atol=3e-5)
    pcs_soft = R.extract_pc(g["soft_mask"], 0.0596, g["az"], g["tm"], diff=True)
    for got, want in zip(pcs_soft, (g["pcs0"], g["pcs1"])):
        assert got.shape == want.shape
        np.testing.assert_allclose(got, want, rtol=0, atol=3e-5)
    kat_mask = np.zeros((1, 2, 1400), dtype=np.float32)
    kat_mask[0, 1, 500:504] = 1.0
    kat = R.extract_pc(kat_mask, 0.0596, np.array([[0.1, 0.3]], np.float32), np.zeros((1, 2), np.float32), diff=False)[0]
    np.testing.assert_allclose(kat, g["kat"], atol=2e-5)
    np.testing.assert_allclose(kat[0], [28.5544, 8.8329, 0.0], atol=1e-3)   # SURVEY.md §8a R4 KAT


def test_grids(golden_dir):
    # torch.linspace's vectorised fp32 evaluation is reproduced to ~1 ulp of the
    # 76 m half-width (7.6e-6 m), hence the absolute tolerances below.
    g = _load(golden_dir, "radar_grids.npz")
    rg, ag = R.form_cart_range_angle_grid()
    np.testing.assert_allclose(rg[::16, ::16], g["range_sub"], rtol=2e-6, atol=1e-5)
    np.testing.assert_allclose(ag[::16, ::16], g["angle_sub"], rtol=0, atol=5e-5)
    np.testing.assert_allclose(rg[319], g["range_row319"], rtol=2e-6, atol=1e-5)
    np.testing.assert_allclose(ag[0], g["angle_row0"], atol=5e-5)
    assert abs(rg.astype(np.float64).sum() - g["range_sum"]) < 1e-1
    rg5, ag5 = R.form_cart_range_angle_grid(0.5, 65)
    np.testing.assert_allclose(rg5, g["range_odd"], rtol=2e-6, atol=1e-5)
    np.testing.assert_allclose(ag5, g["angle_odd"], atol=5e-5)
    pr = R.form_polar_range_grid(0.0596)
    assert tuple(g["polar_shape"]) == pr.shape
    np.testing.assert_allclose(pr[0], g["polar_row"], rtol=3e-7)
    # SURVEY.md §8a R6 KATs
    assert abs(rg[0, 0] - 107.7189) < 1e-3 and abs(rg[319, 319] - 0.168576) < 1e-5 and abs(ag[0, 320] - 0.00156496) < 1e-6


def test_polar_to_cart(golden_dir):
    g = _load(golden_dir, "radar_polar2cart.npz")
    pol = (g["pol"] / np.float32(255.0)).astype(np.float32)
    kw = dict(cart_resolution=0.9536, cart_pixel_width=160)
    np.testing.assert_allclose(R.radar_polar_to_cartesian_diff(pol, g["az"], 0.5, **kw), g["cart"], atol=2e-5)
    np.testing.assert_allclose(R.radar_polar_to_cartesian_diff(pol, g["az"], 0.5, fix_wobble=False, **kw),
                               g["cart_nowob"], atol=2e-4)
    np.testing.assert_allclose(R.radar_polar_to_cartesian_diff(pol, g["az"], 0.5, interpolate_crossover=False, **kw),
                               g["cart_nocross"], atol=2e-5)
    rb = np.random.default_rng(int(g["seed_b"]))
    pol_b = (rb.integers(0, 256, size=(1, 400, 3360), dtype=np.uint8) / np.float32(255.0)).astype(np.float32)
    cart_b = R.radar_polar_to_cartesian_diff(pol_b, g["az_b"], 0.0596)
    # white-noise image sampled at range bins up to ~1800: one fp32 ulp of the bin
    # coordinate (1.2e-4) times the pixel-to-pixel contrast bounds the difference.
    np.testing.assert_allclose(cart_b[:, ::8, ::8], g["cart_b_sub"], atol=5e-4)
    np.testing.assert_allclose(cart_b[0, 200], g["cart_b_row"], atol=5e-4)
    assert abs(cart_b.astype(np.float64).sum() - g["cart_b_sum"]) < 0.5


def test_points_pixels_weights_bev(golden_dir):
    g = _load(golden_dir, "radar_points.npz")
    pts = g["pts"]
    np.testing.assert_allclose(R.point_to_cart_idx(pts), g["idx_plain"], rtol=1e-6, atol=1e-5)
    np.testing.assert_allclose(R.point_to_cart_idx(pts, min_to_plus_1=True), g["idx_norm"], rtol=1e-6, atol=1e-7)
    mask = np.random.default_rng(int(g["seed_mask"])).uniform(0, 1, size=(2, 640, 640)).astype(np.float32)
    w, dmn, mn, mean_w, max_w, min_w = R.extract_weights(mask, pts)
    np.testing.assert_allclose(w, g["weights"], atol=2e-5)
    np.testing.assert_allclose([dmn, mn, mean_w, max_w, min_w], g["stats"], rtol=1e-5, atol=2e-5)
    assert w[0, 50:].max() == 0.0 and w[0, 2] == 0.0          # fake + out-of-image points
    gm = R.extract_weights_grad_mask(mask.shape, pts, g["grad_w"])
    want = np.zeros_like(mask)
    i = g["grad_nz_idx"]
    want[i[0], i[1], i[2]] = g["grad_nz_val"]
    np.testing.assert_allclose(gm, want, atol=3e-5)
    bev = R.extract_bev_from_pts(g["bev_pts"])
    wb = np.zeros_like(bev)
    j = g["bev_nz_idx"]
    wb[j[0], j[1], j[2]] = 1.0
    np.testing.assert_array_equal(bev, wb)


def test_load_radar(golden_dir):
    g = _load(golden_dir, "radar_load.npz")
    fft, az, ts = R.load_radar(g["png"])
    np.testing.assert_array_equal(fft, g["fft"])
    np.testing.assert_array_equal(az, g["az"])
    np.testing.assert_array_equal(ts, g["ts"])


@pytest.mark.parametrize("tag", ["a", "b"])
def test_unet_forward_backward(golden_dir, tag):
    g = _load(golden_dir, "unet.npz")
    in_ch = 1 if tag == "a" else 3
    sd = unet_ref.init_state_dict(in_ch, 1234)
    names = [str(n) for n in g["names_" + tag]]
    assert names == list(sd.keys())
    np.testing.assert_allclose([sd[k].double().sum().item() for k in names], g["psum_" + tag], rtol=0, atol=1e-9)
    np.testing.assert_allclose([sd[k].double().abs().sum().item() for k in names], g["pabs_" + tag], rtol=1e-12)
    assert sum(v.numel() for v in sd.values()) == int(g["n_params_" + tag]) == (1769905 if tag == "a" else 1770049)
    for v in sd.values():
        v.requires_grad_(True)
    fft = torch.from_numpy(g["x_" + tag])
    if tag == "a":
        x = unet_ref.assemble_input(fft)
        m = unet_ref.unet_mask(x, sd)
    else:
        x = unet_ref.assemble_input(fft, torch.from_numpy(g["cfar_b"]), torch.from_numpy(g["range_b"]),
                                    log_transform=True, normalize=("standardize",))
        m = unet_ref.unet_mask(x, sd, leaky=True)
    np.testing.assert_allclose(m.detach().numpy(), g["mask_" + tag], atol=2e-6)
    (m * torch.from_numpy(g["gsel_" + tag])).sum().backward()
    gs = np.array([sd[k].grad.double().sum().item() for k in names])
    ga = np.array([sd[k].grad.double().abs().sum().item() for k in names])
    np.testing.assert_allclose(ga, g["gabs_" + tag], rtol=2e-3, atol=1e-6)
    assert np.all(np.abs(gs - g["gsum_" + tag]) <= 2e-3 * np.maximum(g["gabs_" + tag], 1e-3))


def test_losses(golden_dir):
    g = _load(golden_dir, "losses.npz")
    Tp, Tg = torch.from_numpy(g["T_pred"]), torch.from_numpy(g["T_gt"])
    np.testing.assert_allclose(train_ref.eval_validation_loss(Tp, Tg, True).numpy(), g["val_eye"], rtol=1e-6)
    np.testing.assert_allclose(train_ref.eval_validation_loss(Tp, Tg, False).numpy(), g["val_gt"], rtol=1e-5)
    mask = np.random.default_rng(int(g["seed_mask"])).uniform(0.01, 0.99, size=(5, 640, 640)).astype(np.float32)
    fft = np.random.default_rng(int(g["seed_fft"])).uniform(0, 1, size=(5, 640, 640)).astype(np.float32)
    cfar = (np.random.default_rng(int(g["seed_cfar"])).uniform(0, 1, size=(5, 640, 640)) > 0.95).astype(np.float32)
    keys = [str(k) for k in g["lw_keys"]]
    for tag, ge in (("a", True), ("b", False)):
        lw = dict(zip(keys, g["lw_" + tag]))
        Tpt = Tp.clone().requires_grad_(True)
        mt = torch.from_numpy(mask).requires_grad_(True)
        loss, comp = train_ref.eval_training_loss(Tpt, mt, torch.tensor(33.0), Tg, torch.from_numpy(fft),
                                                  torch.from_numpy(cfar), torch.from_numpy(g["pts"]),
                                                  torch.tensor(40.0), lw, gt_eye=ge)
        loss.backward()
        assert abs(loss.item() - float(g["loss_" + tag])) < 1e-5 * max(1.0, abs(float(g["loss_" + tag])))
        got = [float(comp[k]) for k in ("rot", "trans", "fft", "mask_pts", "cfar", "num_pts")]
        np.testing.assert_allclose(got, g["comp_" + tag], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(Tpt.grad.numpy(), g["gT_" + tag], atol=1e-6)
        assert abs(mt.grad.double().abs().sum().item() - float(g["gmask_abs_" + tag])) < 1e-4 * float(g["gmask_abs_" + tag])


# ----------------------------------------------------------------------------- polar-domain network
@pytest.mark.parametrize("tag", ["p", "q"])
def test_unet_polar_sizes(golden_dir, tag):
    """network_input_type='polar': non-square input whose sizes go odd under the poolings
    (50 x 84 -> ... -> 1 x 2); fixtures from tests/golden/make_golden_polar.py."""
    g = _load(golden_dir, "polar_net.npz")
    in_ch = 1 if tag == "p" else 2
    sd = unet_ref.init_state_dict(in_ch, 1234)
    names = [str(n) for n in g["names_" + tag]]
    assert names == list(sd.keys())
    for v in sd.values():
        v.requires_grad_(True)
    fft = torch.from_numpy(g["x_" + tag])
    rng_mask = torch.from_numpy(g["range_q"]) if tag == "q" else None
    m = unet_ref.unet_mask(unet_ref.assemble_input(fft, None, rng_mask), sd)
    np.testing.assert_allclose(m.detach().numpy(), g["mask_" + tag], atol=2e-6)
    (m * torch.from_numpy(g["gsel_" + tag])).sum().backward()
    gs = np.array([sd[k].grad.double().sum().item() for k in names])
    ga = np.array([sd[k].grad.double().abs().sum().item() for k in names])
    np.testing.assert_allclose(ga, g["gabs_" + tag], rtol=2e-3, atol=1e-6)
    assert np.all(np.abs(gs - g["gsum_" + tag]) <= 2e-3 * np.maximum(g["gabs_" + tag], 1e-3))
    if tag == "q":      # R7: the polar range grid the reference stacks as a channel
        np.testing.assert_allclose(R.form_polar_range_grid(0.0596)[:50, :84], g["range_q"], rtol=1e-6)


@pytest.mark.parametrize("tag", ["s", "f"])
def test_extract_weights_non_square_mask(golden_dir, tag):
    """The reference normalises points by the 640-pixel Cartesian width and lets grid_sample stretch
    [-1,1] over whatever mask it is given: (40,96) and the polar (400,3360)."""
    g = _load(golden_dir, "polar_net.npz")
    B, H, W = (int(v) for v in g["ew_shape_" + tag])
    mask = np.random.default_rng(int(g["ew_seed_" + tag])).uniform(0, 1, size=(B, H, W)).astype(np.float32)
    pts = g["ew_pts_" + tag]
    w, dmn, mn, mean_w, max_w, min_w = R.extract_weights(mask, pts)
    np.testing.assert_allclose(w, g["ew_w_" + tag], atol=2e-5)
    np.testing.assert_allclose([dmn, mn, mean_w, max_w, min_w], g["ew_stats_" + tag], rtol=1e-5, atol=2e-5)
    gm = R.extract_weights_grad_mask(mask.shape, pts, g["ew_gw_" + tag])
    want = np.zeros_like(mask)
    i = g["ew_gidx_" + tag]
    want[i[0], i[1], i[2]] = g["ew_gval_" + tag]
    np.testing.assert_allclose(gm, want, atol=3e-5)
