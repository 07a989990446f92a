"""Parity of the HIP radar operators (through the C ABI, via the radar_utils mirror)
with the golden vectors captured from the reference and with the numpy oracle."""
import os

import numpy as np
import pytest
import torch

from mm_masking_amd import radar_utils as ru
from mm_masking_amd import synthetic
from oracle import radar_ref
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


def test_cfar_long_rows_take_the_one_row_per_block_kernel():
    """Rows whose prefix sums do not fit 40 KB of LDS (more than 3 412 range bins) run on cfar_mask_kernel, one row per block,
    instead of the persistent cfar_mask_rows_kernel: same masks as the oracle there too, and as the persistent kernel's on
    the columns both see."""
    rng = np.random.default_rng(11)
    raw = rng.random((2, 5, 4000), dtype=np.float32) * 0.2
    raw[:, :, 700:3900:97] += 0.6
    got = ru.cfar_mask(_g(raw), 0.0596, diff=False).cpu().numpy()
    want = R.cfar_mask(raw, 0.0596, diff=False)
    _mask_close(raw, got, want, R.cfar_threshold(raw, 0.0596))
    assert 20 < want.sum() < 5000
    soft = ru.cfar_mask(_g(raw), 0.0596, diff=True).cpu().numpy()
    d = np.abs(soft - R.cfar_mask(raw, 0.0596, diff=True))
    assert (d > 0.5).mean() < 1e-3 and d[d <= 0.5].max() < 2e-5


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


def test_cart_to_polar_golden_and_full_size(golden_dir):
    g = np.load(os.path.join(golden_dir, "cart2polar.npz"), allow_pickle=False)
    out = ru.radar_cartesian_to_polar(torch.from_numpy(g["cart"]).to(DEV), torch.from_numpy(g["az"]).to(DEV), 0.0596,
                                      polar_pixel_shape=g["polar"].shape[1:])
    assert out.dtype == torch.float64 and out.is_cuda
    # The golden vectors were made on another host: torch's CPU sin / cos (SLEEF, dispatched per CPU model) differ
    # in the last bit between hosts, for the reference as for us -> a few ulp here; bit-exact against the oracle
    # evaluated on THIS host (below), which tests/test_loader_cpu.py pins bit-exactly to the reference.
    np.testing.assert_allclose(out.cpu().numpy(), g["polar"], rtol=0, atol=1e-12)
    assert np.array_equal(out.cpu().numpy(), radar_ref.radar_cartesian_to_polar(g["cart"], g["az"], 0.0596,
                                                                                polar_pixel_shape=g["polar"].shape[1:]))
    out2 = ru.radar_cartesian_to_polar(torch.from_numpy(g["cart2"]), torch.from_numpy(g["az2"]), 0.1, cart_resolution=0.3,
                                       polar_pixel_shape=(16, 120))
    assert not out2.is_cuda                                                            # CPU in -> CPU out, as every operator
    np.testing.assert_allclose(out2.numpy(), g["polar2"], rtol=0, atol=1e-12)
    assert np.array_equal(out2.numpy(), radar_ref.radar_cartesian_to_polar(g["cart2"], g["az2"], 0.1, cart_resolution=0.3,
                                                                           polar_pixel_shape=(16, 120)))
    with pytest.raises(RuntimeError, match=str(g["fp32_error"])):
        ru.radar_cartesian_to_polar(torch.zeros(1, 8, 8, device=DEV), torch.zeros(1, 4, device=DEV), 0.0596, polar_pixel_shape=(4, 10))
    # full size: 640 x 640 -> 400 x 3360, against the oracle
    rng = np.random.default_rng(8)
    cart = rng.random((2, 640, 640))
    az = np.sort(rng.uniform(0, 2 * np.pi, (2, 400)), axis=1)
    got = ru.radar_cartesian_to_polar(torch.from_numpy(cart).to(DEV), torch.from_numpy(az).to(DEV), 0.0596).cpu().numpy()
    assert got.shape == (2, 400, 3360)
    assert np.array_equal(got, radar_ref.radar_cartesian_to_polar(cart, az, 0.0596))
    # round trip polar -> Cartesian -> polar reproduces a smooth image inside the Cartesian footprint
    A, R = 400, 3360
    rr, aa = np.meshgrid(np.arange(R) * 0.0596, np.arange(A) * 2 * np.pi / A)
    pol = (0.5 + 0.4 * np.sin(rr / 9.0) * np.cos(3 * aa)).astype(np.float32)[None]
    azf = (np.arange(A) * 2 * np.pi / A).astype(np.float32)[None]
    c = ru.radar_polar_to_cartesian_diff(torch.from_numpy(pol).to(DEV), torch.from_numpy(azf).to(DEV), 0.0596)
    back = ru.radar_cartesian_to_polar(c.double(), torch.from_numpy(azf).to(DEV).double(), 0.0596).cpu().numpy()
    inner = slice(40, 1200)                    # ranges well inside the 76 m half-width of the 640-pixel image
    assert np.abs(back[0, :, inner] - pol[0, :, inner]).max() < 0.05


def test_mask_gradient_scatter_is_ordered_and_reproducible():
    """mmk_sample_weights_bwd: many taps on the same pixels (points 1 cm apart) -- the sums equal a sequential loop over
    (point, tap) in fp32, bit for bit, and do not change from run to run."""
    from mm_masking_amd import radar_utils as ru
    g = torch.Generator().manual_seed(3)
    B, N, H = 2, 4096, 64
    pc = torch.zeros(B, N, 3)
    pc[:, :3000, :2] = (torch.rand(B, 3000, 2, generator=g) - 0.5) * 6.0      # 3 000 real points inside ~25 x 25 pixels of a 64 x 64 mask
    pc[:, 100:110] = 0.0                                                      # fake rows in between
    pc[:, 200, :2] = torch.tensor([500.0, 500.0])                            # out of the image
    gw = torch.randn(B, N, generator=g)
    mask = torch.rand(B, H, H, generator=g).to(DEV).requires_grad_(True)
    outs = []
    for rep in range(3):
        mask.grad = None
        w = ru._SampleWeights.apply(mask, pc.to(DEV), 0.2384, H)
        (w * gw.to(DEV)).sum().backward()
        outs.append(mask.grad.clone())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    # sequential fp32 loop, same tap arithmetic as the kernel (csrc/mmk_radar.hip: weight_taps)
    ref = np.zeros((B, H, H), np.float32)
    f = np.float32
    for b in range(B):
        for n in range(N):
            x, y = f(pc[b, n, 0]), f(pc[b, n, 1])
            if x == 0 and y == 0:
                continue
            gx = f(f(f(y / f(0.2384)) / f(H - 1)) * f(2.0))
            gy = f(f(f(-x / f(0.2384)) / f(H - 1)) * f(2.0))
            ix = f(f(f(gx + f(1)) / f(2)) * f(H - 1))
            iy = f(f(f(gy + f(1)) / f(2)) * f(H - 1))
            x0, y0 = np.floor(ix), np.floor(iy)
            wx, wy = f(ix - x0), f(iy - y0)
            taps = [(0, 0, f(f(1 - wy) * f(1 - wx))), (0, 1, f(f(1 - wy) * wx)), (1, 0, f(wy * f(1 - wx))), (1, 1, f(wy * wx))]
            for dy, dx, wt in taps:
                yy, xx = int(y0) + dy, int(x0) + dx
                if 0 <= yy < H and 0 <= xx < H:
                    ref[b, yy, xx] = f(ref[b, yy, xx] + f(f(gw[b, n]) * wt))
    assert np.array_equal(outs[0].cpu().numpy(), ref), float(np.abs(outs[0].cpu().numpy() - ref).max())
