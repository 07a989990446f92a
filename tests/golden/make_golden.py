"""Generate the golden vectors under tests/golden/ by IMPORTING the reference's
own Python modules from /root/reference (build container only; the reference
never travels to the GPU box — only the .npz data written here does).

Run:  python tests/golden/make_golden.py

Stubs are installed for third-party imports that are absent here and unused by
the functions exercised (cv2, neptune, dICP, matplotlib is real).  The stubbed
``dICP.ICP.ICP`` only carries ``target_pad_val``; nothing of the ICP arithmetic
is (or can be) captured — that boundary stays "parity unpinned".

Every fixture stores inputs (or the seed + recipe that regenerates them with
numpy alone) and the reference's outputs; nothing of the reference's source.
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/mm_masking"
OUT = os.path.dirname(os.path.abspath(__file__))


def _install_stubs():
    cv2 = types.ModuleType("cv2")
    sys.modules["cv2"] = cv2
    nep = types.ModuleType("neptune")
    nept = types.ModuleType("neptune.types")
    nept.File = object
    nepu = types.ModuleType("neptune.utils")
    nepu.stringify_unsupported = lambda x: x
    nep.types = nept
    nep.utils = nepu
    sys.modules.update({"neptune": nep, "neptune.types": nept, "neptune.utils": nepu})
    npt = types.ModuleType("neptune_pytorch")
    npt.NeptuneLogger = object
    sys.modules["neptune_pytorch"] = npt
    plg = types.ModuleType("pylgmath")
    plg.se3op = object
    plg.Transformation = object
    sys.modules["pylgmath"] = plg
    d = types.ModuleType("dICP")
    di = types.ModuleType("dICP.ICP")

    class ICP:
        def __init__(self, *a, **k):
            self.target_pad_val = 1000.0

    di.ICP = ICP
    d.ICP = di
    sys.modules.update({"dICP": d, "dICP.ICP": di})
    ds = types.ModuleType("icp_weight_dataset")
    ds.ICPWeightDataset = object
    sys.modules["icp_weight_dataset"] = ds


def speckle_scan(rng, B, A, R, n_blobs=6):
    """Synthetic polar power image: Rayleigh speckle + planted gaussian blobs."""
    img = rng.rayleigh(0.04, size=(B, A, R)).astype(np.float32)
    cols = np.arange(R, dtype=np.float32)
    for b in range(B):
        for a in range(A):
            for _ in range(n_blobs):
                c = rng.uniform(100, min(R, 1280) - 10)
                amp = rng.uniform(0.4, 0.9)
                img[b, a] += (amp * np.exp(-0.5 * ((cols - c) / 2.0) ** 2)).astype(np.float32)
    return np.clip(img, 0, 1).astype(np.float32)


def main():
    _install_stubs()
    sys.path.insert(0, REF)
    import radar_utils as ru
    torch.manual_seed(0)
    rng = np.random.default_rng(20241022)

    # ------------------------------------------------------------------ R2 cfar_mask
    raw = speckle_scan(rng, 2, 8, 1400)
    t = torch.from_numpy(raw)
    hard = ru.cfar_mask(t, 0.0596, a_thresh=1.0, b_thresh=0.09, diff=False).numpy()
    soft = ru.cfar_mask(t, 0.0596, a_thresh=1.0, b_thresh=0.09, diff=True).numpy()
    raw2 = speckle_scan(rng, 1, 4, 160, n_blobs=2)
    kw2 = dict(width=20, minr=1.0, maxr=25.0, guard=2, a_thresh=1.5, b_thresh=0.05)
    hard2 = ru.cfar_mask(torch.from_numpy(raw2), 0.2, diff=False, **kw2).numpy()
    soft2 = ru.cfar_mask(torch.from_numpy(raw2), 0.2, diff=True, steep_fact=7.0, **kw2).numpy()
    np.savez_compressed(os.path.join(OUT, "radar_cfar.npz"), raw=raw, hard=hard.astype(np.uint8), soft=soft,
                        raw2=raw2, hard2=hard2.astype(np.uint8), soft2=soft2,
                        kw2_keys=np.array(list(kw2.keys())), kw2_vals=np.array(list(kw2.values()), dtype=np.float64))

    # ------------------------------------------------------------------ R3/R4 peaks + extract_pc
    A, R = 8, 1400
    az = (np.round(np.arange(A) * 5600 / A + rng.uniform(-0.2, 0.2, A)) * (2 * np.pi / 5600)).astype(np.float32)
    az = np.stack([az, np.roll(az, 0) + np.float32(0.01)]).astype(np.float32)
    tm = (np.arange(A, dtype=np.float32)[None] * 625.0 + np.array([[0.0], [1e6]], dtype=np.float32)).astype(np.float32)
    peaks_hard = ru.mean_peaks_parallel_fast(torch.from_numpy(hard * (0.0596 * np.arange(R, dtype=np.float32))),
                                             diff=False, steep_fact=10.0).numpy()
    pcs = ru.extract_pc(torch.from_numpy(hard), 0.0596, torch.from_numpy(az), torch.from_numpy(tm), diff=False)
    T_ab = np.tile(np.eye(4, dtype=np.float32), (2, 1, 1))
    T_ab[0, :2, :2] = [[np.cos(0.3), -np.sin(0.3)], [np.sin(0.3), np.cos(0.3)]]
    T_ab[0, :3, 3] = [1.5, -2.0, 0.25]
    T_ab[1, :3, 3] = [-4.0, 0.5, 0.0]
    pcs_T = ru.extract_pc(torch.from_numpy(hard), 0.0596, torch.from_numpy(az), torch.from_numpy(tm),
                          T_ab=torch.from_numpy(T_ab), diff=False)
    pcs_soft = ru.extract_pc(torch.from_numpy(soft), 0.0596, torch.from_numpy(az), torch.from_numpy(tm), diff=True)
    # the survey's KAT: blob cols 500..503 at azimuth 0.3 rad
    kat_mask = np.zeros((1, 2, 1400), dtype=np.float32)
    kat_mask[0, 1, 500:504] = 1.0
    kat = ru.extract_pc(torch.from_numpy(kat_mask), 0.0596, torch.tensor([[0.1, 0.3]]), torch.zeros(1, 2), diff=False)
    np.savez_compressed(os.path.join(OUT, "radar_peaks.npz"), mask=hard.astype(np.uint8), soft_mask=soft, az=az, tm=tm,
                        peaks_hard=peaks_hard, pc0=pcs[0].numpy(), pc1=pcs[1].numpy(), T_ab=T_ab,
                        pcT0=pcs_T[0].numpy(), pcT1=pcs_T[1].numpy(),
                        pcs0=pcs_soft[0].numpy(), pcs1=pcs_soft[1].numpy(), kat=kat[0].numpy())

    # ------------------------------------------------------------------ R6/R7 grids
    rg, ag = ru.form_cart_range_angle_grid()
    rg5, ag5 = ru.form_cart_range_angle_grid(cart_resolution=0.5, cart_pixel_width=65)
    pr = ru.form_polar_range_grid(polar_resolution=0.0596)
    np.savez_compressed(os.path.join(OUT, "radar_grids.npz"),
                        range_sub=rg.numpy()[::16, ::16], angle_sub=ag.numpy()[::16, ::16],
                        range_row319=rg.numpy()[319], angle_row0=ag.numpy()[0],
                        range_sum=np.float64(rg.double().sum().item()), angle_sum=np.float64(ag.double().sum().item()),
                        range_odd=rg5.numpy(), angle_odd=ag5.numpy(),
                        polar_row=pr.numpy()[0], polar_shape=np.array(pr.shape))

    # ------------------------------------------------------------------ R5 polar -> cartesian
    # (a) small, fully stored, non-default parameters
    A5, R5 = 100, 600
    pol = rng.integers(0, 256, size=(2, A5, R5), dtype=np.uint8)
    az5 = np.sort((np.arange(A5) * 2 * np.pi / A5 + rng.uniform(-0.01, 0.01, A5)).astype(np.float32))
    az5 = np.stack([az5, np.sort((np.arange(A5) * 2 * np.pi / A5 + 0.02).astype(np.float32))])
    pol_f = (pol / np.float32(255.0)).astype(np.float32)
    cart_a = ru.radar_polar_to_cartesian_diff(torch.from_numpy(pol_f), torch.from_numpy(az5), 0.5,
                                              cart_resolution=0.9536, cart_pixel_width=160).numpy()
    cart_a_nowob = ru.radar_polar_to_cartesian_diff(torch.from_numpy(pol_f), torch.from_numpy(az5), 0.5,
                                                    cart_resolution=0.9536, cart_pixel_width=160,
                                                    fix_wobble=False).numpy()
    cart_a_nocross = ru.radar_polar_to_cartesian_diff(torch.from_numpy(pol_f), torch.from_numpy(az5), 0.5,
                                                      cart_resolution=0.9536, cart_pixel_width=160,
                                                      interpolate_crossover=False).numpy()
    # (b) full BASELINE shape 400x3360 -> 640x640; input regenerated from a seed, output sub-sampled
    seed_b = 777
    rb = np.random.default_rng(seed_b)
    pol_b = (rb.integers(0, 256, size=(1, 400, 3360), dtype=np.uint8) / np.float32(255.0)).astype(np.float32)
    enc = np.round(np.arange(400) * 14.0 + rb.uniform(-0.2, 0.2, 400) * 0)  # exact encoder counts
    az_b = ((enc + rb.integers(-1, 2, 400)) * (2 * np.pi / 5600)).astype(np.float32)[None]
    az_b = np.sort(az_b, axis=1)
    cart_b = ru.radar_polar_to_cartesian_diff(torch.from_numpy(pol_b), torch.from_numpy(az_b), 0.0596).numpy()
    np.savez_compressed(os.path.join(OUT, "radar_polar2cart.npz"), pol=pol, az=az5, cart=cart_a,
                        cart_nowob=cart_a_nowob, cart_nocross=cart_a_nocross,
                        seed_b=np.int64(seed_b), az_b=az_b, cart_b_sub=cart_b[:, ::8, ::8],
                        cart_b_row=cart_b[0, 200], cart_b_sum=np.float64(cart_b.astype(np.float64).sum()))

    # ------------------------------------------------------------------ R8/R9/R10 points <-> pixels
    pts = rng.uniform(-80, 80, size=(2, 64, 3)).astype(np.float32)
    pts[:, 50:, :] = 0.0                      # fake (padded) points
    pts[0, 0] = [0.1192, 0.1192, 0.0]         # pixel centre (row 319, col 320)
    pts[0, 1] = [0.0, 5.0, 0.0]               # x == 0 only: still a real point
    pts[0, 2] = [200.0, 3.0, 0.0]             # out of the image
    pts[1, 0] = [76.1688, -76.1688, 0.0]      # corner pixel (0,0)
    idx_a = ru.point_to_cart_idx(torch.from_numpy(pts)).numpy()
    idx_b = ru.point_to_cart_idx(torch.from_numpy(pts), min_to_plus_1=True).numpy()
    seed_m = 4242
    mask = np.random.default_rng(seed_m).uniform(0, 1, size=(2, 640, 640)).astype(np.float32)
    mt = torch.from_numpy(mask).requires_grad_(True)
    w, dmn, mn, mean_w, max_w, min_w = ru.extract_weights(mt, torch.from_numpy(pts))
    gw = np.random.default_rng(5).normal(size=(2, 64)).astype(np.float32)
    (w * torch.from_numpy(gw)).sum().backward()
    g = mt.grad.numpy()
    nzb, nzr, nzc = np.nonzero(g)
    bev_pts = rng.uniform(-90, 90, size=(2, 300, 6)).astype(np.float32)
    bev_pts[:, 250:, :] = 1000.0              # target_pad_val rows
    bev_pts[0, 0, :2] = [0.05, -0.05]         # lands on the centre pixel neighbourhood
    bev = ru.extract_bev_from_pts(torch.from_numpy(bev_pts)).numpy()
    bb, br, bc = np.nonzero(bev)
    np.savez_compressed(os.path.join(OUT, "radar_points.npz"), pts=pts, idx_plain=idx_a, idx_norm=idx_b,
                        seed_mask=np.int64(seed_m), weights=w.detach().numpy(),
                        stats=np.array([dmn.item(), mn.item(), mean_w.item(), max_w.item(), min_w.item()], dtype=np.float64),
                        grad_w=gw, grad_nz_idx=np.stack([nzb, nzr, nzc]).astype(np.int32), grad_nz_val=g[nzb, nzr, nzc],
                        bev_pts=bev_pts, bev_nz_idx=np.stack([bb, br, bc]).astype(np.int32))

    # ------------------------------------------------------------------ R1 load_radar
    png = rng.integers(0, 256, size=(6, 11 + 40), dtype=np.uint8)
    fft, azs, tss = ru.load_radar(png)
    np.savez_compressed(os.path.join(OUT, "radar_load.npz"), png=png, fft=fft, az=azs, ts=tss)

    # ------------------------------------------------------------------ U-Net (U1-U5) + losses (T2-T3)
    import icp_weight_policy as pol_mod
    import train_icp_weights as trn
    params = {
        "icp_type": "pt2pt", "fft_input": True, "cfar_input": False, "range_input": False,
        "network_input_type": "cartesian", "network_output_type": "cartesian", "leaky": False,
        "dropout": 0.0, "batch_norm": False, "float_type": torch.float32, "device": torch.device("cpu"),
        "init_weights": True, "normalize": ["minmax"], "log_transform": False, "a_thresh": 1.0,
        "b_thresh": 0.09, "gt_eye": True, "max_iter": 10, "loss_icp_rot_weight": 1.0,
        "loss_icp_trans_weight": 1.0, "norm_weights": True,
    }
    unet = {}
    for tag, over in (("a", {}), ("b", {"cfar_input": True, "range_input": True, "leaky": True,
                                         "normalize": ["standardize"], "log_transform": True})):
        p = dict(params)
        p.update(over)
        torch.manual_seed(1234)
        model = pol_mod.LearnICPWeightPolicy(p)
        model.train()
        sd = model.state_dict()
        names = list(sd.keys())
        unet["names_" + tag] = np.array(names)
        unet["shapes_" + tag] = np.array([str(tuple(sd[k].shape)) for k in names])
        unet["psum_" + tag] = np.array([sd[k].double().sum().item() for k in names])
        unet["pabs_" + tag] = np.array([sd[k].double().abs().sum().item() for k in names])
        H = 64
        xin = np.random.default_rng(99).uniform(0.01, 1, size=(2, H, H)).astype(np.float32)
        xcf = (np.random.default_rng(98).uniform(0, 1, size=(2, H, H)) > 0.9).astype(np.float32)
        if p["range_input"]:
            model.range_mask = model.range_mask[:H, :H].contiguous()
        scan = {"fft_data": torch.from_numpy(xin.copy()), "fft_cfar": torch.from_numpy(xcf.copy()),
                "raw_pc": torch.zeros(2, 4, 3)}
        m = model(scan, {"pc": torch.zeros(2, 4, 6)}, torch.eye(4).repeat(2, 1, 1), mask_only=True)
        gsel = torch.from_numpy(np.random.default_rng(97).normal(size=(2, H, H)).astype(np.float32))
        (m * gsel).sum().backward()
        unet["x_" + tag] = xin
        unet["cfar_" + tag] = xcf
        unet["gsel_" + tag] = gsel.numpy()
        unet["mask_" + tag] = m.detach().numpy()
        unet["gsum_" + tag] = np.array([dict(model.named_parameters())[k].grad.double().sum().item() for k in names])
        unet["gabs_" + tag] = np.array([dict(model.named_parameters())[k].grad.double().abs().sum().item() for k in names])
        if tag == "b":
            unet["range_b"] = model.range_mask.numpy()
        unet["n_params_" + tag] = np.int64(sum(v.numel() for v in sd.values()))
    np.savez_compressed(os.path.join(OUT, "unet.npz"), **unet)

    # losses
    Bq = 5
    rl = np.random.default_rng(11)
    Tp = np.tile(np.eye(4, dtype=np.float32), (Bq, 1, 1))
    th = rl.uniform(-0.3, 0.3, Bq)
    Tp[:, 0, 0], Tp[:, 0, 1], Tp[:, 1, 0], Tp[:, 1, 1] = np.cos(th), -np.sin(th), np.sin(th), np.cos(th)
    Tp[:, :2, 3] = rl.uniform(-1, 1, (Bq, 2))
    Tg = np.tile(np.eye(4, dtype=np.float32), (Bq, 1, 1))
    th2 = rl.uniform(-0.3, 0.3, Bq)
    Tg[:, 0, 0], Tg[:, 0, 1], Tg[:, 1, 0], Tg[:, 1, 1] = np.cos(th2), -np.sin(th2), np.sin(th2), np.cos(th2)
    Tg[:, :2, 3] = rl.uniform(-1, 1, (Bq, 2))
    val_eye = trn.eval_validation_loss(torch.from_numpy(Tp), torch.from_numpy(Tg), gt_eye=True).numpy()
    val_gt = trn.eval_validation_loss(torch.from_numpy(Tp), torch.from_numpy(Tg), gt_eye=False).numpy()
    lmask = np.random.default_rng(12).uniform(0.01, 0.99, size=(Bq, 640, 640)).astype(np.float32)
    lpts = rl.uniform(-70, 70, size=(Bq, 100, 6)).astype(np.float32)
    lfft = np.random.default_rng(13).uniform(0, 1, size=(Bq, 640, 640)).astype(np.float32)
    lcfar = (np.random.default_rng(14).uniform(0, 1, size=(Bq, 640, 640)) > 0.95).astype(np.float32)

    class _M:
        mean_all_pts = torch.tensor(40.0)

    lw_a = {"icp_rot": 1.0, "icp_trans": 1.0, "fft": 0.0, "mask_pts": 1.0, "cfar": 0.0, "num_pts": 0.0}
    lw_b = {"icp_rot": 0.5, "icp_trans": 2.0, "fft": 0.3, "mask_pts": 0.7, "cfar": 0.2, "num_pts": 0.01}
    outs = {}
    for tag, lw, ge in (("a", lw_a, True), ("b", lw_b, False)):
        Tpt = torch.from_numpy(Tp).requires_grad_(True)
        mt2 = torch.from_numpy(lmask).requires_grad_(True)
        loss, comp = trn.eval_training_loss(Tpt, mt2, torch.tensor(33.0), torch.from_numpy(Tg),
                                            {"fft_data": torch.from_numpy(lfft), "fft_cfar": torch.from_numpy(lcfar)},
                                            {"pc": torch.from_numpy(lpts)}, _M(), loss_weights=lw, gt_eye=ge, epoch=0)
        loss.backward()
        outs["loss_" + tag] = np.float64(loss.item())
        outs["comp_" + tag] = np.array([float(comp[k]) for k in ("rot", "trans", "fft", "mask_pts", "cfar", "num_pts")])
        outs["gT_" + tag] = Tpt.grad.numpy()
        outs["gmask_sum_" + tag] = np.float64(mt2.grad.double().sum().item())
        outs["gmask_abs_" + tag] = np.float64(mt2.grad.double().abs().sum().item())
    np.savez_compressed(os.path.join(OUT, "losses.npz"), T_pred=Tp, T_gt=Tg, val_eye=val_eye, val_gt=val_gt,
                        seed_mask=np.int64(12), seed_fft=np.int64(13), seed_cfar=np.int64(14), pts=lpts,
                        lw_keys=np.array(list(lw_a.keys())), lw_a=np.array(list(lw_a.values())),
                        lw_b=np.array(list(lw_b.values())), **outs)
    # ------------------------------------------------------------------ dataset-side tensor ops
    # icp_weight_dataset.py needs pyboreas / vtr_pose_graph / vtr_utils / ROS to import; its
    # augment_data and filter_map methods only use torch, so they are called unbound on a stand-in self.
    for name in ("pyboreas", "pyboreas.utils", "pyboreas.utils.odometry", "pyboreas.utils.utils", "vtr_pose_graph",
                 "vtr_pose_graph.graph_utils", "vtr_pose_graph.graph_iterators", "vtr_utils",
                 "vtr_utils.bag_file_parsing", "utils", "utils.extract_graph"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["pyboreas.utils.odometry"].read_traj_file2 = None
    sys.modules["pyboreas.utils.odometry"].read_traj_file_gt2 = None
    for n in ("SE3Tose3", "get_closest_index", "get_inverse_tf", "rotToRollPitchYaw"):
        setattr(sys.modules["pyboreas.utils.utils"], n, None)
    sys.modules["vtr_utils.bag_file_parsing"].Rosbag2GraphFactory = None
    sys.modules["vtr_pose_graph.graph_iterators"].TemporalIterator = None
    sys.modules["utils.extract_graph"].extract_points_and_map = None
    del sys.modules["icp_weight_dataset"]
    import icp_weight_dataset as ds_mod
    DS = ds_mod.ICPWeightDataset
    me = types.SimpleNamespace(gt_eye=True, float_type=torch.float32, loc_sensor="radar", map_sensor="lidar")
    rd = np.random.default_rng(21)
    scan_raw = torch.from_numpy(rd.uniform(-60, 60, (40, 3)).astype(np.float32))
    scan_filt = scan_raw.clone() + 0.01
    map6 = torch.from_numpy(rd.uniform(-60, 60, (50, 6)).astype(np.float32))
    az = torch.from_numpy(np.sort(rd.uniform(0, 2 * np.pi, 16)).astype(np.float32))
    fft = torch.from_numpy(rd.uniform(0, 1, (16, 24)).astype(np.float32))
    cf = (fft > 0.8).float()
    torch.manual_seed(4321)
    outs_aug = DS.augment_data(me, scan_raw.clone(), scan_filt.clone(), map6.clone(), az.clone(), fft.clone(), cf.clone())
    torch.manual_seed(4321)
    angle = (2 * np.pi * torch.rand(1, dtype=torch.float32)).item()
    pts = torch.from_numpy(rd.uniform(-30, 30, (60, 3)).astype(np.float32))
    pts[:, 2] = torch.from_numpy(rd.uniform(-2, 2, 60).astype(np.float32))
    nrm = torch.from_numpy(rd.normal(size=(60, 3)).astype(np.float32))
    nrm = nrm / nrm.norm(dim=1, keepdim=True)
    Tgt = torch.eye(4)
    Tgt[:2, :2] = torch.tensor([[np.cos(0.2), -np.sin(0.2)], [np.sin(0.2), np.cos(0.2)]])
    Tgt[:3, 3] = torch.tensor([1.0, -2.0, 0.1])
    fa_p, fa_n = DS.filter_map(me, pts, nrm, Tgt, return_aligned=True)
    fb_p, fb_n = DS.filter_map(me, pts, nrm, Tgt, return_aligned=False)
    np.savez_compressed(os.path.join(OUT, "dataset_ops.npz"), scan_raw=scan_raw.numpy(), scan_filt=scan_filt.numpy(),
                        map6=map6.numpy(), az=az.numpy(), fft=fft.numpy(), cfar=cf.numpy(), angle=np.float64(angle),
                        aug_raw=outs_aug[0].numpy(), aug_filt=outs_aug[1].numpy(), aug_map=outs_aug[2].numpy(),
                        aug_az=outs_aug[3].numpy(), aug_fft=outs_aug[4].numpy(), aug_cfar=outs_aug[5].numpy(),
                        pts=pts.numpy(), nrm=nrm.numpy(), T_gt=Tgt.numpy(), fa_p=fa_p.numpy(), fa_n=fa_n.numpy(),
                        fb_p=fb_p.numpy(), fb_n=fb_n.numpy())

    print("golden vectors written to", OUT)
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print("  %-28s %8.1f KB" % (f, os.path.getsize(os.path.join(OUT, f)) / 1024))


if __name__ == "__main__":
    main()
