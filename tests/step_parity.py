"""Shared body of the whole-step parity tests (tests/test_gpu_policy.py, tests/test_gpu_step_parity.py): one
training step of the product's default path -- hand-written bf16 U-Net -> extract_weights -> dICP -> loss ->
backward, dropout 0 -- against oracle/train_ref.py (fp32 CPU port of
/root/reference/mm_masking/train_icp_weights.py:22-69 + icp_weight_policy.py:127-275).

Two comparisons, because a bf16 network cannot reproduce an fp32 mask bit for bit:
  (1) U-Net alone: HIP mask vs the oracle's fp32 mask (stated bf16 tolerance), and the parameter gradients of
      the whole step vs the oracle's (global relative L2 and per-tensor cosine: the bf16 budget);
  (2) everything downstream of the mask with the HIP mask fed to the oracle: correspondence indices bit-exact
      over every iteration, pose within north_star's 1e-3 m / 1e-4 rad, the same loss, and d loss / d mask
      (the dICP backward + BCE + bilinear scatter) within 2e-3 of autograd through the oracle.
"""
import numpy as np
import torch

from mm_masking_amd import train_icp_weights as trn
from mm_masking_amd.icp_weight_policy import LearnICPWeightPolicy
from oracle import train_ref, unet_ref


def run(raw, params, batch, max_iter, seed=1234, backward=True, loss_fn=None, dim=2):
    dev = raw["T_init"].device
    loss_fn = loss_fn or {"name": "huber", "metric": 1.0}
    params = dict(params, dropout=0.0, max_iter=max_iter, icp_type="pt2pl", icp_loss_fn=loss_fn, icp_dim=dim)
    torch.manual_seed(seed)
    model = LearnICPWeightPolicy(params).to(dev)
    assert model.unet_backend == "hip"
    model.train()
    lw = trn.loss_weights_from(params)
    res = {}
    # ---------------- product (HIP)
    ctx = torch.enable_grad() if backward else torch.no_grad()
    with ctx:
        T, mask, nn0 = model(batch["loc_data"], batch["map_data"], raw["T_init"])
        if backward:
            mask.retain_grad()
            idx_hip = T.grad_fn.saved_tensors[3].cpu().numpy()          # (K,B,N) per-iteration correspondences
            loss, _ = trn.eval_training_loss(T, mask, nn0, raw["T_gt"], batch["loc_data"], batch["map_data"], model,
                                             loss_weights=lw)
            loss.backward()
    if not backward:
        idx_hip = model.ICP_alg.last_state["idx"].cpu().numpy()         # (1,B,N): the last iteration
    # ---------------- oracle, its own fp32 mask
    sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    cb = {"fft_data": batch["loc_data"]["fft_data"].cpu(), "raw_pc": batch["loc_data"]["raw_pc"].cpu(),
          "filtered_pc": batch["loc_data"]["filtered_pc"].cpu(), "map_pc": raw["map_pc"].cpu(),
          "T_init": raw["T_init"].cpu(), "T_gt": raw["T_gt"].cpu()}
    ref = train_ref.TrainStepRef(icp_type="pt2pl", loss_fn=loss_fn, max_iter=max_iter, dim=dim, dropout=0.0, seed=seed)
    ref.sd = sd
    x = unet_ref.assemble_input(cb["fft_data"])
    with torch.set_grad_enabled(backward):
        mask_ref = unet_ref.unet_mask(x, sd, norm_weights=bool(params.get("norm_weights", True)), dropout_p=0.0, training=True)
    res["mask_max_abs"] = float((mask.detach().cpu() - mask_ref.detach()).abs().max())
    # ---------------- oracle downstream of the HIP mask
    mh = mask.detach().cpu().clone().requires_grad_(backward)
    with torch.set_grad_enabled(backward):
        w = train_ref.gather_weights(mh, cb["raw_pc"])
        icp = ref.icp if backward else type(ref.icp)("pt2pl", differentiable=False, max_iterations=max_iter, tolerance=1e-5)
        out = icp.icp(cb["filtered_pc"], cb["map_pc"], T_init=cb["T_init"], weight=w, trim_dist=5.0, loss_fn=loss_fn, dim=dim)
    mism = 0
    if backward:
        for k in range(out["num_iter"]):
            act = out["hist"]["active"][k].numpy()
            mism += int((idx_hip[k][act] != out["hist"]["idx"][k].numpy()[act]).sum())
    else:
        k = out["num_iter"] - 1
        act = out["hist"]["active"][k].numpy()
        mism += int((idx_hip[0][act] != out["hist"]["idx"][k].numpy()[act]).sum())
    res["idx_mismatches"] = mism
    res["icp_iters"] = out["num_iter"]
    Tg, Tr = T.detach().cpu().numpy().astype(np.float64), out["T"].detach().numpy().astype(np.float64)
    res["pose_trans_err"] = float(np.abs(Tg[:, :dim, 3] - Tr[:, :dim, 3]).max())
    if dim == 2:
        res["pose_rot_err"] = float(np.abs(np.arctan2(Tg[:, 1, 0], Tg[:, 0, 0]) - np.arctan2(Tr[:, 1, 0], Tr[:, 0, 0])).max())
    else:       # SE(3): angle of the relative rotation R_hip R_ref^T (from its skew part: exact for small angles)
        Rd = Tg[:, :3, :3] @ np.transpose(Tr[:, :3, :3], (0, 2, 1))
        skew = 0.5 * np.stack([Rd[:, 2, 1] - Rd[:, 1, 2], Rd[:, 0, 2] - Rd[:, 2, 0], Rd[:, 1, 0] - Rd[:, 0, 1]], axis=1)
        res["pose_rot_err"] = float(np.arcsin(np.clip(np.linalg.norm(skew, axis=1), 0.0, 1.0)).max())
    if not backward:
        return res
    loss_d, _ = train_ref.eval_training_loss(out["T"], mh, None, cb["T_gt"], cb["fft_data"], None, cb["map_pc"], None, lw)
    loss_d.backward()
    res["loss_rel_err"] = abs(float(loss) - float(loss_d)) / max(1.0, abs(float(loss_d)))
    gm, gm_ref = mask.grad.cpu().numpy(), mh.grad.numpy()
    res["mask_grad_rel"] = float(np.abs(gm - gm_ref).max() / np.abs(gm_ref).max())
    # The maximum above is set by ONE pixel per image: the normalised mask is exactly 1 at its arg-max
    # (icp_weight_policy.py:192-193) and BCELoss's gradient there is 1 / (eps = 1e-12) / N, ~1e5.  The part that
    # comes through the dICP backward lives on the pixels the scan points sample; judge it on those pixels
    # alone, against their own scale.
    from oracle import radar_ref
    taps = radar_ref.extract_weights_grad_mask(tuple(mh.shape), cb["raw_pc"].numpy(),
                                               np.ones(cb["raw_pc"].shape[:2], np.float32)) != 0
    sel = taps & (mh.detach().numpy() < 1.0)
    res["mask_grad_rel_taps"] = float(np.abs(gm - gm_ref)[sel].max() / np.abs(gm_ref)[sel].max())
    res["mask_grad_taps_scale"] = float(np.abs(gm_ref)[sel].max())
    # ---------------- oracle end to end (fp32 network): the bf16 budget of the parameter gradients
    w2 = train_ref.gather_weights(mask_ref, cb["raw_pc"])
    out2 = ref.icp.icp(cb["filtered_pc"], cb["map_pc"], T_init=cb["T_init"], weight=w2, trim_dist=5.0, loss_fn=loss_fn, dim=dim)
    loss_r, _ = train_ref.eval_training_loss(out2["T"], mask_ref, None, cb["T_gt"], cb["fft_data"], None, cb["map_pc"], None, lw)
    loss_r.backward()
    gp = dict(model.named_parameters())
    num = sum(float(((gp[k].grad.cpu() - sd[k].grad) ** 2).sum()) for k in sd)
    den = sum(float((sd[k].grad ** 2).sum()) for k in sd)
    res["param_grad_rel"] = (num / den) ** 0.5
    cos = {k: float(torch.nn.functional.cosine_similarity(gp[k].grad.cpu().flatten(), sd[k].grad.flatten(), dim=0)) for k in sd
           if sd[k].numel() > 1}            # (a one-element tensor only has a sign)
    kmin = min(cos, key=cos.get)
    res["param_grad_cos_min"] = cos[kmin]
    res["param_grad_cos_min_name"] = kmin
    res["loss_fp32_oracle"] = float(loss_r)
    res["loss_hip"] = float(loss)
    return res
