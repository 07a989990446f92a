"""Body of the training-trajectory parity test (tests/test_gpu_step_parity.py): N optimizer steps of the product's
default path (hand-written bf16 U-Net, HIP dICP, fused losses, fused Adam) next to the same N steps of
oracle/train_ref.TrainStepRef (fp32 CPU port of /root/reference/mm_masking/train_icp_weights.py:22-69 with the
optimizer of :462-465) from the same state_dict on the same batches, dropout 0.

One-step gradient agreement (tests/step_parity.py) does not say where a run ends up: Adam divides every gradient by
its own running magnitude, so a parameter whose bf16 gradient has the wrong sign moves the wrong way at full step
size.  What a drop-in trainer owes its user is the trajectory: the loss step by step, the validation metric the
reference selects checkpoints by (eval_validation_loss, :255-273, :534-537), and how far the two parameter vectors
drift apart relative to how far training moved them.
"""
import numpy as np
import torch

from mm_masking_amd import synthetic
from mm_masking_amd import train_icp_weights as trn
from mm_masking_amd.icp_weight_policy import LearnICPWeightPolicy
from oracle import dicp_ref, train_ref, unet_ref

LOSS_FN = {"name": "huber", "metric": 1.0}


def _thin(batch, raw, stride):
    """Every ``stride``-th scan point (the clouds come out of the peak extraction in azimuth order: a prefix would be one
    sector of the scene), zero rows stay zero rows.  The images keep their full size."""
    pc = batch["loc_data"]["filtered_pc"][:, ::stride].contiguous()
    loc = dict(batch["loc_data"], raw_pc=pc, filtered_pc=pc)
    return {"loc_data": loc, "map_data": batch["map_data"], "transforms": batch["transforms"]}


def _cpu_batch(batch):
    return {"fft_data": batch["loc_data"]["fft_data"].cpu(), "raw_pc": batch["loc_data"]["raw_pc"].cpu(),
            "filtered_pc": batch["loc_data"]["filtered_pc"].cpu(), "map_pc": batch["map_data"]["pc"].cpu(),
            "T_init": batch["transforms"]["T_ml_init"].cpu(), "T_gt": batch["transforms"]["T_ml_gt"].cpu()}


def run(dev, steps=30, B=4, stride=8, m_valid=2000, m_pad=2048, n_batches=3, norm_weights=True, seed=4321, first=8000,
        max_iter=10, log=None):
    params = trn.default_params(dev)
    params.update({"dropout": 0.0, "icp_type": "pt2pl", "icp_loss_fn": LOSS_FN, "max_iter": max_iter,
                   "norm_weights": norm_weights})
    lw = trn.loss_weights_from(params)
    batches = []
    for i in range(n_batches + 1):                           # the last one is held out for the validation metric
        raw = synthetic.make_batch(list(range(first + i * B, first + (i + 1) * B)), device=dev, m_valid=m_valid, m_pad=m_pad,
                                   dataset_type="train" if i < n_batches else "val")
        batches.append(_thin(trn.prepare_batch(raw, params, max_loc_pts=5120), raw, stride))
    cpu_batches = [_cpu_batch(b) for b in batches]
    torch.manual_seed(seed)
    model = LearnICPWeightPolicy(params).to(dev)
    assert model.unet_backend == "hip"
    theta0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    opt = trn.make_optimizer(model, params)
    ref = train_ref.TrainStepRef(icp_type="pt2pl", loss_fn=LOSS_FN, max_iter=max_iter, dim=2, dropout=0.0,
                                 lr=params["learning_rate"], state_dict=theta0, norm_weights=norm_weights)

    def validation(model_T):
        return [float(v) for v in model_T]

    def val_hip():
        model.eval()
        with torch.no_grad():
            vb = batches[-1]
            T, _, _ = model(vb["loc_data"], vb["map_data"], vb["transforms"]["T_ml_init"])
            out = trn.eval_validation_loss(T, vb["transforms"]["T_ml_gt"])
        model.train()
        return validation(out.cpu())

    def val_ref():
        vb = cpu_batches[-1]
        with torch.no_grad():
            x = unet_ref.assemble_input(vb["fft_data"])
            mask = unet_ref.unet_mask(x, ref.sd, norm_weights=norm_weights, dropout_p=0.0, training=False)
            w = train_ref.gather_weights(mask, vb["raw_pc"])
            icp = dicp_ref.ICPRef("pt2pl", differentiable=False, max_iterations=50, tolerance=1e-5)
            T = icp.icp(vb["filtered_pc"], vb["map_pc"], T_init=vb["T_init"], weight=w, trim_dist=5.0, loss_fn=LOSS_FN, dim=2)["T"]
            return validation(train_ref.eval_validation_loss(T, vb["T_gt"]))

    res = {"steps": steps, "B": B, "scan_rows": int(batches[0]["loc_data"]["filtered_pc"].shape[1]), "map_rows": m_pad,
           "norm_weights": norm_weights, "lr": params["learning_rate"], "loss_hip": [], "loss_ref": [],
           "val_hip_before": val_hip(), "val_ref_before": val_ref()}
    model.train()
    for s in range(steps):
        k = s % n_batches
        lh, _ = trn.train_step(model, batches[k], opt, lw, dev)
        lr_ = ref.step(cpu_batches[k])
        res["loss_hip"].append(float(lh))
        res["loss_ref"].append(float(lr_))
        if log is not None:
            log("step %2d  loss hip %.6f  ref %.6f  rel %.2e" % (s, float(lh), lr_, abs(float(lh) - lr_) / abs(lr_)))
    res["val_hip_after"], res["val_ref_after"] = val_hip(), val_ref()
    lh, lr_ = np.array(res["loss_hip"]), np.array(res["loss_ref"])
    res["loss_rel_max"] = float(np.max(np.abs(lh - lr_) / np.abs(lr_)))
    res["loss_rel_last"] = float(abs(lh[-1] - lr_[-1]) / abs(lr_[-1]))
    # the loss must actually have moved, or "the same loss" says nothing
    res["loss_drop_ref"] = float(lr_[:n_batches].mean() - lr_[-n_batches:].mean())
    res["loss_drop_hip"] = float(lh[:n_batches].mean() - lh[-n_batches:].mean())
    sd_h = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    num = sum(float(((sd_h[k] - ref.sd[k].detach()) ** 2).sum()) for k in theta0)
    moved = sum(float(((ref.sd[k].detach() - theta0[k]) ** 2).sum()) for k in theta0)
    moved_h = sum(float(((sd_h[k] - theta0[k]) ** 2).sum()) for k in theta0)
    res["param_drift_over_distance_moved"] = (num / moved) ** 0.5
    res["distance_moved_ref"], res["distance_moved_hip"] = moved ** 0.5, moved_h ** 0.5
    dots = sum(float(((sd_h[k] - theta0[k]) * (ref.sd[k].detach() - theta0[k])).sum()) for k in theta0)
    res["update_cosine"] = dots / (moved ** 0.5 * moved_h ** 0.5)
    return res
