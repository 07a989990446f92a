"""ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

PyTorch-CPU restatement of the trainer-side pieces inside the timed step
(/root/reference/mm_masking/train_icp_weights.py:22-69 train_policy,
:179-253 eval_training_loss, :255-273 eval_validation_loss; SURVEY.md §8a
Group T) and of one whole ``train_policy`` step (mask U-Net -> extract_weights
-> 10-iteration ICP -> loss -> backward -> Adam) used as the reported CPU
baseline (bench.py ``cpu_baseline``, kind "port").  Losses pinned by
tests/golden/losses.npz.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import dicp_ref, radar_ref, unet_ref


def _xi_wedge(T_pred, T_gt, gt_eye):
    eye = torch.eye(4, dtype=T_pred.dtype)
    if gt_eye:
        return T_pred - eye
    return T_pred @ torch.inverse(T_gt) - eye


def eval_validation_loss(T_pred, T_gt, gt_eye=True):
    """train_icp_weights.py:255-273 -> [||(theta,x,y)||, |theta|, ||(x,y)||] batch means."""
    xi = _xi_wedge(T_pred, T_gt, gt_eye)
    xi_r = xi[:, 0:2, 3]
    xi_t = xi[:, 1, 0].unsqueeze(-1)
    st = torch.cat((xi_t, xi_r), dim=1)
    return torch.stack([st.norm(dim=1).mean(), xi_t.norm(dim=1).mean(), xi_r.norm(dim=1).mean()])


def eval_training_loss(T_pred, mask, num_non0, T_gt, fft_data, fft_cfar, map_pc, mean_all_pts, loss_weights,
                       icp_loss_only_iter=0, gt_eye=True, epoch=0):
    """train_icp_weights.py:179-253 -> (loss, components dict)."""
    z = torch.zeros(1, dtype=T_pred.dtype)
    l_rot = l_trans = l_fft = l_pts = l_cfar = l_num = z
    lw = loss_weights
    if lw["icp_rot"] > 0.0 or lw["icp_trans"] > 0.0:
        xi = _xi_wedge(T_pred, T_gt, gt_eye)
        l_rot = xi[:, 1, 0].unsqueeze(-1).norm(dim=1).mean()
        l_trans = xi[:, 0:2, 3].norm(dim=1).mean()
    if icp_loss_only_iter <= 0 or (icp_loss_only_iter > 0 and epoch < icp_loss_only_iter) or \
            (lw["icp_rot"] <= 0 and lw["icp_trans"] <= 0):
        if lw["fft"] > 0.0:
            mean_scan = fft_data.mean(dim=(1, 2), keepdim=True)
            l_fft = F.binary_cross_entropy(mask, (fft_data > 3.0 * mean_scan).to(mask.dtype))
        if lw["cfar"] > 0.0:
            l_cfar = F.binary_cross_entropy(mask, fft_cfar)
        if lw["mask_pts"] > 0.0:
            bev = torch.from_numpy(radar_ref.extract_bev_from_pts(map_pc.numpy()))
            l_pts = F.binary_cross_entropy(mask, bev)
        if lw["num_pts"] > 0.0:
            l_num = mean_all_pts - num_non0
    loss = lw["icp_rot"] * l_rot + lw["icp_trans"] * l_trans + lw["fft"] * l_fft + lw["mask_pts"] * l_pts \
        + lw["cfar"] * l_cfar + lw["num_pts"] * l_num
    comp = {"rot": lw["icp_rot"] * l_rot, "trans": lw["icp_trans"] * l_trans, "fft": lw["fft"] * l_fft,
            "mask_pts": lw["mask_pts"] * l_pts, "cfar": lw["cfar"] * l_cfar, "num_pts": lw["num_pts"] * l_num}
    return loss, {k: (v.detach() if torch.is_tensor(v) else v) for k, v in comp.items()}


class _GatherWeights(torch.autograd.Function):
    """extract_weights (radar_utils.py:108-128) with the numpy restatement as
    forward and its adjoint scatter-add as backward."""

    @staticmethod
    def forward(ctx, mask, pc):
        w = radar_ref.extract_weights(mask.detach().numpy(), pc.numpy())[0]
        ctx.pc = pc
        ctx.shape = tuple(mask.shape)
        return torch.from_numpy(w)

    @staticmethod
    def backward(ctx, gw):
        g = radar_ref.extract_weights_grad_mask(ctx.shape, ctx.pc.numpy(), gw.numpy())
        return torch.from_numpy(g), None


def gather_weights(mask, pc):
    return _GatherWeights.apply(mask, pc)


DEFAULT_LOSS_WEIGHTS = {"icp_rot": 1.0, "icp_trans": 1.0, "fft": 0.0, "mask_pts": 1.0, "cfar": 0.0, "num_pts": 0.0}


class TrainStepRef:
    """One reference-shaped training step on the CPU (the ``port`` baseline)."""

    def __init__(self, icp_type="pt2pl", loss_fn=None, max_iter=10, dim=2, dropout=0.05, seed=1234, lr=1e-4,
                 loss_weights=None, state_dict=None, norm_weights=True):
        """``state_dict``: start from these parameters (copied) instead of the seeded Xavier initialisation;
        ``norm_weights``: params["norm_weights"] of the reference (icp_weight_policy.py:192-193)."""
        init = unet_ref.init_state_dict(1, seed) if state_dict is None else \
            {k: v.detach().to("cpu", torch.float32).clone() for k, v in state_dict.items()}
        self.sd = {k: v.requires_grad_(True) for k, v in init.items()}
        self.norm_weights = norm_weights
        self.opt = torch.optim.Adam(list(self.sd.values()), lr=lr)
        self.icp = dicp_ref.ICPRef(icp_type, differentiable=True, max_iterations=max_iter, tolerance=1e-5)
        self.loss_fn = loss_fn or {"name": "huber", "metric": 1.0}
        self.dim = dim
        self.dropout = dropout
        self.lw = loss_weights or DEFAULT_LOSS_WEIGHTS

    def forward(self, batch, training=True):
        x = unet_ref.assemble_input(batch["fft_data"])
        mask = unet_ref.unet_mask(x, self.sd, norm_weights=self.norm_weights, dropout_p=self.dropout, training=training)
        w = gather_weights(mask, batch["raw_pc"])
        out = self.icp.icp(batch["filtered_pc"], batch["map_pc"], T_init=batch["T_init"], weight=w,
                           trim_dist=5.0, loss_fn=self.loss_fn, dim=self.dim)
        return out["T"], mask, w

    def step(self, batch):
        self.opt.zero_grad()
        T, mask, _ = self.forward(batch)
        loss, _ = eval_training_loss(T, mask, None, batch["T_gt"], batch["fft_data"], batch.get("fft_cfar"),
                                     batch["map_pc"], None, self.lw)
        loss.backward()
        self.opt.step()
        return float(loss.detach())
