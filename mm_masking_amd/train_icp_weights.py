"""Trainer-side pieces of the hot path — mirror of the step-level functions of the
reference driver ``mm_masking/train_icp_weights.py``:

  train_policy            :22-69     one pass of zero_grad -> forward -> loss -> backward -> step
  validate_policy         :71-177    inference path (no_grad, ICP_alg_inference)
  eval_training_loss      :179-253   rot + trans + optional BCE mask losses
  eval_validation_loss    :255-273
  generate_baseline       :275-344   override-mask baselines (U-Net bypassed)
  default_params          :354-410   the hard-coded ``params`` dict
  fit                     :486-596   the epoch loop (best_policy.pt / epoch_N.pt), plus resume

Neptune logging, figures and checkpoint upload (observability SaaS) are out of
scope.  What the reference's Dataset/DataLoader does per item on the CPU
(CFAR, polar -> Cartesian; icp_weight_dataset.py:182-200,336-352) is done here
per batch on the GPU by ``prepare_batch`` — the synthetic pipeline has no vtr3
point extractor, so the scan cloud comes from cfar_mask + extract_pc.
"""
import time

import torch

from . import radar_utils as ru
from . import synthetic
from .icp_weight_policy import LearnICPWeightPolicy


def default_params(device=None):
    """The reference's params (train_icp_weights.py:354-410) with BASELINE.json's
    configuration of the ICP (pt2pl + Huber) as optional overrides."""
    if device is None:
        device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    return {
        "device": device, "float_type": torch.float32, "gt_eye": True, "pos_std": 2.0, "rot_std": 0.6,
        "log_transform": False, "normalize": ["minmax"], "batch_size_train": 16, "batch_size_test": 32,
        "icp_type": "pt2pt", "learning_rate": 1e-4, "leaky": False, "dropout": 0.05, "batch_norm": False,
        "init_weights": True, "a_thresh": 1.0, "b_thresh": 0.09, "loss_icp_rot_weight": 1.0,
        "loss_icp_trans_weight": 1.0, "loss_fft_mask_weight": 0.0, "loss_map_pts_mask_weight": 1.0,
        "loss_cfar_mask_weight": 0.0, "num_pts_weight": 0.0, "optimizer": "adam", "icp_loss_only_iter": -1,
        "max_iter": 10, "network_input_type": "cartesian", "network_output_type": "cartesian",
        "binary_inference": False, "norm_weights": True, "fft_input": True, "cfar_input": False,
        "range_input": False, "num_epochs": 30,
    }


def loss_weights_from(params):
    """train_icp_weights.py:414-417."""
    return {"icp_rot": params["loss_icp_rot_weight"], "icp_trans": params["loss_icp_trans_weight"],
            "fft": params["loss_fft_mask_weight"], "mask_pts": params["loss_map_pts_mask_weight"],
            "cfar": params["loss_cfar_mask_weight"], "num_pts": params["num_pts_weight"]}


def prepare_batch(raw, params, max_loc_pts=5120, polar_res=0.0596):
    """GPU counterpart of ICPWeightDataset.__getitem__ (icp_weight_dataset.py:323-362)
    for synthetic input: GO-CFAR -> blob centres -> zero-padded scan cloud, and
    polar -> Cartesian for the FFT and CFAR images (kept polar for a polar network).  ``raw`` holds device tensors
    fft_polar (B,400,3360), azimuths (B,400), az_times (B,400), map_pc (B,M,6),
    T_init, T_gt.  Returns the reference's dict-of-dicts batch."""
    fft = raw["fft_polar"]
    az = raw["azimuths"]
    cfar = ru.cfar_mask(fft, polar_res, a_thresh=params["a_thresh"], b_thresh=params["b_thresh"], diff=False)
    pc, _ = ru.extract_pc_padded(cfar, polar_res, az, raw["az_times"], max_loc_pts, diff=False)
    if params.get("network_input_type", "cartesian") == "cartesian":    # icp_weight_dataset.py:350-352
        fft_img, cfar_img = ru._polar_to_cart_pair(fft, cfar, az, polar_res)     # one pass, shared coordinates
    else:                                                               # polar network: images stay (400,3360)
        fft_img, cfar_img = fft, cfar
    loc_data = {"raw_pc": pc, "filtered_pc": pc, "fft_data": fft_img, "fft_cfar": cfar_img, "timestamp": 0}
    map_data = {"pc": raw["map_pc"], "timestamp": 0}
    T_data = {"T_ml_init": raw["T_init"], "T_ml_gt": raw["T_gt"]}
    return {"loc_data": loc_data, "map_data": map_data, "transforms": T_data}


_CONST_CACHE = {}


def _const(kind, device, dtype):
    """Read-only constants (a (1,) zero, the 4x4 identity) built once per device and dtype: the reference creates them
    afresh in every call (train_icp_weights.py:183-188,195), which costs a fill launch each per training step."""
    key = (kind, device, dtype)
    t = _CONST_CACHE.get(key)
    if t is None:
        t = torch.zeros(1, dtype=dtype, device=device) if kind == "zero" else torch.eye(4, dtype=dtype, device=device)
        _CONST_CACHE[key] = t
    return t


# ----------------------------------------------------------------------------- loss terms as single launches (HIP tensors)
_BCE_WS = {}


class _PoseLossFn(torch.autograd.Function):
    """(loss_rot, loss_trans) of the gt_eye form (train_icp_weights.py:193,197-200) in one launch, their gradient in another
    (csrc/mmk_loss.hip) -- through PyTorch the two terms and their backward were ~25 launches of 2-5 us."""

    @staticmethod
    def forward(ctx, T_pred):
        from . import _lib
        T = T_pred.contiguous().float()
        out = torch.empty(2, dtype=torch.float32, device=T.device)
        _lib.check(_lib.lib().mmk_pose_loss_fwd(_lib.ptr(T), T.shape[0], _lib.ptr(out), _lib.stream_ptr(T.device)))
        ctx.save_for_backward(T)
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g_rot, g_trans):
        from . import _lib
        (T,) = ctx.saved_tensors
        gT = torch.empty_like(T)
        gr = None if g_rot is None else g_rot.contiguous().float()
        gt = None if g_trans is None else g_trans.contiguous().float()
        _lib.check(_lib.lib().mmk_pose_loss_bwd(_lib.ptr(T), T.shape[0], _lib.ptr(gr), _lib.ptr(gt), _lib.ptr(gT), _lib.stream_ptr(T.device)))
        return gT


class _BceMeanFn(torch.autograd.Function):
    """torch.nn.BCELoss()(mask, target) (mean reduction, logs clamped at -100) as one pass + an ordered final sum, and its
    gradient as one pass (csrc/mmk_loss.hip); deterministic.  (An input outside [0, 1] gives NaN here where torch raises.)"""

    @staticmethod
    def forward(ctx, mask, target):
        from . import _lib
        L = _lib.lib()
        x, t = mask.contiguous().float(), target.contiguous().float()
        if x.shape != t.shape:
            raise ValueError("Using a target size (%s) that is different to the input size (%s) is deprecated. "
                             "Please ensure they have the same size." % (tuple(t.shape), tuple(x.shape)))
        key = (x.device.index, torch.cuda.current_stream(x.device).cuda_stream)
        ws = _BCE_WS.get(key)
        if ws is None:
            ws = _BCE_WS[key] = torch.empty(int(L.mmk_bce_ws_bytes()), dtype=torch.uint8, device=x.device)
        out = torch.empty((), dtype=torch.float32, device=x.device)
        _lib.check(L.mmk_bce_mean_fwd(_lib.ptr(x), _lib.ptr(t), x.numel(), _lib.ptr(ws), ws.numel(), _lib.ptr(out), _lib.stream_ptr(x.device)))
        ctx.save_for_backward(x, t)
        return out

    @staticmethod
    def backward(ctx, g):
        from . import _lib
        x, t = ctx.saved_tensors
        gx = torch.empty_like(x)
        _lib.check(_lib.lib().mmk_bce_mean_bwd(_lib.ptr(x), _lib.ptr(t), x.numel(), _lib.ptr(g.contiguous().float()), _lib.ptr(gx),
                                               _lib.stream_ptr(x.device)))
        return gx, None


def _bce_mean(mask, target):
    """mask_criterion(mask, target) of the reference (torch.nn.BCELoss, train_icp_weights.py:180): the fused kernels on a HIP
    device, PyTorch's own on the CPU (host-logic tests)."""
    if mask.is_cuda and mask.dtype == torch.float32:
        return _BceMeanFn.apply(mask, target)
    return torch.nn.BCELoss()(mask, target)


def eval_training_loss(T_pred, mask, num_non0, batch_T_gt, batch_scan, batch_map, model, loss_weights=[],
                       icp_loss_only_iter=0, gt_eye=True, epoch=0):
    """train_icp_weights.py:179-253.  Same values and shapes as the reference's expression
    ``w_rot * loss_rot + ... + w_num * loss_num_pts`` (a term that is switched off is a (1,) zero there, so the sum has
    shape (1,) whenever one is); the switched-off terms are not multiplied and added on the GPU, though: they are a
    shared constant zero (x + 0 = x bit for bit), which takes ~20 two-microsecond launches out of a training step."""
    mask_criterion = _bce_mean
    dev, dt = T_pred.device, T_pred.dtype
    zero = _const("zero", dev, dt)
    terms = {"rot": None, "trans": None, "fft": None, "mask_pts": None, "cfar": None, "num_pts": None}

    if (loss_weights["icp_rot"] > 0.0 or loss_weights["icp_trans"] > 0.0) and gt_eye and T_pred.is_cuda and dt == torch.float32:
        terms["rot"], terms["trans"] = _PoseLossFn.apply(T_pred)          # one launch each way (csrc/mmk_loss.hip)
    elif loss_weights["icp_rot"] > 0.0 or loss_weights["icp_trans"] > 0.0:
        eye = _const("eye", dev, dt)
        if gt_eye:
            xi_wedge = T_pred - eye
        else:
            xi_wedge = torch.matmul(T_pred, torch.inverse(batch_T_gt)) - eye
        xi_r = xi_wedge[:, 0:2, 3]
        xi_theta = xi_wedge[:, 1, 0].unsqueeze(-1)
        terms["rot"] = torch.norm(xi_theta, dim=1).mean()
        terms["trans"] = torch.norm(xi_r, dim=1).mean()
    if icp_loss_only_iter <= 0 or (icp_loss_only_iter > 0 and epoch < icp_loss_only_iter) or \
            (loss_weights["icp_rot"] <= 0 and loss_weights["icp_trans"] <= 0):
        if loss_weights["fft"] > 0.0:
            fft_data = batch_scan["fft_data"].to(mask.device)
            mean_azimuth = torch.mean(fft_data, dim=(1, 2), keepdim=True)
            fft_mask = torch.where(fft_data > 3.0 * mean_azimuth, torch.ones_like(fft_data), torch.zeros_like(fft_data))
            terms["fft"] = mask_criterion(mask, fft_mask)
        if loss_weights["cfar"] > 0.0:
            terms["cfar"] = mask_criterion(mask, batch_scan["fft_cfar"].to(mask.device))
        if loss_weights["mask_pts"] > 0.0:
            map_pts_mask = ru.extract_bev_from_pts(batch_map["pc"].to(mask.device))
            terms["mask_pts"] = mask_criterion(mask, map_pts_mask)
        if loss_weights["num_pts"] > 0.0:
            terms["num_pts"] = model.mean_all_pts - num_non0

    wkey = {"rot": "icp_rot", "trans": "icp_trans", "fft": "fft", "mask_pts": "mask_pts", "cfar": "cfar", "num_pts": "num_pts"}
    loss = None
    loss_components = {}
    any_off = False
    for name in ("rot", "trans", "fft", "mask_pts", "cfar", "num_pts"):     # the reference's order of summation
        if terms[name] is None:
            loss_components[name] = zero
            any_off = True
            continue
        w = loss_weights[wkey[name]]
        # (a weight of exactly 1 -- the reference's defaults for the pose and map-point terms -- is not multiplied on the GPU:
        # 1.0 * x = x bit for bit, and the product and its backward were two ~5 us launches per term)
        t = terms[name] if (isinstance(w, (int, float)) and float(w) == 1.0) else w * terms[name]
        loss_components[name] = t.detach()
        loss = t if loss is None else loss + t
    if loss is None:
        loss = zero.clone()
    elif any_off and loss.dim() == 0:
        loss = loss.reshape(1)
    return loss, loss_components


def eval_validation_loss(T_pred, batch_T_gt, gt_eye=True):
    """train_icp_weights.py:255-273 -> [||(theta,x,y)||, |theta|, ||(x,y)||] batch means."""
    eye = _const("eye", T_pred.device, T_pred.dtype)
    if gt_eye:
        xi_wedge = T_pred - eye
    else:
        xi_wedge = torch.matmul(T_pred, torch.inverse(batch_T_gt)) - eye
    xi_r = xi_wedge[:, 0:2, 3]
    xi_theta = xi_wedge[:, 1, 0].unsqueeze(-1)
    xi_stack = torch.cat((xi_theta, xi_r), dim=1)
    return torch.hstack((torch.norm(xi_stack, dim=1).mean(), torch.norm(xi_theta, dim=1).mean(),
                         torch.norm(xi_r, dim=1).mean()))


def train_step(model, batch, opt, loss_weights, device, gt_eye=True, epoch=None, icp_loss_only_iter=0,
               grad_sync=None):
    """Body of the reference's batch loop (train_icp_weights.py:31-58).  ``grad_sync``
    (optional callable) is where a data-parallel job all-reduces the gradients."""
    batch_scan, batch_map, batch_T = batch["loc_data"], batch["map_data"], batch["transforms"]
    batch_T_init = batch_T["T_ml_init"].to(device)
    if grad_sync is not None and hasattr(grad_sync, "zero_grad"):
        grad_sync.zero_grad()           # keeps the flat gradient buffer attached
    else:
        opt.zero_grad()
    T_pred, mask, num_non0 = model(batch_scan, batch_map, batch_T_init)
    batch_T_gt = batch_T["T_ml_gt"].to(device)
    loss, loss_comp = eval_training_loss(T_pred, mask, num_non0, batch_T_gt, batch_scan, batch_map, model,
                                         loss_weights=loss_weights, icp_loss_only_iter=icp_loss_only_iter,
                                         gt_eye=gt_eye, epoch=epoch)
    loss.backward()
    if grad_sync is not None:
        grad_sync()
    opt.step()
    return loss.detach(), loss_comp


def train_policy(model, iterator, opt, loss_weights=[], device="cpu", epoch=None, icp_loss_only_iter=0, gt_eye=True,
                 grad_sync=None):
    """train_icp_weights.py:22-69."""
    model.train()
    loss_hist = 0.0
    loss_comp_hist = []
    n = 0
    for batch in iterator:
        loss, loss_comp = train_step(model, batch, opt, loss_weights, device, gt_eye=gt_eye, epoch=epoch,
                                     icp_loss_only_iter=icp_loss_only_iter, grad_sync=grad_sync)
        loss_hist += loss
        loss_comp_hist.append(loss_comp)
        n += 1
    mean_loss = loss_hist / n
    mean_loss_comp = {k: sum(d[k] for d in loss_comp_hist) / len(loss_comp_hist) for k in loss_comp_hist[0]}
    _check_device_errors(device)
    return mean_loss, mean_loss_comp


def _check_device_errors(device):
    """The ICP kernels report internal errors through a status word that is examined without synchronising the step (at the next
    ICP call: dICP/ICP.py).  At the end of a pass nothing follows, so wait for the calls in flight and raise here: a flagged
    call's poses must not reach a checkpoint or a reported metric unnoticed (one synchronisation per epoch)."""
    if torch.device(device).type == "cuda":
        import importlib
        icp_mod = importlib.import_module(__package__ + ".dICP.ICP")     # the module (the package re-exports the class under its name)
        icp_mod.check_errors(wait=True)


def validate_policy(model, iterator, gt_eye=True, device="cpu", binary=False, neptune_run=None, epoch=None):
    """train_icp_weights.py:71-177 (without the Neptune figures)."""
    model.eval()
    val_acc = torch.zeros((1, 3), device=device)
    mean_num_pc, mean_w, max_w, min_w = 0.0, 0.0, 0.0, 1000.0
    n = 0
    with torch.no_grad():
        for batch in iterator:
            batch_scan, batch_map, batch_T = batch["loc_data"], batch["map_data"], batch["transforms"]
            batch_T_gt = batch_T["T_ml_gt"].to(device)
            batch_T_init = batch_T["T_ml_init"].to(device)
            T_pred, mask, _ = model(batch_scan, batch_map, batch_T_init, binary=binary)
            mean_num_pc += model.mean_num_pts
            max_w = model.max_w if model.max_w > max_w else max_w
            min_w = model.min_w if model.min_w < min_w else min_w
            mean_w += model.mean_w
            val_acc += eval_validation_loss(T_pred, batch_T_gt, gt_eye=gt_eye)
            n += 1
    _check_device_errors(device)
    return val_acc / n, mean_num_pc / n, mean_w / n, max_w, min_w


def generate_baseline(model, iterator, baseline_type="train", device="cpu",
                      loss_weights={"icp": 1.0, "fft": 0.0, "mask_pts": 0.0, "cfar": 0.0}, binary=False, gt_eye=True):
    """train_icp_weights.py:275-344: ICP with a fixed (non-learned) mask."""
    model.train() if baseline_type == "train" else model.eval()
    loss_init_hist, loss_ones_hist = [], []
    with torch.no_grad():
        for batch in iterator:
            batch_scan, batch_map, batch_T = batch["loc_data"], batch["map_data"], batch["transforms"]
            batch_T_gt = batch_T["T_ml_gt"].to(device)
            batch_T_init = batch_T["T_ml_init"].to(device)
            fft_data = batch_scan["fft_data"].to(device)
            if loss_weights.get("cfar", 0.0) > 0.0:
                ones_mask = batch_scan["fft_cfar"].to(device)
            elif loss_weights.get("fft", 0.0) > 0.0:
                mean_azimuth = torch.mean(fft_data, dim=(1, 2), keepdim=True)
                ones_mask = torch.where(fft_data > 3.0 * mean_azimuth, torch.ones_like(fft_data),
                                        torch.zeros_like(fft_data))
            elif loss_weights.get("mask_pts", 0.0) > 0.0:
                ones_mask = ru.extract_bev_from_pts(batch_map["pc"].to(device))
            else:
                ones_mask = torch.ones_like(fft_data)
            T_pred_ones, mask_ones, num_non0 = model(batch_scan, batch_map, batch_T_init, binary=binary,
                                                     override_mask=ones_mask)
            if baseline_type == "train":
                li, _ = eval_training_loss(batch_T_init, mask_ones, num_non0, batch_T_gt, batch_scan, batch_map, model,
                                           loss_weights=loss_weights, gt_eye=gt_eye)
                lo, _ = eval_training_loss(T_pred_ones, mask_ones, num_non0, batch_T_gt, batch_scan, batch_map, model,
                                           loss_weights=loss_weights, gt_eye=gt_eye)
            else:
                li = eval_validation_loss(batch_T_init, batch_T_gt, gt_eye=gt_eye)[0]
                lo = eval_validation_loss(T_pred_ones, batch_T_gt, gt_eye=gt_eye)[0]
            loss_init_hist.append(float(li))
            loss_ones_hist.append(float(lo))
    _check_device_errors(device)
    return sum(loss_init_hist) / len(loss_init_hist), sum(loss_ones_hist) / len(loss_ones_hist)


def make_optimizer(policy, params):
    """train_icp_weights.py:462-465."""
    if params["optimizer"] == "adam":
        # the same update rule; on the GPU all 46 tensors in one launch instead of seven foreach passes
        fused = all(p.is_cuda for p in policy.parameters())
        return torch.optim.Adam(policy.parameters(), lr=params["learning_rate"], fused=fused)
    return torch.optim.SGD(policy.parameters(), lr=params["learning_rate"], nesterov=True, momentum=1.0)


# ----------------------------------------------------------------------------- epoch loop + resume
def save_checkpoint(path, policy, opt, epoch, best_norm):
    """Everything a run needs to continue where it stopped (absent upstream, which saves the
    state_dict only: train_icp_weights.py:534-537,577-578): parameters, optimizer state, the epoch
    that finished, the best validation norm so far and the policy's dropout-seed counter.  Tensors,
    numbers and plain containers only, so it loads with ``weights_only=True``."""
    torch.save({"model": policy.state_dict(), "optimizer": opt.state_dict(), "epoch": int(epoch),
                "best_norm": float(best_norm), "policy_step": int(getattr(policy, "_step", 0))}, path)


def load_checkpoint(path, policy, opt=None, map_location=None):
    """Restores what ``save_checkpoint`` wrote; a bare state_dict file (the reference's
    ``best_policy.pt`` / ``epoch_N.pt``) loads the parameters only.  -> (next epoch, best_norm)."""
    ck = torch.load(path, map_location=map_location, weights_only=True)
    if "model" not in ck:
        policy.load_state_dict(ck)
        return 0, None
    policy.load_state_dict(ck["model"])
    if opt is not None:
        opt.load_state_dict(ck["optimizer"])
    policy._step = int(ck.get("policy_step", 0))
    return int(ck["epoch"]) + 1, float(ck["best_norm"])


def fit(policy, training_iterator, validation_iterator, opt, params, checkpoint_dir, loss_weights=None,
        start_epoch=0, best_norm=None, grad_sync=None, is_main=True, log=print):
    """The reference's epoch loop (train_icp_weights.py:486-596) without Neptune: baselines, a
    validation before training, then per epoch train_policy -> validate_policy, ``best_policy.pt`` when
    the total validation norm improves (or at epoch 0), ``epoch_N.pt`` every epoch (both bare
    state_dicts, as upstream), and finally a validation of the best policy.  Additionally writes
    ``resume.pt`` (save_checkpoint) every epoch; pass ``start_epoch`` / ``best_norm`` from
    ``load_checkpoint`` to continue a run.  Returns a history dict.

    Multi-rank jobs: every rank calls fit() with iterators of the SAME length (baselines, validation and training
    forwards may each hold the global min-max collective); ``is_main`` only gates the files that are written."""
    import os
    dev = params["device"]
    lw = loss_weights if loss_weights is not None else loss_weights_from(params)
    if is_main:
        os.makedirs(checkpoint_dir, exist_ok=True)
    best_path = os.path.join(checkpoint_dir, "best_policy.pt")
    hist = {"loss": [], "loss_comp": [], "acc": [], "epoch_train_time": [], "epoch_val_time": []}
    if start_epoch == 0:
        hist["train_baseline"] = generate_baseline(policy, training_iterator, baseline_type="train", device=dev, binary=False,
                                                   loss_weights=lw, gt_eye=params["gt_eye"])
        hist["val_baseline"] = generate_baseline(policy, validation_iterator, baseline_type="val", device=dev,
                                                 binary=params["binary_inference"], gt_eye=params["gt_eye"])
        avg_norm = _rank_mean(validate_policy(policy, validation_iterator, device=dev, binary=params["binary_inference"],
                                              gt_eye=params["gt_eye"], epoch=-1)[0])
        best_norm = float(avg_norm[0, 0])
        log("Norm before training: %.6f" % best_norm)
    for epoch in range(start_epoch, params["num_epochs"]):
        tic = time.time()
        mean_loss, mean_loss_comp = train_policy(policy, training_iterator, opt, lw, device=dev, epoch=epoch,
                                                 icp_loss_only_iter=params["icp_loss_only_iter"], gt_eye=params["gt_eye"],
                                                 grad_sync=grad_sync)
        mean_loss = float(mean_loss)
        t_train = time.time() - tic
        tic = time.time()
        avg_norm = _rank_mean(validate_policy(policy, validation_iterator, epoch=epoch, device=dev, binary=params["binary_inference"],
                                              gt_eye=params["gt_eye"])[0])
        t_val = time.time() - tic
        total = float(avg_norm[0, 0])
        if total < best_norm or epoch == 0:
            best_norm = total
            if is_main:
                torch.save(policy.state_dict(), best_path)
        if is_main:
            torch.save(policy.state_dict(), os.path.join(checkpoint_dir, "epoch_{}.pt".format(epoch)))
            save_checkpoint(os.path.join(checkpoint_dir, "resume.pt"), policy, opt, epoch, best_norm)
        hist["loss"].append(mean_loss)
        hist["loss_comp"].append({k: float(v) for k, v in mean_loss_comp.items()})
        hist["acc"].append([float(v) for v in avg_norm[0]])
        hist["epoch_train_time"].append(t_train)
        hist["epoch_val_time"].append(t_val)
        log("EPOCH %d  loss %.5f  norm %.5f  best %.5f  (train %.1f s, val %.1f s)" % (epoch, mean_loss, total, best_norm,
                                                                                      t_train, t_val))
    hist["best_norm"] = best_norm
    # Final validation of the best policy (train_icp_weights.py:577-596), on EVERY rank of a multi-rank job: a forward
    # may hold a collective (the global min-max normalisation), so a validation that only the main rank entered would
    # hang or pair with another rank's next collective.  The main rank reads best_policy.pt and its weights are
    # broadcast, so the ranks end the run with identical (best) parameters.
    import torch.distributed as dist
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    have_best = is_main and os.path.exists(best_path)
    if multi:
        flag = torch.tensor([1.0 if have_best else 0.0], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        have_best = bool(flag.item() > 0)
    if have_best:
        if is_main and os.path.exists(best_path):
            policy.load_state_dict(torch.load(best_path, map_location=dev, weights_only=True))
        if multi:
            src = _main_rank(is_main, dev)
            for t in policy.state_dict().values():
                dist.broadcast(t, src=src)
        hist["final_acc"] = [float(v) for v in _rank_mean(validate_policy(policy, validation_iterator, device=dev,
                                                                         binary=params["binary_inference"], gt_eye=params["gt_eye"])[0])[0]]
    return hist


def _rank_mean(t):
    """Mean over the ranks of a data-parallel job (identity in a single process): each rank validates its own shard of the
    validation set, and the reference selects its best policy on the WHOLE set (train_icp_weights.py:534-537) -- with equal
    shard sizes the mean of the shard means.  Every rank then takes the same best-policy decisions."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return t
    t = t.clone()
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t / dist.get_world_size()


def _main_rank(is_main, dev):
    """The rank that passed is_main=True (the lowest one, should several have): found with one MIN all-reduce."""
    import torch.distributed as dist
    t = torch.tensor([float(dist.get_rank()) if is_main else float(dist.get_world_size())], device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return int(t.item())


class SyntheticIterator:
    """Stands in for DataLoader(ICPWeightDataset): yields ``n_batches`` prepared
    batches of ``batch_size`` pairs; pair indices are strided by world size so
    that ranks draw disjoint pairs of one seeded stream."""

    def __init__(self, params, batch_size, n_batches, rank=0, world_size=1, dataset_type="train", max_loc_pts=5120,
                 m_valid=20000, m_pad=20480, start=0, density="survey"):
        self.params, self.bs, self.nb, self.density = params, batch_size, n_batches, density
        self.rank, self.ws, self.kind = rank, world_size, dataset_type
        self.max_loc_pts, self.m_valid, self.m_pad, self.start = max_loc_pts, m_valid, m_pad, start

    def __len__(self):
        return self.nb

    def raw_batch(self, i):
        first = self.start + i * self.bs * self.ws
        idx = [first + self.rank + j * self.ws for j in range(self.bs)]
        return synthetic.make_batch(idx, device=self.params["device"], m_valid=self.m_valid, m_pad=self.m_pad,
                                    dataset_type=self.kind, pos_std=self.params["pos_std"],
                                    rot_std=self.params["rot_std"], density=self.density)

    def __iter__(self):
        for i in range(self.nb):
            yield prepare_batch(self.raw_batch(i), self.params, self.max_loc_pts)
