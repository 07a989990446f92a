"""``LearnICPWeightPolicy`` — host-side mirror of the reference module
``mm_masking/icp_weight_policy.py`` (same constructor ``params`` keys, same
``forward``/``icp`` signatures and return values, same ``state_dict`` keys), with

  * the mask U-Net in hand-written NHWC bf16 MFMA kernels (unet_hip.py ->
    csrc/mmk_unet.hip): ReLU (the reference's default) and LeakyReLU(0.1)
    (``params["leaky"]``) networks, with or without BatchNorm (``params["batch_norm"]``,
    unet_hip_bn.py), Cartesian or polar input of any size >= 32 x 32.
    There is no vendor-library (MIOpen) or CPU fallback on a HIP device; the
    ``nn.Module`` tree is the parameter store (identical ``state_dict``) and, with the
    explicit ``unet_backend="torch"`` on a CPU device, the host-logic mirror the CPU
    tests compare with the reference's golden vectors,
  * ``extract_weights`` and the differentiable ICP in hand-written HIP kernels
    (radar_utils.py / dICP/ICP.py of this package -> libmmk_hip.so).

Reference lines: constructor :25-102, conv_block :104-125, forward :127-275,
icp :277-288.  The Neptune/matplotlib plotting block (:221-264) is observability
SaaS and out of scope: ``neptune_run``, ``epoch`` and ``batch_idx`` are accepted
and ignored.  The per-step ``torch.cuda.empty_cache()`` calls (:187,217) are not
reproduced (they only force syncs).
"""
import torch
import torch.nn as nn
from torch.nn import ModuleList

from .dICP.ICP import ICP
from .radar_utils import _extract_weights_stats, extract_weights, form_cart_range_angle_grid, form_polar_range_grid


def weights_init(m):
    """icp_weight_policy.py:15-22."""
    classname = m.__class__.__name__
    if classname.find("Conv") != -1:
        try:
            nn.init.xavier_uniform_(m.weight.data)
            nn.init.zeros_(m.bias)
        except AttributeError:
            print("Skipping initialization of ", classname)


def _global_extrema(c_min, c_max):
    """min / max over the ranks of a data-parallel job (one MAX all-reduce of [-min, max]); the identity in a
    single-process run.  Same reduction as unet_hip.channel_minmax(global_reduce=True) on the HIP path."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return c_min, c_max
    t = torch.stack((-c_min.detach(), c_max.detach()))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return -t[0], t[1]


class LearnICPWeightPolicy(nn.Module):
    def __init__(self, params):
        super().__init__()
        self.res = 0.0596

        icp_type = params["icp_type"]
        network_inputs = {"fft": params["fft_input"], "cfar": params["cfar_input"], "range": params["range_input"]}
        network_input_type = params["network_input_type"]
        device = params["device"]
        max_iter = params["max_iter"]

        self.use_ICP_4_train = params["loss_icp_rot_weight"] > 0.0 and params["loss_icp_trans_weight"] > 0.0

        config_path = "../external/dICP/config/dICP_config.yaml"
        self.ICP_alg = ICP(icp_type=icp_type, config_path=config_path, differentiable=True,
                           max_iterations=max_iter, tolerance=1e-5)
        self.ICP_alg_inference = ICP(icp_type=icp_type, config_path=config_path, differentiable=False,
                                     max_iterations=50, tolerance=1e-5)
        self.float_type = params["float_type"]
        self.device = device
        self.network_inputs = network_inputs
        if network_input_type == "cartesian":
            self.range_mask, _ = form_cart_range_angle_grid(device=device)
        elif network_input_type == "polar":
            self.range_mask = form_polar_range_grid(polar_resolution=self.res, device=device)

        self.network_input_type = network_input_type
        self.network_output_type = params["network_output_type"]
        self.leaky = params["leaky"]
        self.dropout = params["dropout"]
        self.batch_norm = params["batch_norm"]
        self.normalize_type = params["normalize"]
        self.log_transform = params["log_transform"]
        self.a_thres = params["a_thresh"]
        self.b_thres = params["b_thresh"]
        self.gt_eye = params["gt_eye"]
        self.norm_weights = params["norm_weights"]

        # optional keys (absent upstream): what the reference hard-codes in icp()
        self.icp_loss_fn = params.get("icp_loss_fn", {"name": "cauchy", "metric": 1.0})
        self.icp_trim_dist = params.get("icp_trim_dist", 5.0)
        self.icp_dim = params.get("icp_dim", 2)
        # "hip" (the product path): hand-written NHWC bf16 kernels (csrc/mmk_unet.hip), fp32 masters;
        # "torch": the nn.Module tree evaluated by PyTorch on the CPU -- host-logic mirror for tests only
        self.unet_backend = params.get("unet_backend", "hip")
        if self.unet_backend not in ("hip", "torch"):
            raise ValueError("unet_backend must be 'hip' or 'torch' (got %r)" % (self.unet_backend,))
        # data-parallel jobs: reduce the min-max normalisation's extrema over the ranks (one MAX all-reduce of 2C floats), so
        # that it stays global over the whole batch as in the single-process reference (icp_weight_policy.py:151-155).
        # Default: ON whenever the process is a rank of a multi-rank job, so that N ranks x B pairs normalise like one
        # process over N*B pairs; params["global_minmax"] = False keeps it per rank.  The default is resolved at forward
        # time (the ``global_minmax`` property), so a model built before init_process_group normalises globally too.
        # The reduction is a COLLECTIVE: every rank must run the same number of forwards (training and validation
        # alike) -- train_icp_weights.fit() validates on all ranks for this reason.
        gm = params.get("global_minmax", None)
        self._global_minmax = None if gm is None else bool(gm)
        self._step = 0

        self.mean_num_pts = 0.0
        self.max_w = 0.0
        self.min_w = 0.0
        self.mean_w = 0.0
        self.mean_all_pts = 0.0

        init_c_num = network_inputs["fft"] + network_inputs["cfar"] + network_inputs["range"]
        enc_channels = [init_c_num, 8, 16, 32, 64, 128, 256]
        dec_channels = [256, 128, 64, 32, 16, 8]
        self.encoder = ModuleList([self.conv_block(enc_channels[i], enc_channels[i + 1], i)
                                   for i in range(len(enc_channels) - 1)])
        self.decoder = ModuleList([self.conv_block(dec_channels[i], dec_channels[i + 1])
                                   for i in range(len(dec_channels) - 1)])
        self.final_layer = nn.Sequential(nn.Conv2d(dec_channels[-1], 1, kernel_size=1), nn.Sigmoid())
        if params["init_weights"]:
            self.apply(weights_init)

    @property
    def global_minmax(self):
        """Whether the min-max extrema are reduced over the ranks: the explicit params["global_minmax"] / assigned value,
        else "this process is a rank of a multi-rank job" -- asked when used, not when the model was built."""
        if self._global_minmax is not None:
            return self._global_minmax
        import torch.distributed as dist
        return bool(dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)

    @global_minmax.setter
    def global_minmax(self, value):
        self._global_minmax = None if value is None else bool(value)

    def conv_block(self, in_channels, out_channels, i=0):
        """icp_weight_policy.py:104-125 (module order fixes the state_dict keys)."""
        relu_layer = nn.LeakyReLU(0.1) if self.leaky else nn.ReLU()
        modules = [nn.Conv2d(in_channels, out_channels, kernel_size=3, padding=1), relu_layer]
        if self.batch_norm:
            modules.append(nn.BatchNorm2d(out_channels))
        modules.append(nn.Conv2d(out_channels, out_channels, kernel_size=3, padding=1))
        modules.append(relu_layer)
        if self.batch_norm:
            modules.append(nn.BatchNorm2d(out_channels))
        if self.dropout > 0.0:
            modules.append(nn.Dropout(p=self.dropout))
        if i > 0:
            modules.append(nn.MaxPool2d(kernel_size=2, stride=2))
        return nn.Sequential(*modules)

    # ------------------------------------------------------------------ U1-U4
    def _network_input(self, fft_data, fft_cfar, normalize=True):
        """icp_weight_policy.py:136-159.  ``normalize=False`` stops before the per-channel normalisation
        (the hand-written first layer applies the min-max form while it loads the image)."""
        input_data = None
        if self.network_inputs["fft"]:
            input_data = fft_data.unsqueeze(1)
        if self.network_inputs["cfar"]:
            input_data = torch.cat([input_data, fft_cfar.unsqueeze(1)], dim=1)
        if self.network_inputs["range"]:
            rm = self.range_mask.to(input_data.device)
            range_stack = rm.unsqueeze(0).expand(input_data.shape[0], -1, -1).unsqueeze(1)
            input_data = torch.cat([input_data, range_stack], dim=1)
        if self.log_transform:
            input_data = torch.log(input_data + 1e-6)
        if not normalize:
            return input_data
        return self._normalize_channels(input_data)

    def _normalize_channels(self, input_data):
        chans = []
        for c in range(input_data.shape[1]):
            xc = input_data[:, c, :, :]
            if "minmax" in self.normalize_type:
                c_max, c_min = torch.max(xc), torch.min(xc)
                if self.global_minmax:
                    c_min, c_max = _global_extrema(c_min, c_max)
                xc = (xc - c_min) / (c_max - c_min)
            elif "standardize" in self.normalize_type:
                xc = (xc - torch.mean(xc)) / torch.std(xc)
            chans.append(xc)
        return torch.stack(chans, dim=1)

    def _unet(self, input_data):
        """icp_weight_policy.py:161-184 through the nn.Module tree (fp32, CPU: tests only)."""
        enc_layers = []
        for layer in self.encoder:
            enc_layers.append(input_data)
            input_data = layer(input_data)
        enc_layers.reverse()
        for i, decoder_layer in enumerate(self.decoder):
            skip_con = enc_layers[i]
            input_data = nn.functional.interpolate(input_data, size=(skip_con.shape[2], skip_con.shape[3]),
                                                   mode="bilinear", align_corners=True)
            input_data = decoder_layer(input_data)
            input_data = torch.cat([skip_con, input_data], dim=1)
            input_data = decoder_layer(input_data)   # same weights applied twice (:178,182)
        logits = self.final_layer[0](input_data)
        return torch.sigmoid(logits.float()).squeeze(1)

    def forward(self, batch_scan, batch_map, T_init, binary=False, override_mask=None, neptune_run=None, epoch=0,
                batch_idx=0, mask_only=False):
        fft_data = batch_scan["fft_data"].to(self.device)
        fft_cfar = batch_scan["fft_cfar"].to(self.device)
        scan_pc_raw = batch_scan["raw_pc"].to(self.device)
        map_pc = batch_map["pc"].to(self.device)

        if override_mask is None:
            raw_in = self._network_input(fft_data, fft_cfar, normalize=False)
            if self.unet_backend == "hip":
                from . import _lib, unet_hip
                if not raw_in.is_cuda:
                    raise _lib.MmkError("the mask U-Net is a set of HIP kernels with no CPU path: params['device'] must be "
                                        "a HIP device (unet_backend='torch' is the CPU host-logic mirror used by tests)")
                # any image of at least 32 x 32 (five floor-rounding poolings leave >= 1 pixel): the Cartesian
                # 640 x 640 grid and the polar 400 x 3360 one (network_input_type "polar") alike
                if raw_in.shape[1] > 4 or raw_in.shape[2] < 32 or raw_in.shape[3] < 32:
                    raise _lib.MmkError("mask U-Net: unsupported network input %s (1..4 channels, at least 32 x 32)"
                                        % (tuple(raw_in.shape),))
                self._step += 1
                if "minmax" in self.normalize_type:
                    # min-max normalisation folded into the first layer's loads: one min/max pass, no
                    # normalised copy of the image
                    net_in = raw_in.contiguous().float()
                    pre = unet_hip.channel_minmax(net_in, global_reduce=self.global_minmax)
                else:
                    net_in, pre = self._normalize_channels(raw_in), None
                # (the amax normalisation below rides inside the same autograd node)
                if self.batch_norm:        # Conv, ReLU, BN, Conv, ReLU, BN: BatchNorm kernels between the convolutions
                    from . import unet_hip_bn
                    weight_mask = unet_hip_bn.unet_mask(self, net_in, self.training, self._step, norm=self.norm_weights, pre=pre,
                                                        slope=0.1 if self.leaky else 0.0)
                else:
                    weight_mask = unet_hip.unet_mask(self, net_in, self.training, self._step, norm=self.norm_weights, pre=pre,
                                                     slope=0.1 if self.leaky else 0.0)
                normalised = self.norm_weights
            else:
                if raw_in.is_cuda:
                    from . import _lib
                    raise _lib.MmkError("unet_backend='torch' is the CPU host-logic mirror (tests only); on a HIP device "
                                        "the U-Net runs on the hand-written kernels (unet_backend='hip')")
                weight_mask = self._unet(self._normalize_channels(raw_in))
                normalised = False
        else:
            weight_mask = override_mask.to(self.device)
            normalised = False

        if self.norm_weights and not normalised:
            weight_mask = weight_mask / torch.amax(weight_mask, dim=(1, 2), keepdim=True)
        if binary:
            weight_mask = torch.where(weight_mask > 0.5, 1.0, 0.0)
        if mask_only:
            return weight_mask

        (weights, diff_mean_num_non0, mean_num_non0, mean_w, max_w, min_w), stats = \
            _extract_weights_stats(weight_mask, scan_pc_raw)
        self.mean_num_pts = mean_num_non0
        self.max_w = max_w
        self.min_w = min_w
        self.mean_w = mean_w
        # points with x != 0 and y != 0, per scan (icp_weight_policy.py:209-212): same fused pass
        self.mean_all_pts = stats[5]

        scan_pc_filt = batch_scan["filtered_pc"].to(self.device)

        if self.training and not self.use_ICP_4_train:
            return T_init, weight_mask, diff_mean_num_non0

        T_est = self.icp(scan_pc_filt, map_pc, T_init, weights)
        return T_est, weight_mask, diff_mean_num_non0

    def icp(self, scan_pc, map_pc, T_init, weights):
        """icp_weight_policy.py:277-288 (loss_fn / trim_dist / dim hard-coded there;
        overridable here through optional params keys)."""
        alg = self.ICP_alg if self.training else self.ICP_alg_inference
        icp_result = alg.icp(scan_pc, map_pc, T_init=T_init, weight=weights, trim_dist=self.icp_trim_dist,
                             loss_fn=self.icp_loss_fn, dim=self.icp_dim)
        return icp_result["T"]
