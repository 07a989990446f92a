"""ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Functional fp32 PyTorch-CPU restatement of the mask U-Net
(/root/reference/mm_masking/icp_weight_policy.py:84-99 topology, :136-199
forward; SURVEY.md §8a Group U).  It works directly on a ``state_dict`` with the
reference's key names (encoder.{0..5}.{0,2}, decoder.{0..4}.{0,2},
final_layer.0) so the same checkpoint drives the oracle, the reference and the
product module.  Pinned by tests/golden/unet.npz.
"""
import torch
import torch.nn.functional as F

ENC_CH = [None, 8, 16, 32, 64, 128, 256]
DEC_CH = [256, 128, 64, 32, 16, 8]


def init_state_dict(in_ch, seed):
    """Xavier-uniform weights / zero biases in the reference's construction order
    (icp_weight_policy.py:15-22,88-102), so a given torch seed reproduces the
    reference module's parameters."""
    torch.manual_seed(seed)
    sd = {}
    enc = [in_ch] + ENC_CH[1:]
    layers = []
    for i in range(6):
        layers.append(("encoder.%d" % i, enc[i], enc[i + 1]))
    for i in range(5):
        layers.append(("decoder.%d" % i, DEC_CH[i], DEC_CH[i + 1]))
    # construction draws default-init numbers first (advancing the RNG), then
    # weights_init re-draws xavier in module order; reproduce both passes.
    for name, ci, co in layers:
        for sub, c_in in (("0", ci), ("2", co)):
            torch.nn.Conv2d(c_in, co, 3, padding=1)
    torch.nn.Conv2d(DEC_CH[-1], 1, 1)
    for name, ci, co in layers:
        for sub, c_in in (("0", ci), ("2", co)):
            w = torch.empty(co, c_in, 3, 3)
            torch.nn.init.xavier_uniform_(w)
            sd["%s.%s.weight" % (name, sub)] = w
            sd["%s.%s.bias" % (name, sub)] = torch.zeros(co)
    w = torch.empty(1, DEC_CH[-1], 1, 1)
    torch.nn.init.xavier_uniform_(w)
    sd["final_layer.0.weight"] = w
    sd["final_layer.0.bias"] = torch.zeros(1)
    return sd


def _block(x, sd, name, leaky, pool, dropout_p=0.0, training=False):
    act = (lambda t: F.leaky_relu(t, 0.1)) if leaky else F.relu
    x = act(F.conv2d(x, sd[name + ".0.weight"], sd[name + ".0.bias"], padding=1))
    x = act(F.conv2d(x, sd[name + ".2.weight"], sd[name + ".2.bias"], padding=1))
    if dropout_p > 0.0:
        x = F.dropout(x, dropout_p, training)
    if pool:
        x = F.max_pool2d(x, 2, 2)
    return x


def assemble_input(fft, cfar=None, range_mask=None, log_transform=False, normalize=("minmax",)):
    """U1: icp_weight_policy.py:136-159 (batch-global per-channel normalisation)."""
    chans = [fft.unsqueeze(1)]
    if cfar is not None:
        chans.append(cfar.unsqueeze(1))
    if range_mask is not None:
        chans.append(range_mask.unsqueeze(0).expand(fft.shape[0], -1, -1).unsqueeze(1))
    x = torch.cat(chans, dim=1)
    if log_transform:
        x = torch.log(x + 1e-6)
    outs = []
    for c in range(x.shape[1]):
        xc = x[:, c]
        if "minmax" in normalize:
            xc = (xc - xc.min()) / (xc.max() - xc.min())
        elif "standardize" in normalize:
            xc = (xc - xc.mean()) / xc.std()
        outs.append(xc)
    return torch.stack(outs, dim=1)


def unet_mask(x, sd, leaky=False, norm_weights=True, dropout_p=0.0, training=False):
    """U2-U4: encoder, twice-applied decoder blocks, 1x1 + sigmoid, amax normalise."""
    skips = []
    for i in range(6):
        skips.append(x)
        x = _block(x, sd, "encoder.%d" % i, leaky, pool=(i > 0), dropout_p=dropout_p, training=training)
    skips.reverse()
    for i in range(5):
        skip = skips[i]
        x = F.interpolate(x, size=skip.shape[2:], mode="bilinear", align_corners=True)
        x = _block(x, sd, "decoder.%d" % i, leaky, False, dropout_p, training)
        x = torch.cat([skip, x], dim=1)
        x = _block(x, sd, "decoder.%d" % i, leaky, False, dropout_p, training)
    m = torch.sigmoid(F.conv2d(x, sd["final_layer.0.weight"], sd["final_layer.0.bias"])).squeeze(1)
    if norm_weights:
        m = m / torch.amax(m, dim=(1, 2), keepdim=True)
    return m
