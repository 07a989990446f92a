"""Diagnostic: the batch-norm network on the HIP kernels vs the fp32 nn.Module tree on the CPU, block by block."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from mm_masking_amd import train_icp_weights as trn, unet_hip as uh
from mm_masking_amd.icp_weight_policy import LearnICPWeightPolicy
DEV = torch.device("cuda:0")
p = trn.default_params(DEV); p.update({"dropout": 0.0, "batch_norm": True})
torch.manual_seed(1234)
mh = LearnICPWeightPolicy(p).to(DEV); mh.train()
mc = LearnICPWeightPolicy(dict(p, device=torch.device("cpu"), unet_backend="torch")); mc.load_state_dict({k: v.cpu() for k, v in mh.state_dict().items()}); mc.train()
H = 64
xin = np.random.default_rng(99).uniform(0.01, 1, size=(2, H, H)).astype(np.float32)
scan = {"fft_data": torch.from_numpy(xin), "fft_cfar": torch.zeros(2, H, H), "raw_pc": torch.zeros(2, 4, 3)}
uh.DEBUG = {}
m = mh(scan, {"pc": torch.zeros(2, 4, 6)}, None, mask_only=True)
fw = uh.DEBUG["fwd_bn"]; uh.DEBUG = None
# CPU: replay with hooks
x = mc._normalize_channels(mc._network_input(torch.from_numpy(xin), torch.zeros(2, H, H), normalize=False))
enc = []
cur = x
for i, layer in enumerate(mc.encoder):
    enc.append(cur)
    # inside block: record first conv+relu, bn
    a = layer[1](layer[0](cur)); y = layer[2](a)
    got = fw["saved"][("e", i)]
    print("enc%d aA max|diff| %.4f (max %.3f)  yA %.4f" % (i, (got[2].float().cpu().permute(0,3,1,2) - a).abs().max().item(), a.abs().max().item(), (got[3].float().cpu().permute(0,3,1,2) - y).abs().max().item()))
    cur = layer(cur)
    print("   t[%d] %.4f (max %.3f)" % (i, (fw["t"][i].float().cpu().permute(0,3,1,2) - cur).abs().max().item(), cur.abs().max().item()))
enc.reverse()
for j, dl in enumerate(mc.decoder):
    skip = enc[j]
    cur = torch.nn.functional.interpolate(cur, size=skip.shape[2:], mode="bilinear", align_corners=True)
    cur = dl(cur)
    got = fw["saved"][("d", j, 0)][5]
    print("dec%d first d %.4f (max %.3f)" % (j, (got.float().cpu().permute(0,3,1,2) - cur).abs().max().item(), cur.abs().max().item()))
    cur = torch.cat([skip, cur], 1); cur = dl(cur)
    got = fw["saved"][("d", j, 1)][5]
    print("dec%d second d %.4f (max %.3f)" % (j, (got.float().cpu().permute(0,3,1,2) - cur).abs().max().item(), cur.abs().max().item()))
print("---- stats detail")
cur = x
for i, layer in enumerate(mc.encoder):
    a = layer[1](layer[0](cur))
    got = fw["saved"][("e", i)]
    ah = got[2].float().cpu().permute(0, 3, 1, 2)
    sA = got[6].cpu()
    mean_h, istd_h = sA[:, 0], sA[:, 1]
    mean_c = ah.mean(dim=(0, 2, 3)); var_c = ah.var(dim=(0, 2, 3), unbiased=False)
    istd_c = 1 / torch.sqrt(var_c + 1e-5)
    yh = got[3].float().cpu().permute(0, 3, 1, 2)
    y_from_h = (ah - mean_h[None, :, None, None]) * istd_h[None, :, None, None]
    print("enc%d: stat mean diff %.2e istd rel diff %.2e ; y vs own-stats %.4f ; min var %.2e" % (i, (mean_h - mean_c).abs().max(), ((istd_h - istd_c) / istd_c).abs().max(), (yh - y_from_h).abs().max(), var_c.min()))
    cur = layer(cur)
