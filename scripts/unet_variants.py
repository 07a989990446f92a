"""Diagnostic: U-Net forward+backward time at B=32, 640x640 under different PyTorch/MIOpen settings."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mm_masking_amd import train_icp_weights as trn
from mm_masking_amd.icp_weight_policy import LearnICPWeightPolicy
dev = torch.device("cuda:0")
B = 32
x = torch.rand(B, 640, 640, device=dev)
scan = {"fft_data": x, "fft_cfar": x, "raw_pc": torch.zeros(B, 4, 3, device=dev)}
mp = {"pc": torch.zeros(B, 4, 6, device=dev)}
def run(tag, **over):
    p = trn.default_params(dev); p.update(over)
    torch.manual_seed(0)
    m = LearnICPWeightPolicy(p).to(dev); m.train()
    if over.get("benchmark"):
        torch.backends.cudnn.benchmark = True
    ts = []
    for it in range(6):
        torch.cuda.synchronize(); t = time.time()
        mask = m(scan, mp, None, mask_only=True)
        torch.cuda.synchronize(); t1 = time.time()
        mask.sum().backward()
        torch.cuda.synchronize(); t2 = time.time()
        ts.append((t1 - t, t2 - t1))
    f = min(a for a, b in ts[2:]); b = min(b for a, b in ts[2:])
    print("%-40s fwd %.1f ms bwd %.1f ms  (first iter %.1f s) NHWC_env=%s" % (tag, f * 1e3, b * 1e3, sum(ts[0]), os.environ.get("PYTORCH_MIOPEN_SUGGEST_NHWC")), flush=True)
    torch.backends.cudnn.benchmark = False
which = sys.argv[1] if len(sys.argv) > 1 else "all"
run("bf16 autocast channels_last", )
run("bf16 autocast NCHW", channels_last=False)
run("fp32 channels_last", amp_dtype=torch.float32)
run("fp32 NCHW", amp_dtype=torch.float32, channels_last=False)
if which == "bench":
    run("bf16 channels_last benchmark=True", benchmark=True)
    run("bf16 NCHW benchmark=True", channels_last=False, benchmark=True)
