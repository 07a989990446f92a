"""Does the number of HIP streams a process has touched change the U-Net backward's time?  (The backward runs its weight-gradient
launches on a side stream; HIP maps streams onto a limited number of hardware queues -- GPU_MAX_HW_QUEUES, default 4 -- and streams
that share a queue serialise.)  Prints forward / backward ms of the U-Net alone, B = 32, after touching 0, 1, 2, 3 extra streams.
  python scripts/streams_probe.py            (run it again with GPU_MAX_HW_QUEUES=8 in the environment)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm_masking_amd import synthetic  # noqa: E402
from mm_masking_amd import train_icp_weights as trn  # noqa: E402
from mm_masking_amd.icp_weight_policy import LearnICPWeightPolicy  # noqa: E402

dev = torch.device("cuda:0")
params = trn.default_params(dev)
params.update({"icp_type": "pt2pl", "icp_loss_fn": {"name": "huber", "metric": 1.0}, "dropout": 0.05})
torch.manual_seed(0)
model = LearnICPWeightPolicy(params).to(dev)
model.train()
raw = synthetic.make_batch(list(range(32)), device=dev)
batch = trn.prepare_batch(raw, params, max_loc_pts=5120)


def measure(reps=6):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    f, b = [], []
    for _ in range(reps):
        model.zero_grad(set_to_none=True)
        ev[0].record()
        mask = model(batch["loc_data"], batch["map_data"], None, mask_only=True)
        ev[1].record()
        mask.sum().backward()
        ev[2].record()
        torch.cuda.synchronize()
        f.append(ev[0].elapsed_time(ev[1]))
        b.append(ev[1].elapsed_time(ev[2]))
    return min(f[2:]), min(b[2:])


print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"))
extra = []
for n in range(4):
    print("extra streams touched: %d   fwd %.3f ms  bwd %.3f ms" % ((n,) + measure()), flush=True)
    s = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(s):
        torch.zeros(16, device=dev).add_(1.0)
    extra.append(s)
    torch.cuda.synchronize()
