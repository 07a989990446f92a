"""What makes the U-Net backward slower in `bench.py --force-dist` (one rank on RCCL)?  One process, the U-Net alone at B = 32,
forward / backward ms after each of: nothing; the backward's bucket events armed; an RCCL process group initialised; one
all-reduce issued; a communication stream that waits for the bucket events and all-reduces (the overlapped path).
  python scripts/dist_probe.py"""
import os
import sys

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm_masking_amd import ddp, synthetic, unet_hip  # noqa: E402
from mm_masking_amd import train_icp_weights as trn  # noqa: E402
from mm_masking_amd.icp_weight_policy import LearnICPWeightPolicy  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
params = trn.default_params(dev)
params.update({"icp_type": "pt2pl", "icp_loss_fn": {"name": "huber", "metric": 1.0}, "dropout": 0.05, "global_minmax": False})
torch.manual_seed(0)
model = LearnICPWeightPolicy(params).to(dev)
model.train()
raw = synthetic.make_batch(list(range(32)), device=dev)
batch = trn.prepare_batch(raw, params, max_loc_pts=5120)


def measure(tag, sync=None, reps=7):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    f, b, c = [], [], []
    for _ in range(reps):
        if sync is not None:
            sync.zero_grad()
        else:
            model.zero_grad(set_to_none=True)
        ev[0].record()
        mask = model(batch["loc_data"], batch["map_data"], None, mask_only=True)
        ev[1].record()
        mask.sum().backward()
        ev[2].record()
        if sync is not None:
            sync()
        ev[3].record()
        torch.cuda.synchronize()
        f.append(ev[0].elapsed_time(ev[1]))
        b.append(ev[1].elapsed_time(ev[2]))
        c.append(ev[2].elapsed_time(ev[3]))
    print("%-58s fwd %.3f  bwd %.3f  sync %.3f ms" % (tag, min(f[2:]), min(b[2:]), min(c[2:])), flush=True)


measure("plain")
evs = [torch.cuda.Event() for _ in range(3)]
for e in evs:
    e.record()
unet_hip.GRAD_BUCKET_EVENTS = evs
measure("bucket events armed (mmk_unet_backward_buckets)")
unet_hip.GRAD_BUCKET_EVENTS = None
measure("plain again")
dist.init_process_group("nccl", device_id=dev)
measure("RCCL process group initialised")
t = torch.zeros(1 << 20, device=dev)
dist.all_reduce(t)
torch.cuda.synchronize()
measure("after one all-reduce (communicator built)")
measure("FlatGradSync(overlap=False, force)", ddp.FlatGradSync(model, overlap=False, force_collective=True))
measure("FlatGradSync(overlap=True, force)", ddp.FlatGradSync(model, overlap=True, force_collective=True))
unet_hip.GRAD_BUCKET_EVENTS = None
measure("plain at the end")
dist.destroy_process_group()
