"""Allocator behaviour over training steps (development tool): allocated / reserved bytes, device mallocs."""
import sys, gc
sys.path.insert(0, ".")
import torch
from mm_masking_amd import train_icp_weights as trn, synthetic, ddp
from mm_masking_amd.icp_weight_policy import LearnICPWeightPolicy
dev = torch.device("cuda:0")
params = trn.default_params(dev)
lw = trn.loss_weights_from(params)
model = LearnICPWeightPolicy(params).to(dev); model.train()
opt = trn.make_optimizer(model, params)
B = 32
raws = [synthetic.make_batch(ddp.shard_indices(B, 0, 1, start=i * B), device=dev, m_valid=20000, m_pad=20480) for i in range(2)]
def st():
    m = torch.cuda.memory_stats(dev)
    return (m["allocated_bytes.all.current"] / 2**30, m["allocated_bytes.all.peak"] / 2**30, m["reserved_bytes.all.current"] / 2**30, m["num_device_alloc"])
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
    batch = trn.prepare_batch(raws[i % 2], params, max_loc_pts=5120)
    trn.train_step(model, batch, opt, lw, dev)
    del batch
    if i % 3 == 0:
        torch.cuda.synchronize()
        print("step %2d allocated %.2f GiB peak %.2f reserved %.2f device mallocs %d gc %s" % ((i,) + st() + (gc.get_count(),)), flush=True)
