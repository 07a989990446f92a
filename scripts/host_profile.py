"""Where the host time of a step goes (development tool): cProfile over a few steps, no syncs."""
import cProfile, pstats, sys, io
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
def step(i):
    batch = trn.prepare_batch(raws[i % 2], params, max_loc_pts=5120)
    return trn.train_step(model, batch, opt, lw, dev)
for i in range(4): step(i)
torch.cuda.synchronize()
pr = cProfile.Profile()
import time
t0 = time.perf_counter()
pr.enable()
for i in range(10): step(i)
pr.disable()
t1 = time.perf_counter()
torch.cuda.synchronize()
print("host ms/step (with profiler overhead): %.1f" % ((t1 - t0) / 10 * 1e3))
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
