"""Which PyTorch operators still launch kernels inside a training step (fills, copies, element-wise glue), with shapes and the
source line that issues them: torch.profiler over 3 steps at the bench shape.  Development tool (GPU box)."""
import sys
sys.path.insert(0, ".")
import torch
from torch.profiler import profile, ProfilerActivity
from mm_masking_amd import synthetic
from mm_masking_amd import train_icp_weights as trn
DEV = torch.device("cuda:0")
params = trn.default_params(DEV)
params.update({"icp_type": "pt2pl", "icp_loss_fn": {"name": "huber", "metric": 1.0}, "max_iter": 10})
B = 32
raw = synthetic.make_batch(list(range(B)), device=DEV)
torch.manual_seed(0)
model = trn.LearnICPWeightPolicy(params).to(DEV)
opt = trn.make_optimizer(model, params)
model.train()
lw = trn.loss_weights_from(params)
def step():
    batch = trn.prepare_batch(raw, params)
    return trn.train_step(model, batch, opt, lw, DEV)
for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    for _ in range(3):
        step()
    torch.cuda.synchronize()
rows = []
for ev in prof.key_averages(group_by_input_shape=True, group_by_stack_n=6):
    cuda = getattr(ev, "self_device_time_total", 0) or getattr(ev, "self_cuda_time_total", 0)
    if cuda <= 0 or not ev.key.startswith("aten::"):
        continue
    stack = [s for s in ev.stack if "mm_masking_amd" in s or "bench" in s][:2]
    rows.append((cuda / 3.0, ev.count / 3.0, ev.key, str(ev.input_shapes)[:60], " <- ".join(s.split("/")[-1][:70] for s in stack)))
rows.sort(reverse=True)
print("us/step  calls/step  op  shapes  where")
for r in rows[:40]:
    print("%7.1f %6.1f  %-22s %-60s %s" % r)
print("total aten device time per step: %.1f us" % sum(r[0] for r in rows))
