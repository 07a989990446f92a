"""Is the host ahead of the GPU where the U-Net backward is enqueued?  Wraps _UNetNative.backward / forward and the ICP autograd
hooks: at entry, stream.query() (True = the GPU has nothing left to do: the host is the bottleneck there) and the host time spent
inside.  Development tool (GPU box)."""
import sys, time
sys.path.insert(0, ".")
import torch
from mm_masking_amd import synthetic, unet_hip
from mm_masking_amd import train_icp_weights as trn
import importlib
icpmod = importlib.import_module("mm_masking_amd.dICP.ICP")
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
log = []
def wrap(cls, name, tag):
    orig = getattr(cls, name)
    def f(*a, **k):
        idle = torch.cuda.current_stream(DEV).query()
        t0 = time.perf_counter()
        r = orig(*a, **k)
        log.append((tag, idle, (time.perf_counter() - t0) * 1e6))
        return r
    setattr(cls, name, staticmethod(f))
wrap(unet_hip._UNetNative, "forward", "unet.fwd")
wrap(unet_hip._UNetNative, "backward", "unet.bwd")
wrap(icpmod._IcpFunction, "forward", "icp.fwd")
wrap(icpmod._IcpFunction, "backward", "icp.bwd")
def step():
    t0 = time.perf_counter()
    batch = trn.prepare_batch(raw, params)
    t1 = time.perf_counter()
    out = trn.train_step(model, batch, opt, lw, DEV)
    log.append(("step host", None, (time.perf_counter() - t0) * 1e6))
    log.append(("prepare host", None, (t1 - t0) * 1e6))
    return out
for _ in range(5):
    step()
torch.cuda.synchronize()
log.clear()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(6):
    step()
e1.record()
torch.cuda.synchronize()
print("gpu ms/step %.2f" % (e0.elapsed_time(e1) / 6))
for tag in ("prepare host", "unet.fwd", "icp.fwd", "icp.bwd", "unet.bwd", "step host"):
    rows = [r for r in log if r[0] == tag]
    print("%-13s host us %s   stream idle at entry: %s" % (tag, " ".join("%.0f" % r[2] for r in rows), [r[1] for r in rows]))
