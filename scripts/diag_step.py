"""Diagnostic: time each phase of the bench step, logging progressively."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LOG = open(os.path.join(ROOT, "gpurun_out", "diag.log"), "a")
T0 = time.time()
def log(*a):
    msg = "[%7.2f] " % (time.time() - T0) + " ".join(str(x) for x in a)
    print(msg, flush=True); LOG.write(msg + "\n"); LOG.flush()
from mm_masking_amd import synthetic, train_icp_weights as trn
from mm_masking_amd.icp_weight_policy import LearnICPWeightPolicy
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
params = trn.default_params(dev)
params.update({"icp_type": "pt2pl", "icp_loss_fn": {"name": "huber", "metric": 1.0}, "max_iter": 10})
log("generating", B, "pairs")
raw = synthetic.make_batch(list(range(B)), device=dev)
log("generated")
torch.manual_seed(0)
model = LearnICPWeightPolicy(params).to(dev); model.train()
opt = trn.make_optimizer(model, params); lw = trn.loss_weights_from(params)
def sync(): torch.cuda.synchronize()
for it in range(4):
    t = time.time(); batch = trn.prepare_batch(raw, params); sync(); log("it", it, "prepare", time.time() - t)
    t = time.time(); opt.zero_grad(); T, mask, n0 = model(batch["loc_data"], batch["map_data"], raw["T_init"]); sync(); log("it", it, "forward", time.time() - t)
    t = time.time(); loss, _ = trn.eval_training_loss(T, mask, n0, raw["T_gt"], batch["loc_data"], batch["map_data"], model, loss_weights=lw); sync(); log("it", it, "loss", time.time() - t, float(loss))
    t = time.time(); loss.backward(); sync(); log("it", it, "backward", time.time() - t)
    t = time.time(); opt.step(); sync(); log("it", it, "optstep", time.time() - t)
# finer: U-Net only vs ICP only
x = batch["loc_data"]["fft_data"]
for it in range(3):
    t = time.time(); m = model(batch["loc_data"], batch["map_data"], None, mask_only=True); sync(); log("unet fwd", time.time() - t)
    t = time.time(); m.sum().backward(); sync(); log("unet bwd", time.time() - t)
w = torch.rand(B, 5120, device=dev, requires_grad=True)
for it in range(3):
    t = time.time(); Ti = model.ICP_alg.icp(batch["loc_data"]["filtered_pc"], raw["map_pc"], T_init=raw["T_init"], weight=w, trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0}, dim=2)["T"]; sync(); log("icp fwd", time.time() - t)
    t = time.time(); Ti.sum().backward(); sync(); log("icp bwd", time.time() - t)
log("done")
