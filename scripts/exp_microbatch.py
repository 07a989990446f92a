"""Experiment: one training step of B = 32 as two micro-batches of 16 on two streams, staggered by one U-Net forward, so
that the VALU-bound NN launches of one micro-batch run beside the byte-bound U-Net kernels of the other
(DESIGN.md §8 "Next" 1).  Prints ms/step for the plain step and for the pipelined one.  Development tool."""
import sys, time
import torch
sys.path.insert(0, ".")
from mm_masking_amd import ddp, synthetic
from mm_masking_amd import train_icp_weights as trn
from mm_masking_amd.icp_weight_policy import LearnICPWeightPolicy

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
params = trn.default_params(dev)
params.update({"icp_type": "pt2pl", "icp_loss_fn": {"name": "huber", "metric": 1.0}, "icp_dim": 2, "max_iter": 10, "dropout": 0.05})
lw = trn.loss_weights_from(params)
torch.manual_seed(1234)
model = LearnICPWeightPolicy(params).to(dev)
model.train()
opt = trn.make_optimizer(model, params)
B = 32
raws = [synthetic.make_batch(ddp.shard_indices(B, 0, 1, start=i * B), device=dev, m_valid=20000, m_pad=20480) for i in range(2)]


def half(batch, lo, hi):
    def cut(d):
        return {k: (v[lo:hi] if torch.is_tensor(v) and v.dim() > 0 else v) for k, v in d.items()}
    return {"loc_data": cut(batch["loc_data"]), "map_data": cut(batch["map_data"]), "transforms": cut(batch["transforms"])}


def plain(i):
    batch = trn.prepare_batch(raws[i % 2], params, max_loc_pts=5120)
    return trn.train_step(model, batch, opt, lw, dev)


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def piped(i):
    batch = trn.prepare_batch(raws[i % 2], params, max_loc_pts=5120)
    opt.zero_grad()
    cur = torch.cuda.current_stream()
    hb = [half(batch, 0, B // 2), half(batch, B // 2, B)]
    s1.wait_stream(cur); s2.wait_stream(cur)
    e1 = torch.cuda.Event()
    with torch.cuda.stream(s1):
        m1 = model(hb[0]["loc_data"], hb[0]["map_data"], None, mask_only=True)
        e1.record(s1)
    with torch.cuda.stream(s2):
        s2.wait_event(e1)
        m2 = model(hb[1]["loc_data"], hb[1]["map_data"], None, mask_only=True)
    losses = []
    for s, h, m in ((s1, hb[0], m1), (s2, hb[1], m2)):
        with torch.cuda.stream(s):
            T0 = h["transforms"]["T_ml_init"]
            T, mask, nn0 = model(h["loc_data"], h["map_data"], T0, override_mask=m)
            loss, _ = trn.eval_training_loss(T, mask, nn0, h["transforms"]["T_ml_gt"], h["loc_data"], h["map_data"], model, loss_weights=lw)
            (0.5 * loss).backward()
            losses.append(loss.detach())
    cur.wait_stream(s1); cur.wait_stream(s2)
    opt.step()
    return losses


def timeit(fn, n=20):
    for i in range(6):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print("plain     %.2f ms/step" % timeit(plain))
print("pipelined %.2f ms/step" % timeit(piped))
print("plain     %.2f ms/step" % timeit(plain))
print("pipelined %.2f ms/step" % timeit(piped))
