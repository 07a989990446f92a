"""Correspondences of the dICP forward over many pair sets (the single-GPU bench batches AND the shards a 2-rank run gives each
rank), twice per set: prints, per set, a checksum of the correspondences of the rows that were active (pairs that have converged
keep stale rows), the number of run-to-run differences, and the number of indices that sit on the clamp of
icp_accumulate_kernel (index 0 / M - 1: an unarmed key would land there).  Run once per build / NN engine and diff the output:
    MMK_LIB=... python scripts/nn_sweep.py > a.txt   (compare two builds through MMK_LIB)"""
import os, sys, hashlib
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mm_masking_amd import ddp, synthetic, train_icp_weights as trn
from mm_masking_amd.dICP.ICP import ICP
B = 32
DIM = int(sys.argv[1]) if len(sys.argv) > 1 else 2      # 3: SE(3) scenes (synthetic.make_pair(dim=3)), nn_mfma_kernel<3>
dev = torch.device("cuda:0")
params = trn.default_params(dev)
sets = {"0..31": list(range(32)), "32..63": list(range(32, 64)), "4000..": list(range(4000, 4032))}
for r in range(2):
    for i in range(2):
        sets["rank%d/2 batch%d" % (r, i)] = list(ddp.shard_indices(B * 2, r, 2, start=i * B * 2))
for r in (0, 3, 7):
    sets["rank%d/8 batch0" % r] = list(ddp.shard_indices(B * 8, r, 8, start=0))
icp = ICP("pt2pl", differentiable=True, max_iterations=10, tolerance=1e-5)
tot_diff = 0
for name, ids in sets.items():
    raw = synthetic.make_batch(ids, device=dev, dim=DIM)
    batch = trn.prepare_batch(raw, params, max_loc_pts=5120)
    src = batch["loc_data"]["filtered_pc"]
    torch.manual_seed(5)
    w0 = torch.rand(B, src.shape[1], device=dev)
    outs = []
    for rep in range(2):
        w = w0.clone().requires_grad_(True)
        T = icp.icp(src, raw["map_pc"], T_init=raw["T_init"], weight=w, trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0}, dim=DIM)["T"]
        sv = T.grad_fn.saved_tensors
        idx, act = sv[3].cpu().numpy(), sv[7].cpu().numpy()[:-1]
        idx = np.where(act[:, :, None] != 0, idx, -7)
        outs.append((idx, T.detach().cpu().numpy().copy()))
        x = torch.randn(2048, 2048, device=dev); (x @ x).sum().item()
    M = raw["map_pc"].shape[1]
    d = int((outs[0][0] != outs[1][0]).sum())
    tot_diff += d
    edge = int(((outs[0][0] == 0) | (outs[0][0] == M - 1)).sum())
    print("%-16s idx %s T %s run-to-run diffs %d (T equal %s) clamp-edge indices %d" % (
        name, hashlib.md5(outs[0][0].tobytes()).hexdigest()[:12], hashlib.md5(outs[0][1].tobytes()).hexdigest()[:12], d,
        np.array_equal(outs[0][1], outs[1][1]), edge), flush=True)
print("total run-to-run diffs", tot_diff)
