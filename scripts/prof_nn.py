"""Profiling driver: the dICP forward+backward alone at the bench shape (B=32, N=5120,
M=20480 padded, dim 2, pt2pl Huber, 10 iterations), for rocprofv3 --pmc passes on the
nn_search kernel (bench.py profiles the whole step)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mm_masking_amd import synthetic, train_icp_weights as trn
from mm_masking_amd.dICP.ICP import ICP
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
params = trn.default_params(dev)
raw = synthetic.make_batch(list(range(B)), device=dev)
batch = trn.prepare_batch(raw, params)
src = batch["loc_data"]["filtered_pc"]
icp = ICP("pt2pl", differentiable=True, max_iterations=10, tolerance=1e-5)
w = torch.rand(B, src.shape[1], device=dev, requires_grad=True)
for _ in range(reps):
    T = icp.icp(src, raw["map_pc"], T_init=raw["T_init"], weight=w, trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0}, dim=2)["T"]
    T.sum().backward()
torch.cuda.synchronize()
print("ok")
