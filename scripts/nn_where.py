"""Where do the correspondences of the first (unseeded) ICP iteration differ from a reference run?  Writes / reads the first
iteration's indices of one pair set:  python scripts/nn_where.py save ref.npy ;  python scripts/nn_where.py cmp ref.npy"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mm_masking_amd import synthetic, train_icp_weights as trn
from mm_masking_amd.dICP.ICP import ICP
mode, path = sys.argv[1], sys.argv[2]
B = 32
dev = torch.device("cuda:0")
params = trn.default_params(dev)
raw = synthetic.make_batch(list(range(4000, 4000 + B)), device=dev)
batch = trn.prepare_batch(raw, params, max_loc_pts=5120)
src = batch["loc_data"]["filtered_pc"]
torch.manual_seed(5)
w0 = torch.rand(B, src.shape[1], device=dev)
icp = ICP("pt2pl", differentiable=True, max_iterations=2, tolerance=1e-5)
runs = []
for rep in range(6):
    w = w0.clone().requires_grad_(True)
    T = icp.icp(src, raw["map_pc"], T_init=raw["T_init"], weight=w, trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0}, dim=2)["T"]
    runs.append(T.grad_fn.saved_tensors[3].cpu().numpy()[0].copy())
    x = torch.randn(2048, 2048, device=dev); (x @ x).sum().item()
if mode == "save":
    np.save(path, runs[0])
    print("saved", runs[0].shape, "self-consistent", all(np.array_equal(r, runs[0]) for r in runs))
    sys.exit(0)
ref = np.load(path)
T0 = raw["T_init"].cpu().numpy().astype(np.float64)
S = src.cpu().numpy().astype(np.float64)
Mp = raw["map_pc"].cpu().numpy().astype(np.float64)
valid = (S != 0).any(-1).sum(1)
def dist(b, i, j):
    pt = T0[b, :3, :3] @ S[b, i, :3] + T0[b, :3, 3]
    d = pt[:2] - Mp[b, j, :2]
    return float(d @ d)
M = raw["map_pc"].shape[1]
print("N", src.shape[1], "M", M)
for rep, r in enumerate(runs):
    bad = np.argwhere(r != ref)
    print("rep", rep, "mismatches", len(bad))
    for b, i in bad[:40]:
        jr, jm = int(ref[b, i]), int(r[b, i])
        sb, q = divmod(int(i), 512)
        wv, q2 = divmod(q, 128)
        g, col = divmod(q2, 32)
        print("   b %2d (valid %d) i %4d (sb %d wave %d group %d col %2d)  ref j %5d (tile %2d) d %.4f  got j %5d (tile %2d) d %.4f | ref j of i-64: %d, its d to got j %.4f; of i-32: %d" % (
            b, valid[b], i, sb, wv, g, col, jr, jr // 1024, dist(b, i, jr), jm, jm // 1024, dist(b, i, jm), int(ref[b, i - 64]), dist(b, i - 64, jm), int(ref[b, i - 32])))
