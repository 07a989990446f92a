"""The dICP forward alone at the bench shape (B=32, N=5120, M=20480 padded, dim 2, pt2pl Huber, 10 iterations) on the
synthetic scenes, with the library's NN timing hook: microseconds per NN launch by ICP iteration (seeded iterations 1..9
vs the unseeded iteration 0).  Development tool; MMK_LIB=<path> times another build of the library."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mm_masking_amd import _lib, synthetic, train_icp_weights as trn
from mm_masking_amd.dICP.ICP import ICP
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
L = _lib.lib()
params = trn.default_params(dev)
raw = synthetic.make_batch(list(range(B)), device=dev)
batch = trn.prepare_batch(raw, params)
src = batch["loc_data"]["filtered_pc"]
icp = ICP("pt2pl", differentiable=True, max_iterations=10, tolerance=1e-5)
w = torch.rand(B, src.shape[1], device=dev)
def run():
    with torch.no_grad():
        return icp.icp(src, raw["map_pc"], T_init=raw["T_init"], weight=w, trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0}, dim=2)["T"]
T0 = run().clone()
torch.cuda.synchronize()
cap = reps * 10 + 8
_lib.check(L.mmk_nn_profile_begin(cap))
for _ in range(reps):
    T = run()
torch.cuda.synchronize()
ms = (ctypes.c_float * cap)(); n = ctypes.c_int32(0)
_lib.check(L.mmk_nn_profile_end(ms, cap, ctypes.byref(n)))
t = np.array(ms[:n.value]).reshape(reps, 10) * 1e3
print("lib %s: NN us by iteration %s  mean %.1f  T checksum %.6f" % (os.path.basename(os.environ.get("MMK_LIB", "default")),
      " ".join("%.0f" % v for v in np.median(t, axis=0)), np.median(t, axis=0).mean(), float(T.double().sum())), flush=True)
