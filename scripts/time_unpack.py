"""Times the reduction of the decoder's weight-gradient slices (mmk_conv3x3_wgrad_unpack_batch) at the bench shape."""
import sys, torch
sys.path.insert(0, ".")
from mm_masking_amd import unet_hip as uh
DEV = torch.device("cuda:0")
B = 32
# (cout, cin, H) of decoder convs k = 12..21
layers = [(128, 256, 40), (128, 128, 40), (64, 128, 80), (64, 64, 80), (32, 64, 160), (32, 32, 160), (16, 32, 320), (16, 16, 320), (8, 16, 640), (8, 8, 640)]
items = []
tot = 0
for co, ci, H in layers:
    ns = uh.wgrad_slices(co, ci, ci, B, H, H) * 2
    t = torch.randn(ns, 9 * co * ci + co, device=DEV)
    tot += t.numel() * 4
    items.append((t, co, ci, torch.empty(co, device=DEV)))
def run(): uh.wgrad_unpack_batch(items)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 20 * 1e3
print("unpack of %d layers, %.0f MB of slices: %.1f us = %.2f TB/s" % (len(items), tot / 1e6, t, tot / t * 1e-6))
