"""Timing of the radar pre-processing kernels at the bench shape (B=32 polar scans): GO-CFAR, peak
extraction, polar -> Cartesian, channel min/max, mask normalisation.  Development tool (GPU box)."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "scripts")
import torch
from mm_masking_amd import radar_utils as ru, synthetic, unet_hip as uh
from mm_masking_amd import train_icp_weights as trn
from bench_layers import timeit, DEV

B = 32
raw = synthetic.make_batch(list(range(4)), device=DEV, m_valid=2000, m_pad=2048)
fft = raw["fft_polar"].repeat(B // 4, 1, 1).contiguous()
az = raw["azimuths"].repeat(B // 4, 1).contiguous()
tm = raw["az_times"].repeat(B // 4, 1).contiguous()
print("cfar %.1f us" % timeit(lambda: ru.cfar_mask(fft, 0.0596, a_thresh=1.0, b_thresh=0.09, diff=False)))
cf = ru.cfar_mask(fft, 0.0596, a_thresh=1.0, b_thresh=0.09, diff=False)
print("extract_pc_padded %.1f us" % timeit(lambda: ru.extract_pc_padded(cf, 0.0596, az, tm, 5120, diff=False)))
print("polar_to_cart %.1f us" % timeit(lambda: ru.radar_polar_to_cartesian_diff(fft, az, 0.0596)))
cart = ru.radar_polar_to_cartesian_diff(fft, az, 0.0596).unsqueeze(1).contiguous()
print("channel_minmax %.1f us" % timeit(lambda: uh.channel_minmax(cart)))
print("polar_to_cart pair %.1f us" % timeit(lambda: ru._polar_to_cart_pair(fft, cf, az, 0.0596)))
