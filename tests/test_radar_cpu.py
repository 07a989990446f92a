"""CPU-only checks of the radar host functions: the oracle's Cartesian -> polar restatement against the reference's golden
vectors (bit-exact) and the product's own ``load_radar`` against the golden decode."""
import os

import numpy as np
import pytest
import torch

from mm_masking_amd import radar_utils as ru
from oracle import radar_ref


def test_oracle_cart_to_polar_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "cart2polar.npz"), allow_pickle=False)
    out = radar_ref.radar_cartesian_to_polar(g["cart"], g["az"], 0.0596, polar_pixel_shape=g["polar"].shape[1:])
    assert out.dtype == np.float64 and np.array_equal(out, g["polar"])            # bit-exact
    out2 = radar_ref.radar_cartesian_to_polar(g["cart2"], g["az2"], 0.1, cart_resolution=0.3, polar_pixel_shape=(16, 120))
    assert np.array_equal(out2, g["polar2"])
    assert (g["polar"] == 0).mean() > 0.2 and (g["polar"] != 0).mean() > 0.2       # both inside and outside the image
    with pytest.raises(RuntimeError, match=str(g["fp32_error"])):
        radar_ref.radar_cartesian_to_polar(g["cart"].astype(np.float32), g["az"], 0.0596, polar_pixel_shape=(24, 200))
    # the product raises the reference's error for anything but fp64 before touching a device
    with pytest.raises(RuntimeError, match=str(g["fp32_error"])):
        ru.radar_cartesian_to_polar(torch.zeros(1, 8, 8), torch.zeros(1, 4), 0.0596, polar_pixel_shape=(4, 10))


def test_product_load_radar_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "radar_load.npz"), allow_pickle=False)
    fft, az, ts = ru.load_radar(g["png"])
    assert fft.dtype == np.float32 and np.array_equal(fft, g["fft"])
    assert np.array_equal(az, g["az"]) and np.array_equal(ts, g["ts"])
