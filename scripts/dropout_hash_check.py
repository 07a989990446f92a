"""Offline check of the dropout hash of csrc/mmk_unet_shared.h (dropout_draws4): drop rates, correlations between the four draws of
a group, between neighbouring groups / pixels / rows and between the masks of consecutive layer seeds, uniformity of the 16-bit
draws, and -- ADVICE r04 -- how many DISTINCT draw quadruples the real group counts produce (the B = 32, 640 x 640 x 8 layers have
2^24.6 groups).  numpy only.   python scripts/dropout_hash_check.py [variant ...]     variants: r04 (plain 24-bit multiplies), r05 (state as the addend), r05b (bijective steps: the shipped one)"""
import sys

import numpy as np

M32 = np.uint64(0xffffffff)
M24 = np.uint64(0xffffff)


def mul24(a, b):
    return ((a & M24) * np.uint64(b)) & M32


def draws(idx, seed, variant):
    """(h, g): the two 32-bit words whose halves are the four draws of group `idx`."""
    if variant == "r05b":
        # every step a bijection of the 32-bit state: lo24 * C + x with C EVEN is lo * (C + 1) + (hi << 24), (C + 1) odd
        x = (mul24(idx, 0x9E3779) + np.uint64(seed) + (idx & np.uint64(0xff000000))) & M32
        x ^= x >> np.uint64(13)
        x = (mul24(x, 0x85EBCA) + x) & M32
        x ^= x >> np.uint64(11)
        x = (mul24(x, 0xC2B2AE) + x) & M32
        x ^= x >> np.uint64(15)
        y = x ^ np.uint64(0x85ebca6b)
        y ^= y >> np.uint64(12)
        y = (mul24(y, 0x7FEB34) + y) & M32
        y ^= y >> np.uint64(14)
        return x, y
    x = (mul24(idx, 0x9E3779) + np.uint64(seed) + mul24(idx >> np.uint64(24), 0x9E3779)) & M32
    x ^= x >> np.uint64(13)
    if variant == "r04":
        x = mul24(x, 0x85EBCB)
        x ^= x >> np.uint64(11)
        x = mul24(x, 0xC2B2AF)
    else:       # r05: v_mad_u32_u24 with the running state as the addend -- the high byte a 24-bit multiply drops stays in the state
        x = (mul24(x, 0x85EBCB) + x) & M32
        x ^= x >> np.uint64(11)
        x = (mul24(x, 0xC2B2AF) + x) & M32
    x ^= x >> np.uint64(15)
    y = x ^ np.uint64(0x85ebca6b)
    y ^= y >> np.uint64(12)
    y = mul24(y, 0x7FEB35) if variant == "r04" else (mul24(y, 0x7FEB35) + y) & M32
    y ^= y >> np.uint64(14)
    return x, y


def report(variant, N, thr=3277):
    print("== variant %s, %d groups (2^%.1f), thr %d (p = %.5f)" % (variant, N, np.log2(N), thr, thr / 65536))
    idx = np.arange(N, dtype=np.uint64)
    masks = {}
    for seed in (3 * 64 + 1, 3 * 64 + 2, 12345 * 64 + 7):          # two consecutive layers of one pass, and a far one
        a, b = draws(idx, seed, variant)
        d = np.stack([a & np.uint64(0xffff), a >> np.uint64(16), b & np.uint64(0xffff), b >> np.uint64(16)], 1)
        keep = d >= thr
        masks[seed] = keep
        print(" seed %d: drop rate per draw %s" % (seed, np.round(1 - keep.mean(0), 5)))
        c = np.corrcoef(keep.T.astype(np.float32))
        print("   corr within a group (max off-diagonal) %.2e" % np.abs(c - np.eye(4)).max())
        k0 = keep[:, 0].astype(np.float32)
        print("   corr of draw 0 at group lags 1, 2, 16, 64, 4096, 2^24: " +
              " ".join("%.1e" % np.corrcoef(k0[:-lag], k0[lag:])[0, 1] for lag in (1, 2, 16, 64, 4096, 1 << 24) if lag < N))
        chi = []
        for j in range(4):
            cnt = np.bincount((d[:, j] >> np.uint64(8)).astype(np.int64), minlength=256)
            chi.append(((cnt - N / 256) ** 2 / (N / 256)).sum() / 255)
        print("   chi^2 / 255 of the draws' high bytes: " + " ".join("%.2f" % v for v in chi))
        key = (a << np.uint64(32)) | b
        uniq = np.unique(key).size
        print("   distinct draw quadruples: %d of %d (%.2f %% of the groups share theirs with another group)" % (
            uniq, N, 100.0 * (N - uniq) / N))
        del a, b, d, key
    s = sorted(masks)
    for u, v in ((s[0], s[1]), (s[0], s[2])):
        cc = [np.corrcoef(masks[u][:, j].astype(np.float32), masks[v][:, j].astype(np.float32))[0, 1] for j in range(4)]
        print(" masks of seeds %d and %d, same group: corr %s" % (u, v, " ".join("%.1e" % x for x in cc)))


if __name__ == "__main__":
    variants = sys.argv[1:] or ["r04", "r05", "r05b"]
    for v in variants:
        report(v, 1 << 25)
