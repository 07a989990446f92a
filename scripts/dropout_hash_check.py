"""Offline check of the dropout hash of csrc/mmk_unet_shared.h (dropout_draws4: 24-bit multiplies): drop rates, correlations
between the four draws of a group and between neighbouring groups / pixels / rows, uniformity of the 16-bit draws.  numpy only."""
import numpy as np
M32=np.uint64(0xffffffff)
def mad24(a,b,c): return ((a & np.uint64(0xffffff))*(np.uint64(b)&np.uint64(0xffffff)) + c) & M32
def mul24(a,b): return ((a & np.uint64(0xffffff))*(np.uint64(b)&np.uint64(0xffffff))) & M32
def h(idx, seed):
    x = mad24(idx, 0x9E3779, np.uint64(seed))
    x ^= x >> np.uint64(13)
    x = mul24(x, 0x85EBCB)
    x ^= x >> np.uint64(11)
    x = mul24(x, 0xC2B2AF)
    x ^= x >> np.uint64(15)
    return x
def h2(h1):
    y = h1 ^ np.uint64(0x85ebca6b)
    y ^= y >> np.uint64(12)
    y = mul24(y, 0x7FEB35)
    y ^= y >> np.uint64(14)
    return y
N=1<<22
idx=np.arange(N,dtype=np.uint64)
for seed in (3, 3*64+1, 12345):
    a=h(idx,seed); b=h2(a)
    d=np.stack([a&np.uint64(0xffff), a>>np.uint64(16), b&np.uint64(0xffff), b>>np.uint64(16)],1).astype(np.float64)
    thr=3277
    keep=(d>=thr)
    print("seed",seed,"drop rate per lane", 1-keep.mean(0), "expected", thr/65536)
    # correlations between draws in a group, and between neighbours
    c=np.corrcoef(keep.T.astype(float)); print(" corr within group max offdiag", np.abs(c-np.eye(4)).max())
    k0=keep[:,0].astype(float)
    for lag in (1,2,16,64,4096):
        print("  lag",lag, np.corrcoef(k0[:-lag],k0[lag:])[0,1], end="")
    print()
    # uniformity of 16-bit draws: chi-square over 256 bins
    for j in range(4):
        cnt=np.bincount((d[:,j].astype(np.int64)>>8),minlength=256); 
        chi=((cnt-N/256)**2/(N/256)).sum(); print("  chi2/255 lane",j, chi/255, end="")
    print()
    # 2-D structure: image rows: index = p*COUT + c0: check correlation across pixels for fixed channel group (stride 16 groups for 64 ch)
    k=keep[:,0].reshape(-1,16)[:,0].astype(float)
    print("  pixel-neighbour corr (stride 16 groups):", np.corrcoef(k[:-1],k[1:])[0,1], "row-neighbour(160):", np.corrcoef(k[:-160],k[160:])[0,1])
