// Types and helpers of the mask U-Net kernels (mmk_unet.hip and the experiment kept as scripts/experiments/r04_conv_dx_kernel.hip).
#pragma once
#include "mmk_common.h"

namespace mmku {

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
// native vector type: the HIP uint4 struct is copied by memcpy, which keeps a register ring in scratch
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct ConvOutPart {
    bf16 *y;               // (B,H,W,C)
    const bf16 *relu_src;  // optional (B,H,W,C): y = acc * (relu_src > 0 ? scale : 0)
    int C;
    int accumulate;        // y += result
    float scale;
};

struct ConvArgs {
    const bf16 *x1, *x2;   // input = concat(x1 (C1 channels), x2 (C2 channels)); x2 may be null
    int C1, C2;
    const bf16 *wpack;
    const float *bias;     // [COUT] or null
    ConvOutPart o1, o2;    // output channels [0,o1.C) -> o1, [o1.C, o1.C+o2.C) -> o2
    int B, H, W, CIN, COUT;
    int relu;
    float slope;           // > 0: nn.LeakyReLU(slope) instead of ReLU (forward), and the negative-side factor
                           // slope * scale of the relu_src epilogues (backward); 0 = plain ReLU
    float drop_p;          // forward dropout on the output (0 = none)
    unsigned seed;
    bf16 *pool_y;          // optional (B,H/2,W/2,COUT): 2x2 max-pool of the output, written by the same pass
    unsigned char *pool_arg = nullptr;   // with pool_y: (B,H/2,W/2,COUT/2) arg-max codes of the pooling windows, one nibble per
                                         // channel = position of the first maximum in scan order | (maximum > 0) << 2; the
                                         // full-resolution output o1.y is then NOT written (may be null)
    // sub-batch launches of dispatch_conv_deep (round 4): the element index of the dropout draws continues where the images in
    // front of this sub-batch end, and the packed weights may be laid out for a wider block than the kernel's own (wpack_mtb =
    // 16-channel tiles per group of the packing, 0 = the kernel's own width)
    unsigned hash_base = 0;
    int wpack_mtb = 0;
};

// Inverted-dropout scales (0 or 1/keep) of the 4 consecutive channels starting at element index e4 (a multiple of 4): four
// 16-bit draws compared against thr = round(p * 65536).  Round 4: the draws come from a hash built on FULL-RATE 24-bit
// multiplies (v_mad_u32_u24 / v_mul_u32_u24).  The avalanche hash of rounds 1-3 used 32-bit multiplies, which are quarter
// rate on the VALU: three of them per four values were half of the forward epilogues of the issue-bound kernels (switching
// dropout off moved the thin 640 x 640 forward launches by 13-30 us each, HISTORY.md 9.10).  Nothing downstream depends on
// WHICH elements are dropped -- the backward kernels read the mask back from the stored activation -- only on the rate and
// on independence (scripts/dropout_hash_check.py; the chain itself: below).
struct DropoutParams {
    unsigned thr;      // drop probability thr / 65536: an element is kept when its SIGNED 16-bit draw is >= thr - 32768
    float inv_keep;    // 1 / (1 - thr / 65536): exactly unbiased for the quantised probability
    int thr_s;         // thr - 32768
};

__host__ __device__ inline DropoutParams dropout_params(float p)
{
    DropoutParams d;
    d.thr = (unsigned)(p * 65536.0f + 0.5f);
    d.inv_keep = d.thr ? 65536.0f / (float)(65536u - d.thr) : 1.0f;
    d.thr_s = (int)d.thr - 32768;
    return d;
}

// Round 5: every step is a BIJECTION of the 32-bit state.  A 24-bit multiply alone keeps the low 24 bits of its
// input, so the round-4 chain produced at most 2^24 different draw quadruples whatever the seed and the index: at the group
// counts of the 640 x 640 layers (2^24.6) 78 % of the groups shared their draws with another group, and group 2^24 + k repeated
// group k + 1.  v_mad_u32_u24 with the state itself as the addend is lo24 * C + x = lo24 * (C + 1) + (hi8 << 24), one-to-one for
// EVEN C -- same instruction count, same full-rate pipe.  scripts/dropout_hash_check.py on 2^25 groups: all quadruples distinct,
// byte histograms at chi^2 / dof = 1.0, |correlation| <= 5e-4 within a group, between neighbours and between the masks of
// consecutive layer seeds.
// the two hash words of the group of 4 consecutive channels starting at element index e4: their four 16-bit halves are the draws
__device__ __forceinline__ void dropout_words(unsigned seed, unsigned e4, unsigned &h, unsigned &g)
{
    const unsigned i = e4 >> 2;                       // group number
    h = __umul24(i, 0x9E3779u) + (seed + (i & 0xff000000u));
    h ^= h >> 13;
    h = __umul24(h, 0x85EBCAu) + h;
    h ^= h >> 11;
    h = __umul24(h, 0xC2B2AEu) + h;
    h ^= h >> 15;
    g = h ^ 0x85ebca6bU;
    g ^= g >> 12;
    g = __umul24(g, 0x7FEB34u) + g;
    g ^= g >> 14;
}

// ... as signed 16-bit integers (element r of the group: low / high half of h, low / high half of g).  Signed, so that a kernel
// can form the keep masks of two values at once on the packed halves (saturating v_pk_sub_i16 + arithmetic shift); every
// kernel of the library uses the same rule: an element is kept when draw >= thr - 32768.
__device__ __forceinline__ void dropout_draws4(unsigned seed, unsigned e4, int (&d)[4])
{
    unsigned h, g;
    dropout_words(seed, e4, h, g);
    d[0] = (int)(short)(h & 0xffffu); d[1] = (int)h >> 16; d[2] = (int)(short)(g & 0xffffu); d[3] = (int)g >> 16;
}

__device__ __forceinline__ void dropout_scale4(unsigned seed, unsigned e4, const DropoutParams &d, float (&sc)[4])
{
    int dr[4];
    dropout_draws4(seed, e4, dr);
#pragma unroll
    for (int r = 0; r < 4; ++r) sc[r] = (dr[r] >= d.thr_s) ? d.inv_keep : 0.f;
}

// the same draws as keep flags (the caller has folded the factor 1 / keep into the value already)
__device__ __forceinline__ void dropout_keep4(unsigned seed, unsigned e4, const DropoutParams &d, bool (&keep)[4])
{
    int dr[4];
    dropout_draws4(seed, e4, dr);
#pragma unroll
    for (int r = 0; r < 4; ++r) keep[r] = dr[r] >= d.thr_s;
}

}  // namespace mmku
