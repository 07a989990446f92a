// Mask U-Net building blocks on gfx950: NHWC bf16 activations, fp32 accumulation on the
// matrix cores (v_mfma_f32_16x16x32_bf16), fused bias / ReLU / dropout / ReLU-backward
// epilogues.  Replaces the nn.Conv2d / MaxPool2d / UpsamplingBilinear2d / Sigmoid stack of
// the reference's mask predictor (mm_masking/icp_weight_policy.py:84-99,104-125,161-184)
// for its default configuration (ReLU, no batch norm).
//
// 3x3 convolution as an implicit GEMM, weights as the MFMA A operand (rows = output
// channels) and pixels as the B operand (columns = 16 consecutive pixels of one image
// row), so that every lane ends up with 4 consecutive output channels of one pixel
// (8-byte NHWC stores) and every B fragment is one 16-byte LDS read of 8 consecutive
// input channels of one pixel at one tap.  The same kernel computes the data gradient
// (weights packed transposed + flipped).  The weight gradient contracts over pixels:
// both operands are read from row-major [pixel][channel] LDS tiles with the transposing
// ds_read_b64_tr_b16.
#include <math.h>
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "mmk_common.h"
#include "mmk_unet_shared.h"

namespace {

using namespace mmku;       // shared with mmk_conv_dx.hip: element types, ConvArgs, the dropout hash (mmk_unet_shared.h)

// Packed bf16 / int16 helpers of the forward epilogues.  (hipcc has no builtin for v_cvt_pk_bf16_f32 and turns the vector forms of
// the other two into per-half compares and selects; one instruction per asm statement, so that hipcc pads and schedules around
// each of them itself.)
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi)
{
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ unsigned pk_relu_bf16(unsigned x)          // max(x, 0) on both halves' bit patterns (-0 -> +0)
{
    unsigned r;
    asm("v_pk_max_i16 %0, %1, 0" : "=v"(r) : "v"(x));
    return r;
}
__device__ __forceinline__ unsigned pk_max_i16(unsigned x, unsigned y)     // y = 0: ReLU; y = 0x80008000: identity
{
    unsigned r;
    asm("v_pk_max_i16 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
// 0xffff in the halves where the signed 16-bit draw of `words` is >= thr - 32768 (thr1pk = both halves thr - 32768 - 1)
__device__ __forceinline__ unsigned pk_keep_mask(unsigned words, unsigned thr1pk)
{
    unsigned d, m;
    asm("v_pk_sub_i16 %0, %1, %2 clamp" : "=v"(d) : "v"(thr1pk), "v"(words));
    asm("v_pk_ashrrev_i16 %0, %1, %2" : "=v"(m) : "v"(0x000f000fu), "v"(d));
    return m;
}


constexpr int TH = 8, TW = 32;            // output pixels per block tile
constexpr int HT = TH + 2, WT = TW + 2;   // halo tile
constexpr int CONV_THREADS = 256;

// LDS pixel pitch (elements): 16 bytes of padding de-correlates the banks of consecutive
// pixels for the 16-byte fragment reads (a 128-byte pitch is an 8-way conflict).
// LDS pixel pitch (elements) of the conv kernels' halo tiles.  A B fragment is a ds_read_b128 whose 64
// lanes are served in four groups of 16 ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...: MI355X_MICROARCH.md,
// LDS): lane l reads pixel l%16, 16-byte granule l/16.  Enumerating those groups, the read is
// conflict-free exactly when the pitch is 2 (mod 4) granules of 16 bytes: 16 elements unpadded, 32 -> 48,
// 64 -> 80; the pitches c + 8 cost 2x there (SQ_LDS_BANK_CONFLICT = half of SQ_LDS_IDX_ACTIVE).  The
// 16-channel tiles are therefore unpadded and the >= 64-channel kernel uses 48; the 4-wave kernels keep
// 40 for 32 channels (48 would cost them a resident block per CU, and they are not LDS-bound).
// 8-channel tiles (1 granule per pixel) are conflict-free as they are.
__host__ __device__ constexpr int lds_pitch(int c) { return c >= 32 ? c + 8 : c; }
// weight-gradient tiles (read with ds_read_b64_tr_b16: different lane grouping, see there)
// A 16-lane group of a transposing read fetches 4 consecutive pixels x 32 bytes (16 channels), and the LDS
// serves it in two groups of 32 lanes.  With the k index of the MFMA mapped to pixels so that a group of 32
// lanes covers 8 *consecutive* pixels (lane group g, first/second read: pixels 4g+q / 16+4g+q), the read is
// conflict-free when the pixel pitch is 16 (mod 32) elements: 16 unpadded, 32 -> 48, 64 -> 80 (the old
// mapping, pixels 8g+q, conflicted 2-way at every pitch but 8).
__host__ __device__ constexpr int wg_pitch(int c) { return c >= 32 ? c + 16 : c; }
__host__ __device__ constexpr int ksteps(int ck) { return ck >= 32 ? 9 * (ck / 32) : (ck == 16 ? 5 : 3); }
// weight-gradient kernel: input-channel chunk / output-channel group of one block
__host__ __device__ constexpr int cin_chunk(int cin) { return cin >= 64 ? 64 : cin; }
__host__ __device__ constexpr int cout_group(int cout) { return cout >= 64 ? 64 : (cout <= 16 ? 16 : cout); }
// forward / data-gradient kernels: input channels are consumed in chunks of conv_ck, one block
// produces conv_cm output channels.  Layers with >= 64 channels on either side (and 32 -> >= 64)
// run on conv3x3_deep_kernel (8 waves, up to 128 output channels per block).
__host__ __device__ constexpr int conv_ck(int cin) { return cin >= 32 ? 32 : cin; }
__host__ __device__ constexpr bool conv_is_deep(int cin, int cout) { return cin >= 64 || (cin == 32 && cout >= 64); }
__host__ __device__ constexpr int conv_cm(int cin, int cout)
{
    if (conv_is_deep(cin, cout)) return cout >= 128 ? 128 : (cout <= 16 ? 16 : cout);
    return cout >= 64 ? 64 : (cout <= 16 ? 16 : cout);
}

// (tap, channel offset inside the chunk) of the 8 consecutive k values lane `l` holds in k-step `s`.
template <int CK>
__device__ __forceinline__ void kslot(int s, int l, int &tap, int &ch)
{
    if (CK >= 32) {
        tap = s / (CK / 32);
        ch = (s % (CK / 32)) * 32 + 8 * (l >> 4);
    } else if (CK == 16) {
        tap = 2 * s + (l >> 5);
        ch = 8 * ((l >> 4) & 1);
    } else {
        tap = 4 * s + (l >> 4);
        ch = 0;
    }
}

// ------------------------------------------------------------------------------------------
// Weight packing: fp32 master weights W[COUT][CIN][3][3] -> bf16 in MFMA A-fragment order
// [group][chunk][kstep][mtile][lane][8].  transposed = 1 packs the data-gradient operator
// (out channels = CIN, in channels = COUT, taps flipped).
__device__ __forceinline__ void pack_conv_weight_elem(const float *__restrict__ W, int COUT, int CIN, int transposed,
                                                      bf16 *__restrict__ out, int e)
{
    const int co_n = transposed ? CIN : COUT;   // channels produced by the packed operator
    const int ci_n = transposed ? COUT : CIN;   // channels consumed
    const int CK = conv_ck(ci_n), CM = conv_cm(ci_n, co_n);
    const int NS = ksteps(CK), MT = CM / 16;
    const int nchunk = ci_n / CK;
    int r = e;
    const int j = r & 7; r >>= 3;
    const int lane = r & 63; r >>= 6;
    const int mt = r % MT; r /= MT;
    const int s = r % NS; r /= NS;
    const int chunk = r % nchunk;
    const int group = r / nchunk;
    int tap, ch;
    if (CK >= 32) {
        tap = s / (CK / 32);
        ch = (s % (CK / 32)) * 32 + 8 * (lane >> 4);
    } else if (CK == 16) {
        tap = 2 * s + (lane >> 5);
        ch = 8 * ((lane >> 4) & 1);
    } else {
        tap = 4 * s + (lane >> 4);
        ch = 0;
    }
    // Row i of 16-channel tile mt = output channel co.  Thin layers: the 16 rows of a tile are 16 consecutive channels.  The
    // >= 64-channel kernel (conv3x3_deep_kernel, round 4): a lane group g (16 lanes) ends up with rows 4g .. 4g + 3 of EVERY tile
    // of its wave, so the rows are dealt such that those 4 x MTW values are 4 MTW CONSECUTIVE channels -- channel =
    // (4 MTW) (i / 4) + 4 (tile within the wave) + i % 4 -- and the lane stores (and reads its epilogue operands as) one
    // contiguous run per pixel, with no lane exchange in the epilogue.
    int co = group * CM + mt * 16 + (lane & 15);
    if (conv_is_deep(ci_n, co_n)) {
        const int MTW = MT < 4 ? MT : 4;               // tiles per wave (DeepCfg::MT)
        const int i = lane & 15;
        co = group * CM + (mt / MTW) * (16 * MTW) + (4 * MTW) * (i >> 2) + 4 * (mt % MTW) + (i & 3);
    }
    const int ci = chunk * CK + ch + j;
    float v = 0.f;
    if (tap < 9 && co < co_n && ci < ci_n) {
        if (transposed) v = W[((size_t)ci * CIN + co) * 9 + (8 - tap)];
        else v = W[((size_t)co * CIN + ci) * 9 + tap];
    }
    out[e] = (bf16)v;
}

__global__ void pack_conv_weights_kernel(const float *__restrict__ W, int COUT, int CIN, int transposed,
                                         bf16 *__restrict__ out, int total)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    pack_conv_weight_elem(W, COUT, CIN, transposed, out, e);
}

// All layers of a network in one launch (blockIdx.y = layer): 42 five-microsecond launches per
// training step otherwise.
constexpr int PACK_BATCH_MAX = 32;
struct PackBatch {
    const float *W[PACK_BATCH_MAX];
    bf16 *out[PACK_BATCH_MAX];
    int cout[PACK_BATCH_MAX], cin[PACK_BATCH_MAX], total[PACK_BATCH_MAX];
    int transposed;
};

__global__ void pack_conv_weights_batch_kernel(const PackBatch pb)
{
    const int l = blockIdx.y;
    const int total = pb.total[l];
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x)
        pack_conv_weight_elem(pb.W[l], pb.cout[l], pb.cin[l], pb.transposed, pb.out[l], e);
}

// LeakyReLU variant of the network (params["leaky"], icp_weight_policy.py:106: nn.LeakyReLU(0.1)).
// Forward: max(v, slope v).  The backward factor is recovered from the stored (post-dropout) activation d,
// as for ReLU:  d > 0 -> scale,  d < 0 -> slope * scale,  dropped -> 0.  A pre-activation of exactly zero
// that was kept must still count as "negative side" (torch: x > 0 ? g : slope g) and must not look like a
// dropped element, so a kept zero is stored as -0.0 and a dropped element as +0.0 (the dropout is a select,
// not a product: -x * 0 would be -0.0).  Nothing downstream distinguishes the two zeros except the factor.
__device__ __forceinline__ float act_leaky(float v, float slope)
{
    const float t = fmaxf(v, v * slope);
    return (t == 0.f) ? -0.f : t;
}
__device__ __forceinline__ float drop_leaky(float v, float sc) { return (sc != 0.f) ? v * sc : 0.f; }
__device__ __forceinline__ float bwd_factor_leaky(float src, float scale, float slope)
{
    return (src > 0.f) ? scale : ((__float_as_uint(src) != 0u) ? scale * slope : 0.f);
}

// Software-pipelined, persistent convolution: a block walks the stages (tile, input-channel
// chunk) of its share of the tiles; while the matrix cores work on the stage that sits in
// LDS, the next stage's halo tile (and weight block, when it changes) is already in flight
// from HBM/L2 into registers and is written to LDS after the barrier.
template <int CK, int CM, bool LK = false>
__global__ __launch_bounds__(CONV_THREADS) void conv3x3_kernel(const ConvArgs a, int total_tiles)
{
    constexpr int NS = ksteps(CK);
    constexpr int MT = CM / 16;
    constexpr int NT = 4;
    constexpr int PK = lds_pitch(CK);
    constexpr int GPP = CK / 8;
    constexpr int NIN = HT * WT * GPP;           // 16-byte granules of the halo tile
    constexpr int NW = NS * MT * 64;             // 16-byte granules of a weight block
    constexpr int RIN = (NIN + CONV_THREADS - 1) / CONV_THREADS;
    constexpr int RW = (NW + CONV_THREADS - 1) / CONV_THREADS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16 *in_tile = reinterpret_cast<bf16 *>(smem);
    bf16 *w_lds = in_tile + HT * WT * PK;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
    const int group = blockIdx.y;
    const int nchunk = a.CIN / CK;
    int tile = blockIdx.x;
    if (tile >= total_tiles) return;

    uint4 rin[RIN];
    uint4 rw[RW];

    auto load_in = [&](int t, int chunk) {
        const int b = t / (tiles_x * tiles_y), tr = t % (tiles_x * tiles_y);
        const int tx0 = (tr % tiles_x) * TW, ty0 = (tr / tiles_x) * TH;
#pragma unroll
        for (int i = 0; i < RIN; ++i) {
            const int g = tid + i * CONV_THREADS;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (g < NIN) {
                const int pix = g / GPP, gc = g % GPP;
                const int yy = ty0 + pix / WT - 1, xx = tx0 + pix % WT - 1;
                const int c = chunk * CK + gc * 8;
                if (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) {
                    const size_t p = ((size_t)b * a.H + yy) * a.W + xx;
                    if (c < a.C1) v = *reinterpret_cast<const uint4 *>(a.x1 + p * a.C1 + c);
                    else v = *reinterpret_cast<const uint4 *>(a.x2 + p * a.C2 + (c - a.C1));
                }
            }
            rin[i] = v;
        }
    };
    auto load_w = [&](int chunk) {
        const uint4 *wsrc = reinterpret_cast<const uint4 *>(a.wpack + ((size_t)(group * nchunk + chunk)) * NS * MT * 512);
#pragma unroll
        for (int i = 0; i < RW; ++i) {
            const int g = tid + i * CONV_THREADS;
            rw[i] = (g < NW) ? wsrc[g] : make_uint4(0, 0, 0, 0);
        }
    };
    auto store_in = [&]() {
#pragma unroll
        for (int i = 0; i < RIN; ++i) {
            const int g = tid + i * CONV_THREADS;
            if (g < NIN) *reinterpret_cast<uint4 *>(in_tile + (size_t)(g / GPP) * PK + (g % GPP) * 8) = rin[i];
        }
    };
    auto store_w = [&]() {
#pragma unroll
        for (int i = 0; i < RW; ++i) {
            const int g = tid + i * CONV_THREADS;
            if (g < NW) reinterpret_cast<uint4 *>(w_lds)[g] = rw[i];
        }
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const DropoutParams dp = dropout_params(a.drop_p);
    constexpr bool lk = LK;                 // LeakyReLU variant of the network (a.slope > 0)
    int chunk = 0;
    load_in(tile, 0);
    load_w(0);
    bool first = true;
    while (true) {
        store_in();
        if (first || nchunk > 1) store_w();
        first = false;
        __syncthreads();
        int ntile = tile, nck = chunk + 1;
        if (nck == nchunk) {
            nck = 0;
            ntile = tile + gridDim.x;
        }
        const bool has_next = ntile < total_tiles;
        if (has_next) {
            load_in(ntile, nck);
            if (nchunk > 1) load_w(nck);
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            int tap, ch;
            kslot<CK>(s, lane, tap, ch);
            tap = tap > 8 ? 8 : tap;
            const int ty = tap / 3, tx = tap % 3;
            bf16x8 bf[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int row = 2 * wv + (n >> 1), col = (n & 1) * 16 + (lane & 15);
                bf[n] = *reinterpret_cast<const bf16x8 *>(in_tile + ((size_t)((row + ty) * WT + col + tx)) * PK + ch);
            }
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const bf16x8 af = *reinterpret_cast<const bf16x8 *>(w_lds + ((size_t)((s * MT + m) * 64 + lane)) * 8);
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf[n], acc[m][n], 0, 0, 0);
            }
        }
        if (chunk == nchunk - 1) {
            const int b = tile / (tiles_x * tiles_y), tr = tile % (tiles_x * tiles_y);
            const int tx0 = (tr % tiles_x) * TW, ty0 = (tr / tiles_x) * TH;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int c0 = group * CM + m * 16 + (lane >> 4) * 4;
                if (c0 < a.COUT) {
                    const bool firstp = c0 < a.o1.C;
                    bf16 *o_y = firstp ? a.o1.y : a.o2.y;
                    const bf16 *o_src = firstp ? a.o1.relu_src : a.o2.relu_src;
                    const int o_C = firstp ? a.o1.C : a.o2.C;
                    const bool o_acc = (firstp ? a.o1.accumulate : a.o2.accumulate) != 0;
                    const float o_scale = firstp ? a.o1.scale : a.o2.scale;
                    const int cl = firstp ? c0 : c0 - a.o1.C;
                    float bs[4] = {0.f, 0.f, 0.f, 0.f};
                    if (a.bias) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) bs[r] = a.bias[c0 + r];
                    }
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        const int yy = ty0 + 2 * wv + (n >> 1), xx = tx0 + (n & 1) * 16 + (lane & 15);
                        if (yy < a.H && xx < a.W) {
                            const size_t p = ((size_t)b * a.H + yy) * a.W + xx;
                            float v[4];
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                v[r] = acc[m][n][r] + bs[r];
                                if (a.relu) v[r] = lk ? act_leaky(v[r], a.slope) : fmaxf(v[r], 0.f);
                            }
                            if (a.drop_p > 0.f) {
                                float sc[4];
                                dropout_scale4(a.seed, (unsigned)(p * a.COUT + c0), dp, sc);
#pragma unroll
                                for (int r = 0; r < 4; ++r) v[r] = lk ? drop_leaky(v[r], sc[r]) : v[r] * sc[r];
                            }
                            bf16 *dst = o_y + p * o_C + cl;
                            if (o_src) {
                                const bf16x4 sv = *reinterpret_cast<const bf16x4 *>(o_src + p * o_C + cl);
#pragma unroll
                                for (int r = 0; r < 4; ++r)
                                    v[r] = lk ? v[r] * bwd_factor_leaky((float)sv[r], o_scale, a.slope)
                                              : (((float)sv[r] > 0.f) ? v[r] * o_scale : 0.f);
                            }
                            if (o_acc) {
                                const bf16x4 ov = *reinterpret_cast<const bf16x4 *>(dst);
#pragma unroll
                                for (int r = 0; r < 4; ++r) v[r] += (float)ov[r];
                            }
                            bf16x4 outv;
#pragma unroll
                            for (int r = 0; r < 4; ++r) outv[r] = (bf16)v[r];
                            *reinterpret_cast<bf16x4 *>(dst) = outv;
                        }
                    }
                }
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
        if (!has_next) break;
        __syncthreads();
        tile = ntile;
        chunk = nck;
    }
}

template <int CK, int CM, bool LK = false>
int launch_conv(const ConvArgs &a, hipStream_t st)
{
    const size_t smem = ((size_t)HT * WT * lds_pitch(CK) + (size_t)ksteps(CK) * (CM / 16) * 512) * sizeof(bf16);
    if (smem > 64 * 1024) {
        static bool attr_set[64] = {};   // per template instantiation and device
        int dev = 0;
        MMK_CHECK_HIP(hipGetDevice(&dev));
        if (!attr_set[dev & 63]) {
            MMK_CHECK_HIP(hipFuncSetAttribute((const void *)conv3x3_kernel<CK, CM, LK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            attr_set[dev & 63] = true;
        }
    }
    const int tiles = ((a.W + TW - 1) / TW) * ((a.H + TH - 1) / TH);
    const int groups = (a.COUT + CM - 1) / CM;
    // persistent grid: as many blocks as the LDS footprint lets a CU hold, on 256 CUs
    int per_cu = (int)std::min<size_t>(8, (160 * 1024) / smem);
    per_cu = per_cu < 1 ? 1 : per_cu;
    const int total = tiles * a.B;
    int gx = (256 * per_cu) / groups;
    gx = gx < 1 ? 1 : (gx > total ? total : gx);
    hipLaunchKernelGGL((conv3x3_kernel<CK, CM, LK>), dim3(gx, groups), dim3(CONV_THREADS), smem, st, a, total);
    MMK_LAUNCH_CHECK();
    return MMK_OK;
}


// ------------------------------------------------------------------------------------------
// Streaming variant for the layers whose input channels fit one chunk (CIN = CK <= 32): the
// 640x640 / 320x320 levels, which are bound by HBM latency x bytes in flight, not by the
// matrix cores.  Differences from conv3x3_kernel:
//   * a ring of RD register slots keeps RD halo tiles in flight per block (one slot is
//     written to LDS per stage and re-armed at once), LDS is double buffered -> one barrier
//     per stage, and that barrier only waits for LDS (lgkmcnt), never for the loads in flight;
//   * every global load is unconditional (out-of-image granules read a 16-byte zero word), so
//     the compiler's vmcnt bookkeeping stays exact and a wait for tile k+1 does not drain the
//     loads of tiles k+2..k+RD;
//   * the epilogue's operands (ReLU-backward source, accumulate target) are fetched at the top
//     of the stage, behind nothing, and are consumed after the MFMA loop (EPI = true);
//   * each XCD walks its own contiguous range of tiles, so halo rows shared by neighbouring
//     tiles hit that XCD's L2.
// native vector type: the HIP uint4 struct is copied by memcpy, which keeps a register ring in scratch
__device__ u32x4 g_zero16;   // zero-initialised, never written
__device__ u32x4 g_zero32[2];   // (32 bytes of them)
__device__ u32x4 g_sink32[2];   // where the unconditional stores of lanes without an output go
}  // namespace
// write-only: where lanes without an output element store.  External linkage on purpose: stores to an INTERNAL variable that
// nothing reads are deleted by the compiler, and the ring kernel's prologue issues stores to it only to give the first round
// the load / store queue of every later one (round 5: they had been deleted, and hipcc's merged loop-head state then made every
// stage wait for all but the last one or two of the loads in flight -- the ring was one tile deep instead of RD)
__device__ mmku::u32x4 mmk_sink16[8];
namespace {
#define g_sink16 mmk_sink16

__device__ __forceinline__ void unpack8(const u32x4 v, float (&f)[8])
{
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f[2 * j] = __uint_as_float(v[j] << 16);
        f[2 * j + 1] = __uint_as_float(v[j] & 0xffff0000u);
    }
}

__device__ __forceinline__ u32x4 pack8(const float (&f)[8])
{
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16)f[j];
    return __builtin_bit_cast(u32x4, o);
}

__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Diagnostic build only (-DMMK_DEEP_STAMPS; scripts/deep_stamps.py): the first 64 blocks record s_memtime at eight points of each
// of their first MMK_STAMP_STAGES stages, per wave, in LDS, and copy them to a caller's buffer when they exit (conv3x3_ring_kernel:
// the same per tile, plus every block's start and end on the 100 MHz clock: scripts/ring_stamps.py).  The shipped library
// contains none of this.
#ifdef MMK_DEEP_STAMPS
constexpr int MMK_STAMP_STAGES = 24;
__device__ unsigned long long *g_deep_stamp_buf = nullptr;
#define MMK_STAMP(K)                                                                                          \
    do {                                                                                                      \
        if (stamp_stage < MMK_STAMP_STAGES) {                                                                 \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                       \
            if (lane == 0) stamp_lds[(wv * MMK_STAMP_STAGES + stamp_stage) * 8 + (K)] = t_;                    \
        }                                                                                                     \
    } while (0)
#else
#define MMK_STAMP(K) do { } while (0)
#endif

#ifndef MMK_RING_DIAG
#define MMK_RING_DIAG 0      // diagnostic builds (scripts/build_variant.sh): 1 = no output stores, 2 = no input loads
#endif
// POOL: 0 = no pooling; 1 = the output and its 2x2 max-pool (pool_y); 2 = the max-pool and its arg-max codes (pool_y, pool_arg)
// only: the full-resolution output is not written at all (the backward pass routes the pooled gradient by the codes,
// maxpool2_bwd_arg_kernel, and nothing else reads the block's pre-pool output)
template <int CK, int CM, int RD, bool EPI, int POOL = 0, bool C8 = false>
__global__ __launch_bounds__(CONV_THREADS) void conv3x3_ring_kernel(const ConvArgs a, int total_tiles, int tiles_per_xcd)
{
    static_assert(!(EPI && POOL), "the pooled output belongs to the forward pass (no epilogue operands)");
    static_assert(!C8 || (CM == 16 && !POOL), "C8: a single output part of exactly 8 channels");
    constexpr int NS = ksteps(CK);
    constexpr int MT = CM / 16;
    constexpr int NT = 4;
    constexpr int PK = lds_pitch(CK);
    constexpr int GPP = CK / 8;
    constexpr int NIN = HT * WT * GPP;
    constexpr int NW = NS * MT * 64;
    constexpr int RIN = (NIN + CONV_THREADS - 1) / CONV_THREADS;
    constexpr int RW = (NW + CONV_THREADS - 1) / CONV_THREADS;
    constexpr int NST = C8 ? 1 : (POOL == 2 ? 2 * MT : MT * NT + (POOL ? MT : 0));       // stores per stage
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16 *in_tile = reinterpret_cast<bf16 *>(smem);           // 2 buffers of HT*WT*PK
    bf16 *w_lds = in_tile + 2 * HT * WT * PK;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
    const int tpi = tiles_x * tiles_y;
    const int group = blockIdx.y;

    const int xcd = blockIdx.x & 7, nb = gridDim.x >> 3;
    const int t_begin = xcd * tiles_per_xcd;
    const int t_end = (t_begin + tiles_per_xcd < total_tiles) ? t_begin + tiles_per_xcd : total_tiles;
    const int first = t_begin + (blockIdx.x >> 3);
    if (first >= t_end) return;
    const int nt_blk = (t_end - first + nb - 1) / nb;         // tiles of this block: first + k * nb
#ifdef MMK_DEEP_STAMPS
    unsigned long long *stamp_lds = reinterpret_cast<unsigned long long *>(smem + ((size_t)(2 * HT * WT * PK + NS * MT * 512) * sizeof(bf16) + 15) / 16 * 16);
    int stamp_stage = 0;
    for (int i = tid; i < 4 * MMK_STAMP_STAGES * 8; i += CONV_THREADS) stamp_lds[i] = 0ull;
    __syncthreads();
    if (tid == 0) {     // (slots 6, 7 of a stage are free)
        stamp_lds[6] = gridDim.x; stamp_lds[7] = (unsigned long long)nt_blk;
        stamp_lds[8 + 6] = __builtin_amdgcn_s_memtime(); stamp_lds[8 + 7] = __builtin_amdgcn_s_memrealtime();
    }
#endif

    // (Tried, round 5, no gain: a rotating raised wave priority.  The CU's issue arbiter prefers the OLDEST wave, so of the blocks
    // that share a CU the first one dispatched finishes its tiles in 52 us and the last in 96 us of a 99 us launch
    // (scripts/ring_stamps.py).  Handing s_setprio 2 to another wave slot every 4 096 cycles narrows that to 59 .. 97 us and moves
    // the launch by nothing, and so does giving blocks 4-24 consecutive tiles each and leaving the balance to the hardware
    // dispatcher (slower): two or more resident blocks already keep the CU as busy as five do, profiles/r05_ring_stamps.txt.)
    u32x4 rin[RD][RIN];
    // Everything per-lane that does not depend on the tile is worked out once: addresses inside the
    // loop are then "uniform tile origin (scalar unit) + 32-bit lane offset", a handful of VALU
    // instructions per access instead of a 64-bit index computation (the full-resolution layers
    // are bound by instruction issue, not by the matrix cores).  Offsets are in elements and fit
    // 32 bits (the launcher checks B*H*W*C < 2^31).
    int g_dy[RIN], g_dx[RIN], in_off[RIN];
    bool in_x2[RIN];
#pragma unroll
    for (int i = 0; i < RIN; ++i) {
        int g = tid + i * CONV_THREADS;
        g = g < NIN ? g : NIN - 1;
        const int pix = g / GPP, c = (g % GPP) * 8;
        const int dy1 = pix / WT, dx1 = pix % WT;
        g_dy[i] = dy1 - 1;
        g_dx[i] = dx1 - 1;
        in_x2[i] = c >= a.C1;
        in_off[i] = in_x2[i] ? (dy1 * a.W + dx1) * a.C2 + (c - a.C1) : (dy1 * a.W + dx1) * a.C1 + c;
    }
    // output side: lane (m, n) owns 4 channels from c0 of pixel (2 wv + n/2, 16 (n&1) + lane%16) of the tile
    bool o_cv[MT], o_has_src[MT], o_accm[MT];
    bf16 *o_yb[MT];
    const bf16 *o_sb[MT];
    int o_C[MT], o_e0[MT], o_loff[MT][NT];
    float o_scale[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int c0 = group * CM + m * 16 + (lane >> 4) * 4;
        const bool firstp = c0 < a.o1.C;
        const int cl = firstp ? c0 : c0 - a.o1.C;
        o_cv[m] = c0 < a.COUT;
        o_yb[m] = (firstp ? a.o1.y : a.o2.y) + cl;
        const bf16 *sb = firstp ? a.o1.relu_src : a.o2.relu_src;
        o_has_src[m] = sb != nullptr;
        o_sb[m] = sb + cl;
        o_C[m] = firstp ? a.o1.C : a.o2.C;
        o_scale[m] = firstp ? a.o1.scale : a.o2.scale;
        o_accm[m] = (firstp ? a.o1.accumulate : a.o2.accumulate) != 0;
        o_e0[m] = c0;
#pragma unroll
        for (int n = 0; n < NT; ++n) o_loff[m][n] = ((2 * wv + (n >> 1)) * a.W + (n & 1) * 16 + (lane & 15)) * o_C[m];
    }
    int lane_pix[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) lane_pix[n] = (2 * wv + (n >> 1)) * a.W + (n & 1) * 16 + (lane & 15);
    // POOL: 2x2 max-pool of the output in the same pass (nn.MaxPool2d(2,2) after the block's second conv,
    // icp_weight_policy.py:122-123).  The wave's two tile rows are one pooling row: the vertical maximum is
    // between two of its own accumulators, the horizontal one between neighbouring lanes (DPP); lane l stores
    // 4 channels of pooled pixel (wv, 8 (l & 1) + l%16 / 2) of the tile: even lanes the first 16-pixel half,
    // odd lanes the second.
    const int Hp = a.H >> 1, Wp = a.W >> 1;
    int p_loff[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) p_loff[m] = (wv * Wp + (lane & 1) * 8 + ((lane & 15) >> 1)) * a.COUT + o_e0[m];
    // C8 (exactly 8 output channels): the MFMA layout leaves the four accumulators of a wave half empty
    // (16-lane groups 2, 3 hold channels 8..15 that do not exist).  v_permlane16_swap + v_permlane32_swap
    // gather them into one full wave — lane = (n-tile, pixel) with all 8 channels — so that bias / ReLU /
    // dropout / conversions run once instead of four times and a tile row pair is one 16-byte store (and
    // one 16-byte load per epilogue operand) per lane.
    const int n8 = lane >> 4;
    const int lrow8 = 2 * wv + (n8 >> 1), lcol8 = (n8 & 1) * 16 + (lane & 15);
    const int lpix8 = lrow8 * a.W + lcol8;
    const bool c8_has_src = a.o1.relu_src != nullptr, c8_accm = a.o1.accumulate != 0;
    const float c8_scale = a.o1.scale;
    const float relu_lo = a.relu ? 0.f : -INFINITY;           // v = max(v, relu_lo): ReLU or identity
    const unsigned relu_pk = a.relu ? 0u : 0x80008000u;        // the same on packed bf16 bit patterns (v_pk_max_i16)
    // LDS read offsets of the B fragments: per-lane part (tap of the lane's k group) per k-step; the
    // N-tile part is an immediate
    int b_lane[NS];
#pragma unroll
    for (int ks = 0; ks < NS; ++ks) {
        int tap, ch;
        kslot<CK>(ks, lane, tap, ch);
        tap = tap > 8 ? 8 : tap;
        b_lane[ks] = ((2 * wv + tap / 3) * WT + (lane & 15) + tap % 3) * PK + ch;
    }

    // ---- the block's tile walk, on the scalar unit: tile `first + k nb` is (ld_b, ld_ty, ld_tx)
    const int adv_tx = nb % tiles_x, adv_ty = (nb / tiles_x) % tiles_y, adv_b = nb / tpi;
    int ld_b = first / tpi, ld_ty = (first % tpi) / tiles_x, ld_tx = first % tiles_x;
    int ld_left = nt_blk - 1;                                  // advances still allowed (then the walk parks)
    int s_pix0[RD], s_ty0[RD], s_tx0[RD];                      // first pixel / origin of the tile in each ring slot
    int cur_pix0 = 0, cur_ty0 = 0, cur_tx0 = 0;                // ... of the tile in LDS
    int s_pp0[RD], cur_pp0 = 0;                                // first pooled pixel of the tile (POOL)

#define MMK_RING_LOAD(SLOT)                                                                                  \
    {                                                                                                        \
        const int ty0_ = ld_ty * TH, tx0_ = ld_tx * TW;                                                      \
        const int pix0_ = (ld_b * a.H + ty0_) * a.W + tx0_;                                                  \
        s_pix0[SLOT] = pix0_; s_ty0[SLOT] = ty0_; s_tx0[SLOT] = tx0_;                                        \
        if constexpr (POOL) s_pp0[SLOT] = (ld_b * Hp + (ty0_ >> 1)) * Wp + (tx0_ >> 1);                      \
        const long org_ = (long)pix0_ - a.W - 1;                       /* halo origin pixel */               \
        const bf16 *base1_ = a.x1 + org_ * a.C1, *base2_ = a.x2 + org_ * a.C2;                               \
        _Pragma("unroll") for (int i = 0; i < RIN; ++i) {                                                    \
            const bool ok = (unsigned)(ty0_ + g_dy[i]) < (unsigned)a.H && (unsigned)(tx0_ + g_dx[i]) < (unsigned)a.W; \
            const bf16 *src = (in_x2[i] ? base2_ : base1_) + in_off[i];                                      \
            const u32x4 *sp = (ok && MMK_RING_DIAG != 2) ? reinterpret_cast<const u32x4 *>(src) : &g_zero16; \
            rin[SLOT][i] = *sp;                                                                              \
        }                                                                                                    \
        /* next tile of the walk (parks on the block's last tile: the ring keeps re-loading it) */           \
        const bool adv_ = ld_left > 0;                                                                       \
        ld_left -= adv_ ? 1 : 0;                                                                             \
        int ntx_ = ld_tx + adv_tx;                                                                           \
        const int cx_ = ntx_ >= tiles_x ? 1 : 0;                                                             \
        ntx_ -= cx_ ? tiles_x : 0;                                                                           \
        int nty_ = ld_ty + adv_ty + cx_;                                                                     \
        const int cy_ = nty_ >= tiles_y ? 1 : 0;                                                             \
        nty_ -= cy_ ? tiles_y : 0;                                                                           \
        ld_tx = adv_ ? ntx_ : ld_tx;                                                                         \
        ld_ty = adv_ ? nty_ : ld_ty;                                                                         \
        ld_b = adv_ ? ld_b + adv_b + cy_ : ld_b;                                                             \
    }
#define MMK_RING_SINKS()                                                                                     \
    {                                                                                                        \
        _Pragma("unroll") for (int i = 0; i < NST; ++i) reinterpret_cast<unsigned long long *>(g_sink16)[i] = 0ull; \
    }
#define MMK_RING_STORE(SLOT, BUF)                                                                            \
    {                                                                                                        \
        bf16 *dst_ = in_tile + (BUF) * (HT * WT * PK);                                                       \
        _Pragma("unroll") for (int i = 0; i < RIN; ++i) {                                                    \
            const int g = tid + i * CONV_THREADS;                                                            \
            if (g < NIN) *reinterpret_cast<u32x4 *>(dst_ + (size_t)(g / GPP) * PK + (g % GPP) * 8) = rin[SLOT][i]; \
        }                                                                                                    \
    }

    // weights: one block of NS x MT fragments, resident for the whole kernel
    {
        const u32x4 *wsrc = reinterpret_cast<const u32x4 *>(a.wpack + (size_t)group * NS * MT * 512);
#pragma unroll
        for (int i = 0; i < RW; ++i) {
            const int g = tid + i * CONV_THREADS;
            if (g < NW) reinterpret_cast<u32x4 *>(w_lds)[g] = wsrc[g];
        }
    }
    const DropoutParams dp = dropout_params(a.drop_p);
    // per-lane output channel bookkeeping (does not depend on the tile)
    float bs[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int c0 = group * CM + m * 16 + (lane >> 4) * 4;
        const float *bp = (a.bias && c0 < a.COUT) ? a.bias + c0 : reinterpret_cast<const float *>(&g_zero16);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            bs[m][r] = bp[r];
            // consumed here: a load still pending at the loop header would make every stage wait for
            // "everything older than the bias", i.e. for the whole ring
            asm volatile("" : "+v"(bs[m][r]));
        }
    }

    float bs8[8];
    if constexpr (C8) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            bs8[r] = a.bias ? a.bias[r] : 0.f;
            asm volatile("" : "+v"(bs8[r]));
        }
    }
    // Inverted dropout's factor 1 / keep rides in the bias addition: v = max(fma(acc, dk, bias dk), 0) (dk > 0 commutes with
    // the ReLU; dk = 1 without dropout, where the fma IS the addition), and the draws then only zero elements -- one
    // multiplication per output value less in kernels that are bound by vector-instruction issue.
    const float dk = a.drop_p > 0.f ? dp.inv_keep : 1.f;
    // both halves thr - 32768 - 1: (that) - draw, saturating, is negative exactly for the draws that are kept (pk_keep_mask)
    const unsigned thr1pk = ((unsigned)(dp.thr_s - 1 < -32768 ? -32768 : dp.thr_s - 1) & 0xffffu) * 0x00010001u;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) bs[m][r] *= dk;
    if constexpr (C8) {
#pragma unroll
        for (int r = 0; r < 8; ++r) bs8[r] *= dk;
    }

    // Fill the ring.  Sink stores stand in for the epilogues that have not run yet, so that the
    // first stage meets the same load/store queue as every later one:
    //   [tile k+1] NST stores [tile k+2] NST stores ... [tile k+RD] NST stores
    MMK_RING_LOAD(0);
    MMK_RING_LOAD(1);
    MMK_RING_SINKS();
    if constexpr (RD > 2) {
        MMK_RING_LOAD(2);
        MMK_RING_SINKS();
    }
    cur_pix0 = s_pix0[0]; cur_ty0 = s_ty0[0]; cur_tx0 = s_tx0[0];
    if constexpr (POOL) cur_pp0 = s_pp0[0];
    MMK_RING_STORE(0, 0);
    MMK_RING_LOAD(0);
    MMK_RING_SINKS();

    // The first round is peeled off the loop: hipcc merges the load/store queue state of every edge
    // into the loop header by its minimum, and the prologue's short queue would otherwise turn the
    // header's wait for tile k+1 into a drain of the whole ring on every round.
    // Round 5: and the loop itself runs WHOLE rounds only (no exit test between its stages), the last < RD tiles are a tail
    // behind it.  With `if (k >= nt_blk) goto ring_done` in front of every stage of the loop the compiler's single-exit form
    // of the loop was "skip this stage's body, raise a flag, go on to the latch": control-flow paths on which a stage's loads
    // and stores were not issued reach the loop header, the merged queue state there guaranteed only the last one or two
    // entries, and every stage waited with vmcnt(2) / (1) / (0) where (3 RD - 1) was meant -- the ring was one tile deep
    // (and the third stage also waited for the previous stage's store to complete).
    int k0 = 0;
    {
#define STAGE_CHECKED 1
#define STAGE_SLOT 0
#include "mmk_conv_ring_stage.inc"
#undef STAGE_SLOT
#define STAGE_SLOT 1
#include "mmk_conv_ring_stage.inc"
#undef STAGE_SLOT
        if constexpr (RD > 2) {
#define STAGE_SLOT 2
#include "mmk_conv_ring_stage.inc"
#undef STAGE_SLOT
        }
#undef STAGE_CHECKED
    }
    for (k0 = RD; k0 + RD <= nt_blk; k0 += RD) {
#define STAGE_CHECKED 0
#define STAGE_SLOT 0
#include "mmk_conv_ring_stage.inc"
#undef STAGE_SLOT
#define STAGE_SLOT 1
#include "mmk_conv_ring_stage.inc"
#undef STAGE_SLOT
        if constexpr (RD > 2) {
#define STAGE_SLOT 2
#include "mmk_conv_ring_stage.inc"
#undef STAGE_SLOT
        }
#undef STAGE_CHECKED
    }
    {   // tail: the last nt_blk % RD tiles
#define STAGE_CHECKED 1
#define STAGE_SLOT 0
#include "mmk_conv_ring_stage.inc"
#undef STAGE_SLOT
        if constexpr (RD > 2) {
#define STAGE_SLOT 1
#include "mmk_conv_ring_stage.inc"
#undef STAGE_SLOT
        }
#undef STAGE_CHECKED
    }
ring_done:;
#ifdef MMK_DEEP_STAMPS
    if (tid == 0) { stamp_lds[16 + 6] = __builtin_amdgcn_s_memtime(); stamp_lds[16 + 7] = __builtin_amdgcn_s_memrealtime(); }
    __syncthreads();
    // every block of the first 4 096: when it started and ended (100 MHz clock), behind the first 64 blocks' stamps
    if (g_deep_stamp_buf != nullptr && blockIdx.y == 0 && blockIdx.x < 4096 && tid == 0) {
        unsigned long long *life = g_deep_stamp_buf + (size_t)64 * (4 * MMK_STAMP_STAGES * 8) + (size_t)blockIdx.x * 2;
        life[0] = stamp_lds[8 + 7];
        life[1] = stamp_lds[16 + 7];
    }
    if (g_deep_stamp_buf != nullptr && blockIdx.y == 0 && blockIdx.x < 64)
        for (int i = tid; i < 4 * MMK_STAMP_STAGES * 8; i += CONV_THREADS)
            g_deep_stamp_buf[(size_t)blockIdx.x * (4 * MMK_STAMP_STAGES * 8) + i] = stamp_lds[i];
#endif
#undef MMK_RING_LOAD
#undef MMK_RING_STORE
}

template <int CK, int CM, int RD, bool EPI, int POOL = 0, bool C8 = false>
int launch_conv_ring(const ConvArgs &a, hipStream_t st)
{
#ifdef MMK_DEEP_STAMPS
    const size_t smem = (((size_t)2 * HT * WT * lds_pitch(CK) + (size_t)ksteps(CK) * (CM / 16) * 512) * sizeof(bf16) + 15) / 16 * 16 +
                        (size_t)4 * MMK_STAMP_STAGES * 8 * sizeof(unsigned long long);
#else
    const size_t smem = ((size_t)2 * HT * WT * lds_pitch(CK) + (size_t)ksteps(CK) * (CM / 16) * 512) * sizeof(bf16);
#endif
    static int per_cu[64] = {};     // resident blocks per CU (registers / LDS), per device
    int dev = 0;
    MMK_CHECK_HIP(hipGetDevice(&dev));
    if (per_cu[dev & 63] == 0) {
        if (smem > 64 * 1024)
            MMK_CHECK_HIP(hipFuncSetAttribute((const void *)conv3x3_ring_kernel<CK, CM, RD, EPI, POOL, C8>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        int nblk = 0;
        MMK_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, conv3x3_ring_kernel<CK, CM, RD, EPI, POOL, C8>, CONV_THREADS,
                                                                   smem));
        per_cu[dev & 63] = nblk < 1 ? 1 : (nblk > 8 ? 8 : nblk);
    }
    const int tiles = ((a.W + TW - 1) / TW) * ((a.H + TH - 1) / TH);
    const int groups = (a.COUT + CM - 1) / CM;
    const int total = tiles * a.B;
    const int per_xcd = (total + 7) / 8;
    int nb = (32 * per_cu[dev & 63]) / groups;                 // blocks per XCD (32 CUs each)
    nb = nb < 1 ? 1 : (nb > per_xcd ? per_xcd : nb);
    hipLaunchKernelGGL((conv3x3_ring_kernel<CK, CM, RD, EPI, POOL, C8>), dim3(8 * nb, groups), dim3(CONV_THREADS), smem, st, a, total,
                       per_xcd);
    MMK_LAUNCH_CHECK();
    return MMK_OK;
}

inline bool use_c8_epilogue()
{
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("MMK_CONV_C8");
        v = (e && e[0] == '0') ? 0 : 1;
    }
    return v == 1;
}

template <int CK, int CM>
int launch_conv_ring_epi(const ConvArgs &a, hipStream_t st)
{
    constexpr int RD = CK <= 16 ? 3 : 2;
    // with epilogue operands the in-order load queue keeps ~1 tile in flight whatever the ring holds:
    // two slots there, and the registers go to occupancy
    constexpr int RDE = 2;
    const bool epi = a.o1.relu_src || a.o1.accumulate || (a.o2.C > 0 && (a.o2.relu_src || a.o2.accumulate));
    if (a.pool_y != nullptr) {
        if constexpr (CK == CM && (CK == 16 || CK == 32)) {   // the encoder's second convs below 64 channels
            if (!epi && a.o2.C == 0)
                return a.pool_arg != nullptr ? launch_conv_ring<CK, CM, RD, false, 2>(a, st) : launch_conv_ring<CK, CM, RD, false, 1>(a, st);
        }
        mmk::set_error("mmk_conv3x3: pool_y is not supported for this layer (see mmk_conv3x3_pool_fusable)");
        return MMK_ERR_ARG;
    }
    if constexpr (CM == 16) {
        if (a.COUT == 8 && a.o2.C == 0 && use_c8_epilogue())
            return epi ? launch_conv_ring<CK, CM, RDE, true, false, true>(a, st) : launch_conv_ring<CK, CM, RD, false, false, true>(a, st);
    }
    return epi ? launch_conv_ring<CK, CM, RDE, true>(a, st) : launch_conv_ring<CK, CM, RD, false>(a, st);
}


// ------------------------------------------------------------------------------------------
// Layers with >= 64 channels: 8 waves per block (two per SIMD, so one wave's LDS traffic and
// epilogue overlap the other's MFMAs), input channels in chunks of 32 (one k-step per tap),
// up to 128 output channels per block, and a tile of TH rows x NT*16 pixels whose width is
// picked per layer (NT = 3 for 40-pixel rows, 5 for 80 / 160) so that no MFMA column is wasted
// on padding.  Wave (wm, wn) owns output channels [wm*MT*16, +MT*16) of tile row wn.
constexpr int DEEP_THREADS = 512;

template <int BM, int NT>
struct DeepCfg {
    static constexpr int WM = BM >= 128 ? 2 : 1;          // waves along the output channels
    static constexpr int WN = 8 / WM;                     // waves along the tile rows
    static constexpr int MT = BM / 16 / WM;               // 16-channel tiles per wave
    static constexpr int MTB = BM / 16;                   // ... per block
    static constexpr int TH = WN, TWD = NT * 16;
    static constexpr int HT = TH + 2, WT = TWD + 2;
    static constexpr int CK = 32, PK = 48, NS = 9, GPP = 4;
    static constexpr int NIN = HT * WT * GPP;
    static constexpr int NW = NS * MTB * 64;
    static constexpr int RIN = (NIN + DEEP_THREADS - 1) / DEEP_THREADS;
    static constexpr int RW = (NW + DEEP_THREADS - 1) / DEEP_THREADS;
    static constexpr size_t IN_BYTES = (size_t)HT * WT * PK * sizeof(bf16), W_BYTES = (size_t)NW * 8 * sizeof(bf16);
    // LDS: the halo tile, `wchunks` stages of packed weights (1, or all CIN / 32 of them when they stay resident), the bias
    static constexpr size_t smem_core(int wchunks) { return IN_BYTES + (size_t)wchunks * W_BYTES + BM * sizeof(float); }
#ifdef MMK_DEEP_STAMPS
    static constexpr size_t smem(int wchunks) { return (smem_core(wchunks) + 15) / 16 * 16 + (size_t)8 * MMK_STAMP_STAGES * 8 * sizeof(unsigned long long); }
#else
    static constexpr size_t smem(int wchunks) { return smem_core(wchunks); }
#endif
};

// Round 5.  What the round-4 kernel's stages spent their cycles on (s_memtime stamps of a diagnostic build, scripts/deep_stamps.py,
// 64 -> 64 at 160 x 160; per stage of ~11 500 cycles, with an epilogue every second stage): issuing the 12 prefetch loads of a
// thread 2 300-5 400 cycles (the wave sits in the issue of its loads until the CU's memory pipeline has taken them: it cannot
// start its MFMAs), the MFMA loop 4 800, waiting at the barrier for the waves that got their loads out last 1 700-2 500, writing
// the stage to LDS 1 200, the epilogue 6 500 forward / 11 000-12 600 data gradient, the second barrier 540-1 900: the matrix
// cores saw work a third of the time.  This version
//   * issues the prefetch loads INSIDE the MFMA loop, one or two per tap (a scheduling barrier per tap keeps them there), so no
//     wave waits in a load burst and all waves reach the barrier together;
//   * WRES: keeps the packed weights of ALL input-channel chunks in LDS for the whole launch where they fit beside the halo
//     tile (64 -> 64, 32 -> 64, 64 -> 32: 5 of the 12 loads and LDS writes of every stage gone, and 37 KB per stage and CU of L2
//     traffic with them);
//   * writes the next stage to LDS BEFORE the epilogue (the staging registers are dead during it);
//   * compile-time epilogue roles: R_FWD = bias + ReLU (+ dropout) on packed bf16 pairs (fma with the keep factor folded into the
//     bias, v_cvt_pk, ReLU as v_pk_max_i16, the dropout mask as saturating packed subtract + arithmetic shift of the hash words:
//     ~6.8 instead of ~11 vector instructions per output value); R_BWD = ReLU source OR accumulate target, fetched for all
//     n-tiles right behind the MFMA loop, in front of the barrier and the LDS write (fetched n-tile by n-tile each load sat behind
//     the previous n-tile's stores in the in-order memory counter: five load -> wait -> store round trips per tile); ROLE 0 =
//     everything, for the LeakyReLU network and other callers of the ABI.
// SUBW: the packed weights are laid out for blocks of a.wpack_mtb 16-channel tiles (a wider BM); this block's BM channels are a
// slice of one such group (the tail launches of dispatch_conv_deep's sub-batch split)
template <int BM, int NT, bool LK = false, bool SUBW = false, int ROLE = 0, bool WRES = false>
__global__ __launch_bounds__(DEEP_THREADS) void conv3x3_deep_kernel(const ConvArgs a, int total_tiles, int tiles_per_xcd)
{
    using C = DeepCfg<BM, NT>;
    constexpr int MT = C::MT, MTB = C::MTB, PK = C::PK, WT = C::WT, HT = C::HT, GPP = C::GPP;
    constexpr int NIN = C::NIN, NW = C::NW, RIN = C::RIN, RW = C::RW, NS = C::NS;
    constexpr bool R_FWD = ROLE == 1, R_BWD = ROLE == 2;
    static_assert(!(WRES && SUBW), "resident weights: the kernel's own packing only");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16 *in_tile = reinterpret_cast<bf16 *>(smem);
    bf16 *w_lds = in_tile + HT * WT * PK;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv / C::WN, wn = wv % C::WN;
    const int tiles_x = (a.W + C::TWD - 1) / C::TWD, tiles_y = (a.H + C::TH - 1) / C::TH;
    const int tpi = tiles_x * tiles_y;
    const int group = blockIdx.y;
    const int nchunk = a.CIN / 32;
    const int wchunks = WRES ? nchunk : 1;

    const int xcd = blockIdx.x & 7, nb = gridDim.x >> 3;
    const int t_begin = xcd * tiles_per_xcd;
    const int t_end = (t_begin + tiles_per_xcd < total_tiles) ? t_begin + tiles_per_xcd : total_tiles;
    int tile = t_begin + (blockIdx.x >> 3);
    if (tile >= t_end) return;

    constexpr int RWL = WRES ? 0 : RW;          // weight granules a thread prefetches per stage
    u32x4 rin[RIN], rw[RW];
    // Per lane and granule, once: position inside the halo tile.  Inside the loop an address is "uniform halo origin of the
    // tile (scalar unit) + 32-bit lane offset": a multiply-add, two compares and a pointer select per 16-byte load instead of a
    // 64-bit index computation (6 quarter-rate multiplies among ~45 instructions per load).
    int g_dy[RIN], g_dx[RIN], g_c[RIN], g_pix[RIN];
#pragma unroll
    for (int i = 0; i < RIN; ++i) {
        int g = tid + i * DEEP_THREADS;
        g = g < NIN ? g : NIN - 1;
        const int pix = g / GPP;
        g_dy[i] = pix / WT - 1;
        g_dx[i] = pix % WT - 1;
        g_c[i] = (g % GPP) * 8;
        g_pix[i] = (pix / WT) * a.W + pix % WT;       // pixel offset from the halo origin (row -1, column -1 of the tile)
    }
    // the stage a prefetch is for: wave-uniform, set once per stage
    const bf16 *ld_base = nullptr;
    int ld_xc = 0, ld_ty0 = 0, ld_tx0 = 0;
    const u32x4 *ld_w = nullptr;
    int ld_moff = 0;
    auto set_stage = [&](int t, int chunk) {
        const int b = t / tpi, tr = t - b * tpi;
        const int tyi = tr / tiles_x;
        ld_tx0 = (tr - tyi * tiles_x) * C::TWD;
        ld_ty0 = tyi * C::TH;
        const int c0 = chunk * 32;
        // a 32-channel chunk lies entirely in one of the two concatenated inputs
        const bool in1 = c0 < a.C1;
        const bf16 *xb = in1 ? a.x1 : a.x2;
        ld_xc = in1 ? a.C1 : a.C2;
        const int cb = in1 ? c0 : c0 - a.C1;
        const long org = ((long)b * a.H + ld_ty0 - 1) * a.W + ld_tx0 - 1;      // halo origin pixel (may lie outside the image)
        ld_base = xb + org * ld_xc + cb;
        if constexpr (SUBW) {
            const int pm = a.wpack_mtb, per = pm / MTB;           // tiles per packed group, kernel groups per packed group
            ld_w = reinterpret_cast<const u32x4 *>(a.wpack + ((size_t)((group / per) * nchunk + chunk)) * NS * pm * 512);
            ld_moff = (group % per) * MTB;
        } else {
            ld_w = reinterpret_cast<const u32x4 *>(a.wpack + ((size_t)(group * nchunk + chunk)) * NW * 8);
        }
    };
    auto issue_in = [&](int i) {               // (i: a compile-time constant at every call)
        const bool ok = (unsigned)(ld_ty0 + g_dy[i]) < (unsigned)a.H && (unsigned)(ld_tx0 + g_dx[i]) < (unsigned)a.W;
        const unsigned off = (unsigned)(g_pix[i] * ld_xc + g_c[i]);
        const u32x4 *sp = ok ? reinterpret_cast<const u32x4 *>(ld_base + off) : &g_zero16;
        rin[i] = *sp;
    };
    auto issue_w = [&](int i) {
        int g = tid + i * DEEP_THREADS;
        g = g < NW ? g : NW - 1;
        if constexpr (SUBW) rw[i] = ld_w[((g / (MTB * 64)) * a.wpack_mtb + ld_moff) * 64 + g % (MTB * 64)];
        else rw[i] = ld_w[g];
    };
    auto write_in = [&]() {
#pragma unroll
        for (int i = 0; i < RIN; ++i) {
            const int g = tid + i * DEEP_THREADS;
            if (g < NIN) *reinterpret_cast<u32x4 *>(in_tile + (size_t)(g / GPP) * PK + (g % GPP) * 8) = rin[i];
        }
    };
    auto write_w = [&](int slot) {
#pragma unroll
        for (int i = 0; i < RW; ++i) {
            const int g = tid + i * DEEP_THREADS;
            if (g < NW) reinterpret_cast<u32x4 *>(w_lds + (size_t)slot * NW * 8)[g] = rw[i];
        }
    };

    // The bias of the block's channels sits in LDS behind the weights and is added in the epilogue (the accumulators
    // start from zero).  A global load of it per tile, issued behind the epilogue's stores, made the first MFMA of the next
    // tile wait for those stores to drain: loads and stores share one counter and complete out of order with each other.
    // R_FWD: it is stored times the dropout's keep factor (the epilogue is one fma per value: acc * k + bias * k).
    const DropoutParams dp = dropout_params(a.drop_p);
    float *bias_lds = reinterpret_cast<float *>(w_lds + (size_t)wchunks * NW * 8);
    if (tid < BM) {
        int c = group * BM + tid;
        if constexpr (SUBW) {      // the block's channels are a strided slice of the packed group: 4 MT per lane group (see the epilogue)
            const int pm = a.wpack_mtb, per = pm / MTB, m_off = (group % per) * MTB;
            c = (group / per) * (pm * 16) + (m_off / 4) * 64 + 16 * (tid / (4 * MT)) + 4 * (m_off % 4) + tid % (4 * MT);
        }
        const float bz = (a.bias && c < a.COUT) ? a.bias[c] : 0.f;
        bias_lds[tid] = R_FWD ? bz * dp.inv_keep : bz;
    }
    f32x4 acc[MT][NT];
    auto reset_acc = [&]() {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    reset_acc();

    constexpr bool lk = LK;                 // LeakyReLU variant of the network (a.slope > 0)
    // fragment addresses that do not depend on the stage
    const bf16 *b_base = in_tile + ((size_t)(wn * WT + (lane & 15))) * PK + 8 * (lane >> 4);
    const bf16 *a_base0 = w_lds + ((size_t)(wm * MT) * 64 + lane) * 8;

    int chunk = 0;
#ifdef MMK_DEEP_STAMPS
    unsigned long long *stamp_lds = reinterpret_cast<unsigned long long *>(smem + (C::smem_core(wchunks) + 15) / 16 * 16);
    int stamp_stage = 0;
    for (int i = tid; i < 8 * MMK_STAMP_STAGES * 8; i += DEEP_THREADS) stamp_lds[i] = 0ull;
#endif
    // ---- prologue: the first stage (and, WRES, every chunk's weights) into LDS in front of the loop.  (Every later stage is
    // written at the loop's END: a write at the loop's head would merge the prologue's outstanding loads with the back edge's
    // outstanding epilogue stores in hipcc's counter bookkeeping, and the waits it then puts in front of the ds_writes make every
    // block drain its own output stores right behind issuing them.)
    if constexpr (WRES) {
        for (int ck = 0; ck < nchunk; ++ck) {
            set_stage(tile, ck);
#pragma unroll
            for (int i = 0; i < RW; ++i) issue_w(i);
            write_w(ck);
        }
    }
    set_stage(tile, 0);
#pragma unroll
    for (int i = 0; i < RIN; ++i) issue_in(i);
    if constexpr (!WRES) {
#pragma unroll
        for (int i = 0; i < RW; ++i) issue_w(i);
        write_w(0);
    }
    write_in();
    __syncthreads();

    constexpr int NCH = 4 * MT;                              // the lane's consecutive output channels per pixel
    constexpr int LW = NCH >= 8 ? 8 : 4;                     // channels per operand load / store (16 or 8 bytes)
    typedef __attribute__((ext_vector_type(LW))) __bf16 bfl;
    constexpr int NP = NCH / LW;

    while (true) {
        MMK_STAMP(0);
        int ntile = tile, nck = chunk + 1;
        if (nck == nchunk) {
            nck = 0;
            ntile = tile + nb;
        }
        const bool has_next = ntile < t_end;
        set_stage(has_next ? ntile : tile, nck);     // (clamped: the loads stay unconditional)
        const bf16 *a_base = a_base0 + (WRES ? (size_t)chunk * NW * 8 : (size_t)0);
        MMK_STAMP(1);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int ty = s / 3, tx = s % 3;
            bf16x8 bf[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n)
                bf[n] = *reinterpret_cast<const bf16x8 *>(b_base + ((size_t)(ty * WT + n * 16 + tx)) * PK);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const bf16x8 af = *reinterpret_cast<const bf16x8 *>(a_base + ((size_t)(s * MTB + m) * 64) * 8);
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf[n], acc[m][n], 0, 0, 0);
            }
            // this tap's share of the next stage's prefetch: loads s and s + 9 of the thread's RIN + RWL
#pragma unroll
            for (int j = s; j < RIN + RWL; j += NS) {
                if (j < RIN) issue_in(j);
                else issue_w(j - RIN);
            }
            __builtin_amdgcn_sched_barrier(0x38F);       // everything but vector-memory instructions may be scheduled across
        }
        MMK_STAMP(2);
        // Order of a tile's last stage: MFMA loop -> epilogue (its stores issued) -> delivery of the prefetch -> barrier -> LDS write
        // -> barrier.  The prefetch loads are OLDER than the stores in the in-order memory counter, so their delivery is a counted
        // wait that leaves the stores in flight (straight-line stores: hipcc knows the count); the stores then have the two
        // barriers and the LDS write to complete before the next stage overwrites the registers they read (a store reads its data
        // and address registers when the memory pipeline takes it, and hipcc makes the next writer of such a register wait for
        // the store to complete).
        auto deliver_and_write = [&]() {
#pragma unroll
            for (int i = 0; i < RIN; ++i) asm volatile("" : "+v"(rin[i]));
#pragma unroll
            for (int i = 0; i < RWL; ++i) asm volatile("" : "+v"(rw[i]));
            MMK_STAMP(4);
            if (has_next) {
                __syncthreads();          // every wave is done reading this stage's fragments
                MMK_STAMP(5);
                write_in();
                if constexpr (!WRES) write_w(0);
            }
            MMK_STAMP(6);
        };
        if (chunk != nchunk - 1) {
            MMK_STAMP(3);
            deliver_and_write();
        } else {
            // ---- per-lane output addressing of this tile.  The weights are packed so that the lane's 4 rows of each of its MT
            // tiles are NCH = 4 MT consecutive output channels (pack_conv_weight_elem): tile m, register r = channel c0 + 4 m + r.
            // The lane therefore stores -- and reads its ReLU source / accumulate target as -- one contiguous run of 2 NCH bytes
            // per pixel, with no lane exchange.
            const int b = tile / tpi, tr = tile - b * tpi;
            const int tyi = tr / tiles_x;
            const int tx0 = (tr - tyi * tiles_x) * C::TWD, yy = tyi * C::TH + wn;
            int lv = lane;
            asm volatile("" : "+v"(lv));        // (derived here, once per tile, instead of living in registers across the MFMA loop)
            const int g4 = lv >> 4;
            int c0, e_bias_at;                               // first of the lane's NCH channels, and where their bias sits
            if constexpr (SUBW) {
                const int pm = a.wpack_mtb, per = pm / MTB, m_off = (group % per) * MTB;
                c0 = (group / per) * (pm * 16) + (m_off / 4) * 64 + 16 * g4 + 4 * (m_off % 4);
                e_bias_at = NCH * g4;
            } else {
                c0 = group * BM + wm * (16 * MT) + NCH * g4;
                e_bias_at = wm * (16 * MT) + NCH * g4;
            }
            const bool e_on = c0 < a.COUT;
            const bool firstp = c0 < a.o1.C;
            bf16 *const o_y = firstp ? a.o1.y : a.o2.y;
            const bf16 *const o_src = firstp ? a.o1.relu_src : a.o2.relu_src;
            const int o_C = firstp ? a.o1.C : a.o2.C;
            const bool e_acc = (firstp ? a.o1.accumulate : a.o2.accumulate) != 0;
            const float e_scale = firstp ? a.o1.scale : a.o2.scale;
            const int cl = firstp ? c0 : c0 - a.o1.C;
            const int e_x0 = tx0 + (lv & 15);
            const long p0 = ((long)b * a.H + yy) * a.W + e_x0;                 // the lane's pixel of n-tile 0
            const unsigned e_e0 = (unsigned)p0 * (unsigned)a.COUT + (unsigned)c0 + a.hash_base;       // dropout element index
            bf16 *const e_y0 = o_y + p0 * o_C + cl;
            const int e_step = 16 * o_C;                                        // elements between the pixels of consecutive n-tiles
            const bool e_row_ok = yy < a.H;
            const bool e_src = o_src != nullptr;
            const bf16 *const e_pf0 = e_src ? o_src + p0 * o_C + cl : (e_acc ? (const bf16 *)e_y0 : nullptr);
            // the epilogue's operands (ReLU source, or accumulate target) of ALL n-tiles in front of the first store.  Every slot is
            // written (lanes without an operand read zeros): nothing of it is live outside this branch.
            bfl opnd[R_BWD ? NT : 1][NP];
            auto fetch_operands = [&]() {
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const bool okp = e_on && e_pf0 != nullptr && e_row_ok && e_x0 + n * 16 < a.W;
                    const bf16 *pp = okp ? e_pf0 + n * e_step : reinterpret_cast<const bf16 *>(g_zero32);
#pragma unroll
                    for (int k = 0; k < NP; ++k) opnd[R_BWD ? n : 0][k] = *reinterpret_cast<const bfl *>(pp + k * LW);
                }
            };
            if constexpr (R_BWD) fetch_operands();
            if (e_on) {
                if constexpr (R_FWD) {
                    // ---- forward: y = dropout(relu(acc + bias)) as packed bf16 pairs
                    float bk[NCH];
#pragma unroll
                    for (int k = 0; k < NCH; k += 4) {
                        const f32x4 t = *reinterpret_cast<const f32x4 *>(bias_lds + e_bias_at + k);
                        bk[k] = t[0]; bk[k + 1] = t[1]; bk[k + 2] = t[2]; bk[k + 3] = t[3];
                    }
                    const float kf = dp.inv_keep;
                    // keep <=> (signed 16-bit draw) >= thr - 32768: (thr - 32768 - 1) - draw, saturating, is negative exactly then
                    // (thr = 0, no dropout: the subtrahend saturates at -32768, which the no-dropout draws of +32767 pass)
                    const int t1s = (int)dp.thr - 32768 - 1;
                    const unsigned t1 = (unsigned)(t1s < -32768 ? -32768 : t1s) & 0xffffu;
                    const unsigned thr1 = t1 | (t1 << 16);
                    const bool drop = a.drop_p > 0.f;
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        const bool okp = e_row_ok && e_x0 + n * 16 < a.W;
                        // the hash words of the n-tile's MT channel groups; without dropout: draws of +32767, which every
                        // threshold keeps (one scalar branch per n-tile)
                        unsigned hw[MT][2];
#pragma unroll
                        for (int m = 0; m < MT; ++m) hw[m][0] = hw[m][1] = 0x7fff7fffu;
                        if (drop) {
#pragma unroll
                            for (int m = 0; m < MT; ++m)
                                dropout_words(a.seed, e_e0 + (unsigned)n * (16u * (unsigned)a.COUT) + 4u * m, hw[m][0], hw[m][1]);
                        }
                        unsigned pk[NCH / 2];
#pragma unroll
                        for (int m = 0; m < MT; ++m)
#pragma unroll
                            for (int hlf = 0; hlf < 2; ++hlf) {
                                const float t0 = __builtin_fmaf(acc[m][n][2 * hlf], kf, bk[4 * m + 2 * hlf]);
                                const float t1f = __builtin_fmaf(acc[m][n][2 * hlf + 1], kf, bk[4 * m + 2 * hlf + 1]);
                                pk[2 * m + hlf] = pk_relu_bf16(cvt_pk_bf16(t0, t1f)) & pk_keep_mask(hw[m][hlf], thr1);
                            }
                        // (unconditional, straight-line stores: pixels outside the image go to a sink)
                        bf16 *const dst = okp ? e_y0 + n * e_step : reinterpret_cast<bf16 *>(g_sink32);
                        if constexpr (NCH >= 8) {
#pragma unroll
                            for (int k = 0; k < NCH / 2; k += 4)
                                *reinterpret_cast<u32x4 *>(dst + 2 * k) = (u32x4){pk[k], pk[k + 1], pk[k + 2], pk[k + 3]};
                        } else {
                            *reinterpret_cast<unsigned long long *>(dst) = ((unsigned long long)pk[1] << 32) | pk[0];
                        }
                        __builtin_amdgcn_sched_barrier(0);     // one n-tile at a time: bounds the live temporaries
                    }
                } else {
                    // ---- data gradient (R_BWD) / generic epilogue
                    float bv[NCH];
                    if constexpr (!R_BWD) {
#pragma unroll
                        for (int k = 0; k < NCH; k += 4) {
                            const f32x4 t = *reinterpret_cast<const f32x4 *>(bias_lds + e_bias_at + k);
                            bv[k] = t[0]; bv[k + 1] = t[1]; bv[k + 2] = t[2]; bv[k + 3] = t[3];
                        }
                    }
                    const bool o_accf = !e_src && e_acc;            // R_BWD: the operand slots hold the accumulate target
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        const bool okp = e_row_ok && e_x0 + n * 16 < a.W;
                        float v[NCH];
#pragma unroll
                        for (int m = 0; m < MT; ++m)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                float t = acc[m][n][r];
                                if constexpr (!R_BWD) {
                                    t += bv[4 * m + r];
                                    if (a.relu) t = lk ? act_leaky(t, a.slope) : fmaxf(t, 0.f);
                                }
                                v[4 * m + r] = t;
                            }
                        if constexpr (!R_BWD) {
                            if (a.drop_p > 0.f) {
#pragma unroll
                                for (int m = 0; m < MT; ++m) {
                                    float sc[4];
                                    dropout_scale4(a.seed, e_e0 + (unsigned)n * (16u * (unsigned)a.COUT) + 4u * m, dp, sc);
#pragma unroll
                                    for (int r = 0; r < 4; ++r) v[4 * m + r] = lk ? drop_leaky(v[4 * m + r], sc[r]) : v[4 * m + r] * sc[r];
                                }
                            }
                        }
                        bf16 *const dst0 = e_y0 + n * e_step;
                        if constexpr (R_BWD) {
                            if (e_pf0 != nullptr && okp) {
                                if (!o_accf) {
#pragma unroll
                                    for (int k = 0; k < NCH; k += LW) {
                                        const bfl sv = opnd[n][k / LW];
#pragma unroll
                                        for (int r = 0; r < LW; ++r) v[k + r] = ((float)sv[r] > 0.f) ? v[k + r] * e_scale : 0.f;
                                    }
                                } else {
#pragma unroll
                                    for (int k = 0; k < NCH; k += LW) {
                                        const bfl ov = opnd[n][k / LW];
#pragma unroll
                                        for (int r = 0; r < LW; ++r) v[k + r] += (float)ov[r];
                                    }
                                }
                            }
                        } else {
                            if (e_src && okp) {
                                const bf16 *sp = e_pf0 + n * e_step;
#pragma unroll
                                for (int k = 0; k < NCH; k += LW) {
                                    const bfl sv = *reinterpret_cast<const bfl *>(sp + k);
#pragma unroll
                                    for (int r = 0; r < LW; ++r)
                                        v[k + r] = lk ? v[k + r] * bwd_factor_leaky((float)sv[r], e_scale, a.slope)
                                                      : (((float)sv[r] > 0.f) ? v[k + r] * e_scale : 0.f);
                                }
                            }
                            if (e_acc && okp) {
#pragma unroll
                                for (int k = 0; k < NCH; k += LW) {
                                    const bfl ov = *reinterpret_cast<const bfl *>(dst0 + k);
#pragma unroll
                                    for (int r = 0; r < LW; ++r) v[k + r] += (float)ov[r];
                                }
                            }
                        }
                        if (R_BWD || okp) {
                            bf16 *const dst = (R_BWD && !okp) ? reinterpret_cast<bf16 *>(g_sink32) : dst0;
                            if constexpr (NCH >= 8) {
#pragma unroll
                                for (int k = 0; k < NCH; k += 8) {
                                    bf16x8 o8;
#pragma unroll
                                    for (int r = 0; r < 8; ++r) o8[r] = (bf16)v[k + r];
                                    *reinterpret_cast<bf16x8 *>(dst + k) = o8;
                                }
                            } else {
                                bf16x4 o4;
#pragma unroll
                                for (int r = 0; r < 4; ++r) o4[r] = (bf16)v[r];
                                *reinterpret_cast<bf16x4 *>(dst) = o4;
                            }
                        }
                    }
                }
            }
            reset_acc();
            MMK_STAMP(3);
            deliver_and_write();
        }
        if (!has_next) break;
        __syncthreads();
        MMK_STAMP(7);
#ifdef MMK_DEEP_STAMPS
        ++stamp_stage;
#endif
        tile = ntile;
        chunk = nck;
    }
#ifdef MMK_DEEP_STAMPS
    __syncthreads();
    if (g_deep_stamp_buf != nullptr && blockIdx.y == 0 && blockIdx.x < 64)
        for (int i = tid; i < 8 * MMK_STAMP_STAGES * 8; i += DEEP_THREADS)
            g_deep_stamp_buf[(size_t)blockIdx.x * (8 * MMK_STAMP_STAGES * 8) + i] = stamp_lds[i];
#endif
}

template <int BM, int NT, bool LK = false, bool SUBW = false, int ROLE = 0, bool WRES = false>
int launch_conv_deep(const ConvArgs &a, hipStream_t st)
{
    using C = DeepCfg<BM, NT>;
    static bool attr_set[64] = {};
    int dev = 0;
    MMK_CHECK_HIP(hipGetDevice(&dev));
    if (!attr_set[dev & 63]) {
        MMK_CHECK_HIP(hipFuncSetAttribute((const void *)conv3x3_deep_kernel<BM, NT, LK, SUBW, ROLE, WRES>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set[dev & 63] = true;
    }
    const size_t smem = C::smem(WRES ? a.CIN / 32 : 1);
    MMK_REQUIRE(smem <= (size_t)160 * 1024, "mmk_conv3x3: %zu bytes of LDS for CIN=%d COUT=%d", smem, a.CIN, a.COUT);
    const int tiles = ((a.W + C::TWD - 1) / C::TWD) * ((a.H + C::TH - 1) / C::TH);
    const int groups = (a.COUT + BM - 1) / BM;
    const int total = tiles * a.B;
    const int per_xcd = (total + 7) / 8;
    int nb = 32 / groups;                                    // one 8-wave block per CU, 32 CUs per XCD
    nb = nb < 1 ? 1 : (nb > per_xcd ? per_xcd : nb);
    hipLaunchKernelGGL((conv3x3_deep_kernel<BM, NT, LK, SUBW, ROLE, WRES>), dim3(8 * nb, groups), dim3(DEEP_THREADS), smem, st, a, total, per_xcd);
    MMK_LAUNCH_CHECK();
    return MMK_OK;
}

// the packed weights of every input-channel chunk stay in LDS beside the halo tile (conv3x3_deep_kernel: WRES)
template <int BM, int NT>
bool deep_weights_fit(int cin) { return DeepCfg<BM, NT>::smem(cin / 32) <= (size_t)160 * 1024; }

// Rounds of tiles the persistent grid of the plain launch walks (one 8-wave block per CU): blocks of an XCD share its
// contiguous range of tiles.
int deep_rounds(int B, int H, int W, int cout, int BM, int NT, int *tiles_per_image)
{
    const int TH = BM >= 128 ? 4 : 8;
    const int tpi = ((W + NT * 16 - 1) / (NT * 16)) * ((H + TH - 1) / TH);
    const int groups = (cout + BM - 1) / BM;
    const int nb = std::max(1, 32 / groups), per_xcd = (B * tpi + 7) / 8;
    if (tiles_per_image) *tiles_per_image = tpi;
    return (per_xcd + nb - 1) / nb;
}

int dispatch_conv_deep_plain(const ConvArgs &a, hipStream_t st);

// Tile quantisation (round 4).  At the 80 x 80 and 40 x 40 levels a launch is a few hundred tiles on 256 CUs: 320 tiles are two
// rounds of which the second keeps 64 CUs busy (dec0.x, dec1.x forward, four data gradients: 62 % of the chip over the launch).
// Such a launch is split by images: the first images fill exactly one round with the layer's own kernel, the rest run as a
// second launch of 32-channel blocks (4 x / 2 x the blocks for a quarter / half of the work each), which fits one round of its
// own.  Same arithmetic per output element (the k order does not depend on the block width) and the same dropout draws (the
// element index continues across the split: ConvArgs::hash_base): results are bit-identical to the single launch.  Measured in
// one process (scripts/ab_env.py MMK_CONV_SPLIT): dec0.0 54 -> 48 us, dec0.2 32 -> 30, dec1.0 50 -> 47, dec1.2 33 -> 31.
int dispatch_conv_deep(const ConvArgs &a, hipStream_t st)
{
    const int BM = conv_cm(a.CIN, a.COUT);
    const bool narrow = a.W <= 48;
    const int NT = narrow ? 3 : 5;
    const char *split_env = getenv("MMK_CONV_SPLIT");       // MMK_CONV_SPLIT=0: always one launch (A/B, bit-identity test; read per call)
    const bool split_on = !(split_env && split_env[0] == '0');
    int tpi = 0;
    // (forward launches only: in the backward pass the idle CUs of a data-gradient launch's second round are not idle -- the
    // weight-gradient kernels of the side stream run there -- and the split cost the pass 0.15 ms instead of saving 0.01)
    const bool forward_role = a.relu != 0 && a.o1.relu_src == nullptr && a.o2.relu_src == nullptr && !a.o1.accumulate && !a.o2.accumulate;
    if (split_on && forward_role && a.slope == 0.f && (BM == 64 || BM == 128) && a.COUT % 32 == 0 && a.pool_y == nullptr && a.wpack_mtb == 0 &&
        deep_rounds(a.B, a.H, a.W, a.COUT, BM, NT, &tpi) == 2) {
        const int groups = (a.COUT + BM - 1) / BM, nb = std::max(1, 32 / groups);
        const int Bm = (8 * nb) / tpi;                      // images that fill one round
        const int Bt = a.B - Bm;
        const bool under_filled = (long)a.B * tpi * groups * 10 < (long)256 * 2 * 7;          // < 70 % of two rounds
        if (Bm >= 1 && Bt >= 1 && under_filled && deep_rounds(Bm, a.H, a.W, a.COUT, BM, NT, nullptr) == 1 &&
            deep_rounds(Bt, a.H, a.W, a.COUT, 32, NT, nullptr) == 1 && (a.o1.C % 32 == 0) && (a.o2.C % 32 == 0)) {
            ConvArgs m = a;
            m.B = Bm;
            int rc = dispatch_conv_deep_plain(m, st);
            if (rc != MMK_OK) return rc;
            ConvArgs t = a;
            const size_t px = (size_t)Bm * a.H * a.W;
            t.B = Bt;
            t.x1 = a.x1 + px * a.C1;
            if (a.x2) t.x2 = a.x2 + px * a.C2;
            auto adv = [&](ConvOutPart &o) {
                if (o.y) o.y += px * o.C;
                if (o.relu_src) o.relu_src += px * o.C;
            };
            adv(t.o1);
            adv(t.o2);
            t.hash_base = a.hash_base + (unsigned)(px * a.COUT);
            t.wpack_mtb = BM / 16;
            return narrow ? launch_conv_deep<32, 3, false, true, 1>(t, st) : launch_conv_deep<32, 5, false, true, 1>(t, st);
        }
    }
    return dispatch_conv_deep_plain(a, st);
}

int dispatch_conv_deep_plain(const ConvArgs &a, hipStream_t st)
{
    const int BM = conv_cm(a.CIN, a.COUT);
    const bool narrow = a.W <= 48;          // NT = 3 (48-pixel tile rows) wastes less than NT = 5 there
    if (a.slope > 0.f) {
#define MMK_DEEP_CASE(M) if (BM == M) return narrow ? launch_conv_deep<M, 3, true>(a, st) : launch_conv_deep<M, 5, true>(a, st)
        MMK_DEEP_CASE(16); MMK_DEEP_CASE(32); MMK_DEEP_CASE(64); MMK_DEEP_CASE(128);
#undef MMK_DEEP_CASE
    }
    // the epilogue's role (conv3x3_deep_kernel: ROLE): forward = activation, no epilogue operand; backward = no bias / activation /
    // dropout and at most ONE operand kind (ReLU source or accumulate target) per output half; anything else: the generic kernel
    const bool any_src = a.o1.relu_src != nullptr || (a.o2.C > 0 && a.o2.relu_src != nullptr);
    const bool any_acc = a.o1.accumulate != 0 || (a.o2.C > 0 && a.o2.accumulate != 0);
    const bool both1 = a.o1.relu_src != nullptr && a.o1.accumulate != 0, both2 = a.o2.C > 0 && a.o2.relu_src != nullptr && a.o2.accumulate != 0;
    const int role = (a.relu != 0 && !any_src && !any_acc) ? 1
                   : ((a.relu == 0 && a.bias == nullptr && a.drop_p == 0.f && !both1 && !both2) ? 2 : 0);
#define MMK_DEEP_CASE(M, R)                                                                                                        \
    if (BM == M && role == R) {                                                                                                    \
        if (M <= 64 && a.wpack_mtb == 0 && (narrow ? deep_weights_fit<M, 3>(a.CIN) : deep_weights_fit<M, 5>(a.CIN)))               \
            return narrow ? launch_conv_deep<M, 3, false, false, R, (M <= 64)>(a, st) : launch_conv_deep<M, 5, false, false, R, (M <= 64)>(a, st); \
        return narrow ? launch_conv_deep<M, 3, false, false, R>(a, st) : launch_conv_deep<M, 5, false, false, R>(a, st);              \
    }
    MMK_DEEP_CASE(32, 1); MMK_DEEP_CASE(64, 1); MMK_DEEP_CASE(128, 1);
    MMK_DEEP_CASE(32, 2); MMK_DEEP_CASE(64, 2); MMK_DEEP_CASE(128, 2);
#undef MMK_DEEP_CASE
#define MMK_DEEP_CASE(M) if (BM == M) return narrow ? launch_conv_deep<M, 3>(a, st) : launch_conv_deep<M, 5>(a, st)
    MMK_DEEP_CASE(16); MMK_DEEP_CASE(32); MMK_DEEP_CASE(64); MMK_DEEP_CASE(128);
#undef MMK_DEEP_CASE
    mmk::set_error("mmk_conv3x3: unsupported channel counts CIN=%d COUT=%d", a.CIN, a.COUT);
    return MMK_ERR_ARG;
}

// the layers whose producing kernel can write the 2x2 max-pool of its output as well
bool pool_fusable(int cin, int cout, int B, int H, int W)
{
    const bool fits32 = (size_t)B * H * W * (size_t)std::max(cin, cout) < ((size_t)1 << 31);
    return cin == cout && (cin == 16 || cin == 32) && fits32 && H >= 2 && W >= 2;
}

int dispatch_conv(const ConvArgs &a, hipStream_t st)
{
    if (a.pool_y != nullptr && (a.slope > 0.f || !pool_fusable(a.CIN, a.COUT, a.B, a.H, a.W))) {
        mmk::set_error("mmk_conv3x3: pool_y is not supported for this layer (see mmk_conv3x3_pool_fusable)");
        return MMK_ERR_ARG;
    }
    if (conv_is_deep(a.CIN, a.COUT)) return dispatch_conv_deep(a, st);
    const int CK = conv_ck(a.CIN), CM = conv_cm(a.CIN, a.COUT);
    const bool fits32 = (size_t)a.B * a.H * a.W * (size_t)std::max(a.CIN, a.COUT) < ((size_t)1 << 31);
    // (the LeakyReLU variant of the network runs the thin layers on the plain pipelined kernel: the ring
    // kernel's straight-line epilogue is tuned for the reference's default configuration)
    if (a.slope > 0.f) {
#define MMK_CONV_CASE(K, M) if (CK == K && CM == M) return launch_conv<K, M, true>(a, st)
        MMK_CONV_CASE(8, 16); MMK_CONV_CASE(8, 32); MMK_CONV_CASE(8, 64);
        MMK_CONV_CASE(16, 16); MMK_CONV_CASE(16, 32); MMK_CONV_CASE(16, 64);
        MMK_CONV_CASE(32, 16); MMK_CONV_CASE(32, 32);
#undef MMK_CONV_CASE
    }
    if (a.CIN == CK && CM <= 32 && fits32) {
#define MMK_RING_CASE(K, M) if (CK == K && CM == M) return launch_conv_ring_epi<K, M>(a, st)
        MMK_RING_CASE(8, 16); MMK_RING_CASE(8, 32); MMK_RING_CASE(16, 16); MMK_RING_CASE(16, 32);
        MMK_RING_CASE(32, 16); MMK_RING_CASE(32, 32);
#undef MMK_RING_CASE
    }
#define MMK_CONV_CASE(K, M) if (CK == K && CM == M) return launch_conv<K, M>(a, st)
    MMK_CONV_CASE(8, 16); MMK_CONV_CASE(8, 32); MMK_CONV_CASE(8, 64);
    MMK_CONV_CASE(16, 16); MMK_CONV_CASE(16, 32); MMK_CONV_CASE(16, 64);
    MMK_CONV_CASE(32, 16); MMK_CONV_CASE(32, 32);
#undef MMK_CONV_CASE
    mmk::set_error("mmk_conv3x3: unsupported channel counts CIN=%d COUT=%d", a.CIN, a.COUT);
    return MMK_ERR_ARG;
}


// ------------------------------------------------------------------------------------------
// Weight gradient: dW[co][ci][tap] = sum over pixels of g[p][co] * x[p + tap][ci].
// GEMM with the contraction over pixels (k = 32 consecutive pixels of one tile row):
// A = g^T (rows = co), B = shifted x (cols = (tap, ci)); both fragments come out of the
// row-major [pixel][channel] LDS tiles through ds_read_b64_tr_b16 (4 pixels x 16 channels
// per 16-lane group, delivered channel-major).  Blocks are persistent over spatial tiles
// and add their fp32 partial sums once at the end into dWt[tap][co][ci] (64-byte
// contiguous atomic segments); the bias gradient rides along as an extra all-ones column.
typedef __attribute__((ext_vector_type(2))) int i32x2;
typedef __attribute__((ext_vector_type(4))) int i32x4;

__device__ __forceinline__ unsigned lds_addr(const void *p)
{
    return (unsigned)(size_t)(__attribute__((address_space(3))) const void *)p;
}

__device__ __forceinline__ i32x2 tr_read(unsigned addr)
{
    i32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}

// same with the offset in the instruction's 16-bit immediate field (no address register per read)
template <int OFF>
__device__ __forceinline__ i32x2 tr_read_o(unsigned addr)
{
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
    i32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}

__device__ __forceinline__ bf16x8 frag_from(i32x2 lo, i32x2 hi)
{
    i32x4 t = {lo.x, lo.y, hi.x, hi.y};
    return __builtin_bit_cast(bf16x8, t);
}

struct WgradArgs {
    const bf16 *x1, *x2;
    int C1, C2;
    const bf16 *g;   // (B,H,W,COUT)
    int B, H, W, CIN, COUT;
    float *partials; // [gridDim.x][9 * COUT * CIN + COUT]: per-block partial sums (weights, then bias), plain stores
    int acc_partials;   // add to what `partials` holds (second application of shared weights)
};

template <int CK, int CM, bool G8>
__global__ __launch_bounds__(CONV_THREADS) void conv3x3_wgrad_kernel(const WgradArgs a)
{
    // G8: the gradient tensor has 8 channels (COUT = 8, CM = 16): its LDS pitch is 8 elements
    constexpr int MT = CM / 16;
    constexpr int NTT = (CK >= 16) ? 9 * (CK / 16) : 5;  // n-tiles of 16 (tap, ci) columns
    constexpr int NTW = (NTT + 3) / 4;                    // per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int PK = wg_pitch(CK);
    constexpr int GCOLS = G8 ? 8 : CM;                    // channels of the g tile
    constexpr int PG = wg_pitch(GCOLS);
    bf16 *x_tile = reinterpret_cast<bf16 *>(smem);                 // HT*WT*PK (+ pad)
    bf16 *g_tile = x_tile + (HT * WT + 8) * PK;                    // TH*TW*PG (+ pad)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: branches on it are scalar
    const int chunk = blockIdx.y, group = blockIdx.z;
    const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
    const int tpi = tiles_x * tiles_y;
    const int total_tiles = tpi * a.B;
    const int i16 = lane & 15, g4 = lane >> 4, q = i16 >> 2, pp = i16 & 3;

    f32x4 acc[MT][NTW];
    f32x4 accb[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        accb[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int n = 0; n < NTW; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (bf16)1.0f;

    // per-lane LDS addresses that do not depend on the tile row (the row goes into the read's
    // immediate offset)
    unsigned b_ad[NTW];
#pragma unroll
    for (int n = 0; n < NTW; ++n) {
        const int nt = wv + 4 * n;
        int tap, col;
        if (CK >= 16) {
            tap = nt / (CK / 16);
            col = (nt % (CK / 16)) * 16 + 4 * pp;
        } else {
            tap = 2 * nt + (pp >> 1);
            col = 4 * (pp & 1);
        }
        tap = tap > 8 ? 8 : tap;
        const int ty = tap / 3, tx = tap % 3;
        b_ad[n] = lds_addr(x_tile) + (unsigned)((((ty * WT) + 4 * g4 + q + tx) * PK + col) * 2);
    }
    const unsigned g_ad = lds_addr(g_tile) + (unsigned)(((4 * g4 + q) * PG + 4 * pp) * 2);

    // register prefetch of the next tile's operands while the current one is consumed; every load
    // is unconditional (out-of-image granules read the zero word)
    constexpr int GPP = CK / 8;
    constexpr int NIN = HT * WT * GPP;
    constexpr int RIN = (NIN + CONV_THREADS - 1) / CONV_THREADS;
    constexpr int GPG = GCOLS / 8;
    constexpr int NG = TH * TW * GPG;
    constexpr int RG = (NG + CONV_THREADS - 1) / CONV_THREADS;
    u32x4 rin[RIN], rg[RG];
    auto load_tile = [&](int b, int tyi, int txi) {
        int tv = tid;                       // opaque: per-granule offsets are recomputed, not kept live
        asm volatile("" : "+v"(tv));
        const int tx0 = txi * TW, ty0 = tyi * TH;
        const int pix0 = (b * a.H + ty0) * a.W + tx0;
#pragma unroll
        for (int i = 0; i < RIN; ++i) {
            int gi = tv + i * CONV_THREADS;
            gi = gi < NIN ? gi : NIN - 1;
            const int pix = gi / GPP, gc = gi % GPP;
            const int dy = pix / WT - 1, dx = pix % WT - 1;
            const bool ok = (unsigned)(ty0 + dy) < (unsigned)a.H && (unsigned)(tx0 + dx) < (unsigned)a.W;
            const int c = chunk * CK + gc * 8;
            const int p = pix0 + dy * a.W + dx;
            const bf16 *src = (c < a.C1) ? a.x1 + ((long)p * a.C1 + c) : a.x2 + ((long)p * a.C2 + (c - a.C1));
            const u32x4 *sp = ok ? reinterpret_cast<const u32x4 *>(src) : &g_zero16;
            rin[i] = *sp;
        }
#pragma unroll
        for (int i = 0; i < RG; ++i) {
            int gi = tv + i * CONV_THREADS;
            gi = gi < NG ? gi : NG - 1;
            const int pix = gi / GPG, gc = gi % GPG;
            const int dy = pix / TW, dx = pix % TW;
            const bool ok = ty0 + dy < a.H && tx0 + dx < a.W;
            const int p = pix0 + dy * a.W + dx;
            const u32x4 *sp = ok ? reinterpret_cast<const u32x4 *>(a.g + ((long)p * a.COUT + group * CM + gc * 8)) : &g_zero16;
            rg[i] = *sp;
        }
    };
    auto store_tile = [&]() {
        int tv = tid;
        asm volatile("" : "+v"(tv));
#pragma unroll
        for (int i = 0; i < RIN; ++i) {
            const int gi = tv + i * CONV_THREADS;
            if (gi < NIN) *reinterpret_cast<u32x4 *>(x_tile + (size_t)(gi / GPP) * PK + (gi % GPP) * 8) = rin[i];
        }
#pragma unroll
        for (int i = 0; i < RG; ++i) {
            const int gi = tv + i * CONV_THREADS;
            if (gi < NG) *reinterpret_cast<u32x4 *>(g_tile + (size_t)(gi / GPG) * PG + (gi % GPG) * 8) = rg[i];
        }
    };

    // XCD-contiguous tile walk (workgroups are dealt to the 8 XCDs round-robin by their linear id; the
    // launcher makes gridDim.x a multiple of 8, so that is blockIdx.x & 7): each XCD works through its
    // own eighth of the tiles and the halo rows shared by neighbouring tiles are fetched into one L2
    // only.  A block without tiles still writes its (zero) partial slice.
    int t_first, t_step, t_end;
    if ((gridDim.x & 7) == 0) {
        const int per_xcd = (total_tiles + 7) / 8;
        const int xcd = blockIdx.x & 7;
        t_first = xcd * per_xcd + ((int)blockIdx.x >> 3);
        t_step = (int)gridDim.x >> 3;
        t_end = min(total_tiles, (xcd + 1) * per_xcd);
    } else {
        t_first = blockIdx.x; t_step = gridDim.x; t_end = total_tiles;
    }
    // (the walk's (image, tile row, tile column) by scalar increments, as in conv_bwd_fused_kernel: no integer division per tile)
    const int adv_tx = t_step % tiles_x, adv_ty = (t_step / tiles_x) % tiles_y, adv_b = t_step / tpi;
    int ld_b = t_first / tpi, ld_ty = (t_first % tpi) / tiles_x, ld_tx = t_first % tiles_x;
    if (t_first < t_end) load_tile(ld_b, ld_ty, ld_tx);
    for (int t = t_first; t < t_end; t += t_step) {
        __syncthreads();
        store_tile();
        __syncthreads();
        {
            if (t + t_step < t_end) {      // (past the last tile the current one is loaded again: the loads stay unconditional)
                ld_tx += adv_tx;
                const int cx = ld_tx >= tiles_x ? 1 : 0;
                ld_tx -= cx ? tiles_x : 0;
                ld_ty += adv_ty + cx;
                const int cy = ld_ty >= tiles_y ? 1 : 0;
                ld_ty -= cy ? tiles_y : 0;
                ld_b += adv_b + cy;
            }
            load_tile(ld_b, ld_ty, ld_tx);
        }
        // fragments of tile row r+1 are fetched from LDS while the matrix cores consume row r; the
        // rows are unrolled so that every read carries its row offset as an immediate
        i32x2 fa[2][2 * MT], fb[2][2 * NTW];
#define WG_ISSUE(R, BUF)                                                                                 \
    {                                                                                                    \
        _Pragma("unroll") for (int m = 0; m < MT; ++m) {                                                 \
            fa[BUF][2 * m] = (m == 0) ? tr_read_o<((R) * TW * PG) * 2>(g_ad)                             \
                           : (m == 1) ? tr_read_o<((R) * TW * PG + 16) * 2>(g_ad)                        \
                           : (m == 2) ? tr_read_o<((R) * TW * PG + 32) * 2>(g_ad)                        \
                                      : tr_read_o<((R) * TW * PG + 48) * 2>(g_ad);                       \
            fa[BUF][2 * m + 1] = (m == 0) ? tr_read_o<((R) * TW * PG + 16 * PG) * 2>(g_ad)                \
                               : (m == 1) ? tr_read_o<((R) * TW * PG + 16 + 16 * PG) * 2>(g_ad)           \
                               : (m == 2) ? tr_read_o<((R) * TW * PG + 32 + 16 * PG) * 2>(g_ad)           \
                                          : tr_read_o<((R) * TW * PG + 48 + 16 * PG) * 2>(g_ad);          \
        }                                                                                                \
        _Pragma("unroll") for (int n = 0; n < NTW; ++n) {                                                \
            fb[BUF][2 * n] = tr_read_o<((R) * WT * PK) * 2>(b_ad[n]);                                    \
            fb[BUF][2 * n + 1] = tr_read_o<((R) * WT * PK + 16 * PK) * 2>(b_ad[n]);                       \
        }                                                                                                \
    }
#define WG_CONSUME(BUF)                                                                                  \
    {                                                                                                    \
        _Pragma("unroll") for (int m = 0; m < MT; ++m) {                                                 \
            const bf16x8 af = frag_from(fa[BUF][2 * m], fa[BUF][2 * m + 1]);                             \
            /* unconditional (a padding n-tile or a bias sum nobody flushes costs an MFMA; a branch  */ \
            /* here makes hipcc shuttle every accumulator between AGPRs and VGPRs)                    */ \
            _Pragma("unroll") for (int n = 0; n < NTW; ++n)                                              \
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, frag_from(fb[BUF][2 * n], fb[BUF][2 * n + 1]), acc[m][n], 0, 0, 0); \
            accb[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, ones, accb[m], 0, 0, 0);               \
        }                                                                                                \
    }
#define WG_STEP(R)                                                                                       \
    {                                                                                                    \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                               \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        if ((R) + 1 < TH) WG_ISSUE(((R) + 1 < TH ? (R) + 1 : 0), ((R) + 1) & 1);                         \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        WG_CONSUME((R) & 1);                                                                             \
    }
        WG_ISSUE(0, 0);
        WG_STEP(0) WG_STEP(1) WG_STEP(2) WG_STEP(3) WG_STEP(4) WG_STEP(5) WG_STEP(6) WG_STEP(7)
        static_assert(TH == 8, "WG_STEP expansion above covers 8 tile rows");
#undef WG_STEP
#undef WG_CONSUME
#undef WG_ISSUE
    }

    // ---- flush: the block's partial slice (9*COUT*CIN weight sums + COUT bias sums), or atomics
    const size_t pstride = (size_t)9 * a.COUT * a.CIN + a.COUT;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int co = group * CM + m * 16 + g4 * 4 + rr;
            if (co >= a.COUT) continue;
#pragma unroll
            for (int n = 0; n < NTW; ++n) {
                const int nt = wv + 4 * n;
                if (nt >= NTT) continue;
                int tap, ci;
                if (CK >= 16) {
                    tap = nt / (CK / 16);
                    ci = chunk * CK + (nt % (CK / 16)) * 16 + i16;
                } else {
                    tap = 2 * nt + (i16 >> 3);
                    ci = i16 & 7;
                }
                if (tap < 9) {          // this block's own slice, plain stores (see the 8-wave kernel)
                    float *d = a.partials + (size_t)blockIdx.x * pstride + ((size_t)tap * a.COUT + co) * a.CIN + ci;
                    *d = a.acc_partials ? *d + acc[m][n][rr] : acc[m][n][rr];
                }
            }
            if (wv == 0 && chunk == 0 && i16 == 0) {      // bias partial: last COUT floats of the slice
                float *d = a.partials + (size_t)blockIdx.x * pstride + (size_t)9 * a.COUT * a.CIN + co;
                *d = a.acc_partials ? *d + accb[m][rr] : accb[m][rr];
            }
        }
    }
}

// blocks along the pixel tiles (= partial slices of dW): what the 256 CUs hold at once (every extra
// block costs one more pass over its dW slice); -1 on a HIP error
template <int CK, int CM, bool G8>
int wgrad_spatial(int cout, int cin, int B, int H, int W)
{
    const size_t smem = ((size_t)(HT * WT + 8) * wg_pitch(CK) + (size_t)(TH * TW + 8) * wg_pitch(G8 ? 8 : CM)) * sizeof(bf16);
    static int per_cu[64] = {};      // resident blocks per CU (registers / LDS), per device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    if (per_cu[dev & 63] == 0) {
        if (smem > 64 * 1024 &&
            hipFuncSetAttribute((const void *)conv3x3_wgrad_kernel<CK, CM, G8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return -1;
        int nblk = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, conv3x3_wgrad_kernel<CK, CM, G8>, CONV_THREADS, smem) != hipSuccess)
            return -1;
        // (the 8/16-channel shapes share their slices -- one per block -- with the fused backward kernels, which hold 4 blocks
        // per CU: with 6-8 slices per CU those run a second, half-empty round of blocks, 3-8 % slower)
        const int cap = (CK <= 16 && CM == 16) ? 4 : 8;
        per_cu[dev & 63] = nblk < 1 ? 1 : (nblk > cap ? cap : nblk);
    }
    const int tiles = ((W + TW - 1) / TW) * ((H + TH - 1) / TH) * B;
    const int chunks = cin / CK, groups = (cout + CM - 1) / CM;
    int spatial = (256 * per_cu[dev & 63]) / (chunks * groups);
    spatial = spatial < 1 ? 1 : (spatial > tiles ? tiles : spatial);
    return spatial >= 8 ? (spatial & ~7) : spatial;      // multiple of 8: XCD-contiguous tile walk
}

template <int CK, int CM, bool G8>
int launch_wgrad(const WgradArgs &a, hipStream_t st)
{
    const size_t smem = ((size_t)(HT * WT + 8) * wg_pitch(CK) + (size_t)(TH * TW + 8) * wg_pitch(G8 ? 8 : CM)) * sizeof(bf16);
    const int spatial = wgrad_spatial<CK, CM, G8>(a.COUT, a.CIN, a.B, a.H, a.W);
    if (spatial < 1) {
        mmk::set_error("mmk_conv3x3_wgrad: occupancy query failed");
        return MMK_ERR_HIP;
    }
    const int chunks = a.CIN / CK, groups = (a.COUT + CM - 1) / CM;
    hipLaunchKernelGGL((conv3x3_wgrad_kernel<CK, CM, G8>), dim3(spatial, chunks, groups), dim3(CONV_THREADS), smem, st, a);
    MMK_LAUNCH_CHECK();
    return MMK_OK;
}


// ------------------------------------------------------------------------------------------
// Backward pass of a thin layer in ONE kernel, for the layers whose data gradient takes the layer's own input as ReLU
// source.  The data gradient (input: the output gradient g; epilogue operand: a stored activation x as ReLU source) and the
// weight gradient (inputs: x and g) then read the same two tensors; as two kernels, on two streams, each of them streams
// both from HBM, and at 640 x 640 the pair takes as long side by side as one after the other (both are bandwidth-bound).
// Here a block stages the halo tiles of x and of g once: the weight-gradient contraction reads x with its halo and the
// interior of g (conv3x3_wgrad_kernel: same fragment reads, same row order, same persistent tile walk and partial slices),
// the data gradient reads g with its halo and takes the ReLU source from the interior of the x tile in LDS
// (conv3x3_ring_kernel: same k-steps, same epilogue arithmetic) -- 630 MB per launch for 8 -> 8 channels at B = 32,
// 640 x 640, instead of 1 050, and results bit-identical to the two-kernel path (tests/test_gpu_unet_kernels.py::
// test_bwd*_fused_*).  CX = channels of x = output channels of the data gradient = input channels of the weight gradient;
// CG = channels of g.  Instantiated for the four pairs of the reference network's 640 x 640 / 320 x 320 levels:
//   <8, 8>, <16, 16>  second convolution of a block (wgrad<CX, 16, CX == 8> + ring<CX, 16, ., true, false, CX == 8>)
//   <8, 16>           first convolution of encoder block 1: the data gradient also ADDS to what dx holds (the skip gradient)
//   <16, 8>           first convolution of the last decoder block: input concat(x | x2) with the data gradient's halves
//                     dx | dx2 masked by x | x2 (second application), or one 16-channel input and one unmasked output
//                     (first application: the input is the up-sampled tensor, which is no ReLU source)
struct BwdFusedArgs {
    const bf16 *x, *x2;      // (B,H,W,CX), or two (B,H,W,8) halves when x2 is set
    const bf16 *g;           // (B,H,W,CG)
    const bf16 *wpack_t;     // the data-gradient operator's packed weights (mmk_conv3x3_pack_weights, transposed = 1)
    bf16 *dx, *dx2;          // (B,H,W,CX), or two halves
    float scale;
    int masked;              // dx = ((x > 0) ? scale : 0) * conv_T(g); 0: dx = conv_T(g)
    int accumulate_dx;       // dx += ... (CX = 8)
    int B, H, W;
    float *partials;         // [gridDim.x][9*CG*CX + CG] partial sums of the weight / bias gradient
    int acc_partials;
};

template <int CX, int CG>
__global__ __launch_bounds__(CONV_THREADS) void conv_bwd_fused_kernel(const BwdFusedArgs a)
{
    static_assert((CX == 8 || CX == 16) && (CG == 8 || CG == 16), "thin layers");
    constexpr int NTT = CX == 8 ? 5 : 9;          // weight gradient: 16-column tiles of (tap, ci)
    constexpr int NTW = (NTT + 3) / 4;            // ... per wave
    constexpr int NS = CG == 8 ? 3 : 5;           // data gradient: k-steps (4 taps x 8 channels | 2 taps x 16 channels)
    constexpr int NT = 4;
    constexpr int GX = CX / 8, GG = CG / 8;       // 16-byte granules per pixel
    __shared__ __attribute__((aligned(16))) bf16 x_tile[(HT * WT + 8) * CX];
    __shared__ __attribute__((aligned(16))) bf16 g_tile[(HT * WT + 8) * CG];
    __shared__ __attribute__((aligned(16))) bf16 w_lds[NS * 64 * 8];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
    const int tpi = tiles_x * tiles_y;
    const int total_tiles = tpi * a.B;
    const int i16 = lane & 15, g4 = lane >> 4, q = i16 >> 2, pp = i16 & 3;
    const bool split = a.x2 != nullptr;

    // ---- weight-gradient side (conv3x3_wgrad_kernel<CX, 16, CG == 8>)
    f32x4 acc[NTW], accb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int n = 0; n < NTW; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (bf16)1.0f;
    unsigned b_ad[NTW];
#pragma unroll
    for (int n = 0; n < NTW; ++n) {
        const int nt = wv + 4 * n;
        int tap = CX == 8 ? 2 * nt + (pp >> 1) : nt;
        const int col = CX == 8 ? 4 * (pp & 1) : 4 * pp;
        tap = tap > 8 ? 8 : tap;
        const int ty = tap / 3, tx = tap % 3;
        b_ad[n] = lds_addr(x_tile) + (unsigned)((((ty * WT) + 4 * g4 + q + tx) * CX + col) * 2);
    }
    // (the g tile carries its halo here: the interior starts one row and one column in)
    const unsigned g_ad = lds_addr(g_tile) + (unsigned)(((WT + 1 + 4 * g4 + q) * CG + 4 * pp) * 2);

    // ---- data-gradient side (conv3x3_ring_kernel<CG, 16, ., true, false, CX == 8>)
    int d_lane[NS];
#pragma unroll
    for (int ks = 0; ks < NS; ++ks) {
        int tap = CG == 8 ? 4 * ks + (lane >> 4) : 2 * ks + (lane >> 5);
        const int ch = CG == 8 ? 0 : 8 * ((lane >> 4) & 1);
        tap = tap > 8 ? 8 : tap;
        d_lane[ks] = ((2 * wv + tap / 3) * WT + (lane & 15) + tap % 3) * CG + ch;
    }
    // CX = 8: after the lane exchange a lane holds the 8 channels of pixel (lrow8, lcol8); CX = 16: the lane's 4 channels
    // of pixel (2 wv + n/2, 16 (n&1) + lane%16) per n-tile
    const int n8 = lane >> 4;
    const int lrow8 = 2 * wv + (n8 >> 1), lcol8 = (n8 & 1) * 16 + (lane & 15);
    const int lpix8 = lrow8 * a.W + lcol8;
    const int s_off8 = ((lrow8 + 1) * WT + lcol8 + 1) * CX;      // the lane's pixel in the x tile (ReLU source)
    int o_pix[NT], s_off[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int r = 2 * wv + (n >> 1), c = (n & 1) * 16 + (lane & 15);
        o_pix[n] = r * a.W + c;
        s_off[n] = ((r + 1) * WT + c + 1) * CX + (lane >> 4) * 4;
    }
    bf16 *const o_base = (a.dx2 != nullptr) ? ((lane >> 5) ? a.dx2 : a.dx) + ((lane >> 4) & 1) * 4 : a.dx + (lane >> 4) * 4;
    const int o_stride = (a.dx2 != nullptr) ? 8 : 16;
    for (int i = tid; i < NS * 64; i += CONV_THREADS) reinterpret_cast<u32x4 *>(w_lds)[i] = reinterpret_cast<const u32x4 *>(a.wpack_t)[i];

    // register prefetch of the next tile's operands while the current one is consumed; every load is unconditional
    constexpr int NX = HT * WT * GX, NG = HT * WT * GG;
    constexpr int RX = (NX + CONV_THREADS - 1) / CONV_THREADS, RG = (NG + CONV_THREADS - 1) / CONV_THREADS;
    u32x4 rx[RX], rg[RG];
    auto load_tile = [&](int b, int tyi, int txi) {
        int tv = tid;
        asm volatile("" : "+v"(tv));
        const int tx0 = txi * TW, ty0 = tyi * TH;
        const int pix0 = (b * a.H + ty0) * a.W + tx0;
#pragma unroll
        for (int i = 0; i < RX; ++i) {
            int gi = tv + i * CONV_THREADS;
            gi = gi < NX ? gi : NX - 1;
            const int pix = gi / GX, half = gi % GX;
            const int dy = pix / WT - 1, dx = pix % WT - 1;
            const bool ok = (unsigned)(ty0 + dy) < (unsigned)a.H && (unsigned)(tx0 + dx) < (unsigned)a.W;
            const long po = pix0 + dy * a.W + dx;
            const bf16 *src = split ? (half ? a.x2 : a.x) + po * 8 : a.x + po * CX + half * 8;
            rx[i] = *(ok ? reinterpret_cast<const u32x4 *>(src) : &g_zero16);
        }
#pragma unroll
        for (int i = 0; i < RG; ++i) {
            int gi = tv + i * CONV_THREADS;
            gi = gi < NG ? gi : NG - 1;
            const int pix = gi / GG, gc = gi % GG;
            const int dy = pix / WT - 1, dx = pix % WT - 1;
            const bool ok = (unsigned)(ty0 + dy) < (unsigned)a.H && (unsigned)(tx0 + dx) < (unsigned)a.W;
            rg[i] = *(ok ? reinterpret_cast<const u32x4 *>(a.g + (long)(pix0 + dy * a.W + dx) * CG + gc * 8) : &g_zero16);
        }
    };
    auto store_tile = [&]() {          // (pitch = channel count: granule gi of a tile sits at element 8 gi)
        int tv = tid;
        asm volatile("" : "+v"(tv));
#pragma unroll
        for (int i = 0; i < RX; ++i) {
            const int gi = tv + i * CONV_THREADS;
            if (gi < NX) *reinterpret_cast<u32x4 *>(x_tile + (size_t)gi * 8) = rx[i];
        }
#pragma unroll
        for (int i = 0; i < RG; ++i) {
            const int gi = tv + i * CONV_THREADS;
            if (gi < NG) *reinterpret_cast<u32x4 *>(g_tile + (size_t)gi * 8) = rg[i];
        }
    };

    // XCD-contiguous persistent tile walk, as in conv3x3_wgrad_kernel (one partial slice per block)
    int t_first, t_step, t_end;
    if ((gridDim.x & 7) == 0) {
        const int per_xcd = (total_tiles + 7) / 8;
        const int xcd = blockIdx.x & 7;
        t_first = xcd * per_xcd + ((int)blockIdx.x >> 3);
        t_step = (int)gridDim.x >> 3;
        t_end = min(total_tiles, (xcd + 1) * per_xcd);
    } else {
        t_first = blockIdx.x; t_step = gridDim.x; t_end = total_tiles;
    }
    // (image, tile row, tile column) of the walk's tiles by increments on the scalar unit instead of four integer divisions per tile
    const int adv_tx = t_step % tiles_x, adv_ty = (t_step / tiles_x) % tiles_y, adv_b = t_step / tpi;
    int ld_b = t_first / tpi, ld_ty = (t_first % tpi) / tiles_x, ld_tx = t_first % tiles_x;
    int cb = ld_b, cty = ld_ty, ctx = ld_tx;
    // The first tile goes into LDS in front of the loop, every later one at the loop's END (round 5).  With store_tile() at the
    // loop's head the head merged the prologue's outstanding loads with the back edge's outstanding dx STORES; hipcc covered the
    // former with s_waitcnt vmcnt(5) ... vmcnt(0) in front of the ds_writes, which on the back edge made every wave wait for its own
    // output stores right behind issuing them (loads and stores share the counter).  Now no load is outstanding at the head on
    // either edge; the stores are waited for one tile later, by the delivery of the next prefetch.
    if (t_first < t_end) {
        load_tile(ld_b, ld_ty, ld_tx);
        store_tile();
    }
    __syncthreads();
    for (int t = t_first; t < t_end; t += t_step) {
        const int b = cb;
        const int tx0 = ctx * TW, ty0 = cty * TH;
        const int pix0 = (b * a.H + ty0) * a.W + tx0;
        const bool ok8 = (ty0 + lrow8) < a.H && (tx0 + lcol8) < a.W;
        // the accumulate target of this tile's epilogue, ahead of the next tile's operands in the load queue
        u32x4 e8_acc = {0u, 0u, 0u, 0u};
        if constexpr (CX == 8) e8_acc = *((ok8 && a.accumulate_dx) ? reinterpret_cast<const u32x4 *>(a.dx + (long)(pix0 + lpix8) * 8) : &g_zero16);
        {
            if (t + t_step < t_end) {      // the next tile of the walk (past the last one the current tile is loaded again)
                ld_tx += adv_tx;
                const int cx = ld_tx >= tiles_x ? 1 : 0;
                ld_tx -= cx ? tiles_x : 0;
                ld_ty += adv_ty + cx;
                const int cy = ld_ty >= tiles_y ? 1 : 0;
                ld_ty -= cy ? tiles_y : 0;
                ld_b += adv_b + cy;
            }
            load_tile(ld_b, ld_ty, ld_tx);
            cb = ld_b; cty = ld_ty; ctx = ld_tx;
        }
        // ---- weight gradient: tile rows 0..7, the fragments of row r+1 in flight while row r is consumed
        i32x2 fa[2][2], fb[2][2 * NTW];
#define BF_ISSUE(R, BUF)                                                                     \
    {                                                                                        \
        fa[BUF][0] = tr_read_o<((R) * WT * CG) * 2>(g_ad);                                   \
        fa[BUF][1] = tr_read_o<((R) * WT * CG + 16 * CG) * 2>(g_ad);                         \
        _Pragma("unroll") for (int n = 0; n < NTW; ++n) {                                    \
            fb[BUF][2 * n] = tr_read_o<((R) * WT * CX) * 2>(b_ad[n]);                        \
            fb[BUF][2 * n + 1] = tr_read_o<((R) * WT * CX + 16 * CX) * 2>(b_ad[n]);          \
        }                                                                                    \
    }
#define BF_CONSUME(BUF)                                                                      \
    {                                                                                        \
        const bf16x8 af = frag_from(fa[BUF][0], fa[BUF][1]);                                 \
        _Pragma("unroll") for (int n = 0; n < NTW; ++n)                                      \
            acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, frag_from(fb[BUF][2 * n], fb[BUF][2 * n + 1]), acc[n], 0, 0, 0); \
        accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, ones, accb, 0, 0, 0);             \
    }
#define BF_STEP(R)                                                                           \
    {                                                                                        \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                   \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        if ((R) + 1 < TH) BF_ISSUE(((R) + 1 < TH ? (R) + 1 : 0), ((R) + 1) & 1);             \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        BF_CONSUME((R) & 1);                                                                 \
    }
        BF_ISSUE(0, 0);
        BF_STEP(0) BF_STEP(1) BF_STEP(2) BF_STEP(3) BF_STEP(4) BF_STEP(5) BF_STEP(6) BF_STEP(7)
        static_assert(TH == 8, "BF_STEP expansion above covers 8 tile rows");
#undef BF_STEP
#undef BF_CONSUME
#undef BF_ISSUE
        // ---- data gradient of the tile
        f32x4 dacc[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) dacc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NS; ++ks) {
            const bf16 *bl = g_tile + d_lane[ks];
            bf16x8 bf[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) bf[n] = *reinterpret_cast<const bf16x8 *>(bl + ((n >> 1) * WT + (n & 1) * 16) * CG);
            const bf16x8 af = *reinterpret_cast<const bf16x8 *>(w_lds + ((size_t)(ks * 64 + lane)) * 8);
#pragma unroll
            for (int n = 0; n < NT; ++n) dacc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf[n], dacc[n], 0, 0, 0);
        }
        u32x4 e8_src = {0u, 0u, 0u, 0u};
        bf16x4 e_src[NT];
        if constexpr (CX == 8) {
            e8_src = *reinterpret_cast<const u32x4 *>(x_tile + s_off8);
        } else {
#pragma unroll
            for (int n = 0; n < NT; ++n) e_src[n] = *reinterpret_cast<const bf16x4 *>(x_tile + s_off[n]);
        }
        // the next tile's operands have had the MFMA work to arrive: take delivery in front of this tile's stores (loads and
        // stores share vmcnt and complete out of order: a wait for them behind the stores is a wait for the stores to drain)
#pragma unroll
        for (int i = 0; i < RX; ++i) asm volatile("" : "+v"(rx[i]));
#pragma unroll
        for (int i = 0; i < RG; ++i) asm volatile("" : "+v"(rg[i]));
        // (the additions of zero below are the ring kernel's bias and accumulate-target additions: they turn -0 into +0)
        if constexpr (CX == 8) {
            float v8[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const auto s01 = __builtin_amdgcn_permlane16_swap(__float_as_uint(dacc[0][r]), __float_as_uint(dacc[1][r]), false, false);
                const auto s23 = __builtin_amdgcn_permlane16_swap(__float_as_uint(dacc[2][r]), __float_as_uint(dacc[3][r]), false, false);
                const auto lo = __builtin_amdgcn_permlane32_swap(s01[0], s23[0], false, false);
                const auto hi = __builtin_amdgcn_permlane32_swap(s01[1], s23[1], false, false);
                v8[r] = __uint_as_float(lo[0]);
                v8[4 + r] = __uint_as_float(hi[0]);
            }
            float sv[8], av[8];
            unpack8(e8_src, sv);
            unpack8(e8_acc, av);
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float v = fmaxf(v8[r] + 0.f, -INFINITY);
                const float masked = (sv[r] > 0.f) ? v * a.scale : 0.f;
                v8[r] = masked + av[r];
            }
            u32x4 *dst8 = ok8 ? reinterpret_cast<u32x4 *>(a.dx + (long)(pix0 + lpix8) * 8) : reinterpret_cast<u32x4 *>(g_sink16);
            *dst8 = pack8(v8);
        } else {
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const bool okp = (ty0 + 2 * wv + (n >> 1)) < a.H && (tx0 + (n & 1) * 16 + (lane & 15)) < a.W;
                bf16x4 outv;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = fmaxf(dacc[n][r] + 0.f, -INFINITY);
                    const float masked = ((float)e_src[n][r] > 0.f) ? v * a.scale : 0.f;
                    outv[r] = (bf16)(a.masked ? masked + 0.f : v);
                }
                bf16x4 *dst = okp ? reinterpret_cast<bf16x4 *>(o_base + (long)(pix0 + o_pix[n]) * o_stride) : reinterpret_cast<bf16x4 *>(g_sink16);
                *dst = outv;
            }
        }
        if (t + t_step >= t_end) break;
        __syncthreads();          // every wave is done with this tile's LDS image
        store_tile();
        __syncthreads();
    }

    // ---- flush the block's partial slice: [tap][co < CG][ci < CX] weight sums, then CG bias sums
    constexpr size_t pstride = (size_t)9 * CG * CX + CG;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int co = g4 * 4 + rr;
        if (co >= CG) continue;
#pragma unroll
        for (int n = 0; n < NTW; ++n) {
            const int nt = wv + 4 * n;
            if (nt >= NTT) continue;
            const int tap = CX == 8 ? 2 * nt + (i16 >> 3) : nt;
            const int ci = CX == 8 ? (i16 & 7) : i16;
            if (tap < 9) {
                float *d = a.partials + (size_t)blockIdx.x * pstride + ((size_t)tap * CG + co) * CX + ci;
                *d = a.acc_partials ? *d + acc[n][rr] : acc[n][rr];
            }
        }
        if (wv == 0 && i16 == 0) {
            float *d = a.partials + (size_t)blockIdx.x * pstride + (size_t)9 * CG * CX + co;
            *d = a.acc_partials ? *d + accb[rr] : accb[rr];
        }
    }
}

// ------------------------------------------------------------------------------------------
// Weight gradient of the layers with >= 64 channels on both sides: 8 waves per block (two per
// SIMD) on a 64 (co) x 64 (ci) x 9 (taps) slice.  Wave (wm, wc) owns 2 co-tiles x 1 ci-tile
// x 9 taps.  The contraction walks the halo rows rho of a tile: the three x fragments of row
// rho (tx = 0..2) meet the g fragments of the tile rows rho, rho-1, rho-2 (ty = 0, 1, 2), so
// every x fragment is read from LDS once instead of three times (10 transposing reads per 18
// MFMAs).  Fragment reads of row rho+1 are in flight while row rho is on the matrix cores.
constexpr int WGD_THREADS = 512;

__global__ __launch_bounds__(WGD_THREADS) void conv3x3_wgrad_deep_kernel(const WgradArgs a)
{
    constexpr int CK = 64, CM = 64, PK = wg_pitch(64), PG = wg_pitch(64);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16 *x_tile = reinterpret_cast<bf16 *>(smem);                 // (HT*WT + 8) * PK
    bf16 *g_tile = x_tile + (HT * WT + 8) * PK;                    // (TH*TW + 8) * PG

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: branches on it are scalar
    const int wm = wv >> 2, wc = wv & 3;
    const int chunk = blockIdx.y, group = blockIdx.z;
    const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
    const int tpi = tiles_x * tiles_y;
    const int total_tiles = tpi * a.B;
    const int i16 = lane & 15, g4 = lane >> 4, q = i16 >> 2, pp = i16 & 3;

    f32x4 acc[3][3][2];
    f32x4 accb[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        accb[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ty = 0; ty < 3; ++ty)
#pragma unroll
            for (int tx = 0; tx < 3; ++tx) acc[ty][tx][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (bf16)1.0f;

    const unsigned xb0 = lds_addr(x_tile) + (unsigned)(((4 * g4 + q) * PK + wc * 16 + 4 * pp) * 2);
    const unsigned gb0 = lds_addr(g_tile) + (unsigned)(((4 * g4 + q) * PG + wm * 32 + 4 * pp) * 2);

    // register staging of the next tile (same scheme as conv3x3_wgrad_kernel)
    constexpr int GPP = CK / 8;
    constexpr int NIN = HT * WT * GPP;
    constexpr int RIN = (NIN + WGD_THREADS - 1) / WGD_THREADS;
    constexpr int GPG = CM / 8;
    constexpr int NG = TH * TW * GPG;
    constexpr int RG = (NG + WGD_THREADS - 1) / WGD_THREADS;
    u32x4 rin[RIN], rg[RG];
    // `tv` is the thread index behind an opaque asm: everything derived from it is recomputed per tile
    // (a few VALU instructions) instead of being hoisted out of the tile loop into ~50 live registers
    auto load_tile = [&](int b, int tyi, int txi) {
        int tv = tid;
        asm volatile("" : "+v"(tv));
        const int tx0 = txi * TW, ty0 = tyi * TH;
        const int c0 = chunk * CK;
        const bool in1 = c0 < a.C1;                     // a 64-channel chunk lies in one of the two inputs
        const bf16 *xb = in1 ? a.x1 : a.x2;
        const int xc = in1 ? a.C1 : a.C2, cb = in1 ? c0 : c0 - a.C1;
#pragma unroll
        for (int i = 0; i < RIN; ++i) {
            int gi = tv + i * WGD_THREADS;
            gi = gi < NIN ? gi : NIN - 1;
            const int pix = gi / GPP, gc = gi % GPP;
            const int yy = ty0 + pix / WT - 1, xx = tx0 + pix % WT - 1;
            const bool ok = yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
            const size_t p = ((size_t)b * a.H + yy) * a.W + xx;
            const u32x4 *sp = ok ? reinterpret_cast<const u32x4 *>(xb + p * xc + cb + gc * 8) : &g_zero16;
            rin[i] = *sp;
        }
#pragma unroll
        for (int i = 0; i < RG; ++i) {
            int gi = tv + i * WGD_THREADS;
            gi = gi < NG ? gi : NG - 1;
            const int pix = gi / GPG, gc = gi % GPG;
            const int yy = ty0 + pix / TW, xx = tx0 + pix % TW;
            const bool ok = yy < a.H && xx < a.W;
            const size_t p = ((size_t)b * a.H + yy) * a.W + xx;
            const u32x4 *sp = ok ? reinterpret_cast<const u32x4 *>(a.g + p * a.COUT + group * CM + gc * 8) : &g_zero16;
            rg[i] = *sp;
        }
    };
    auto store_tile = [&]() {
        int tv = tid;
        asm volatile("" : "+v"(tv));
#pragma unroll
        for (int i = 0; i < RIN; ++i) {
            const int gi = tv + i * WGD_THREADS;
            if (gi < NIN) *reinterpret_cast<u32x4 *>(x_tile + (size_t)(gi / GPP) * PK + (gi % GPP) * 8) = rin[i];
        }
#pragma unroll
        for (int i = 0; i < RG; ++i) {
            const int gi = tv + i * WGD_THREADS;
            if (gi < NG) *reinterpret_cast<u32x4 *>(g_tile + (size_t)(gi / GPG) * PG + (gi % GPG) * 8) = rg[i];
        }
    };

    // XCD-contiguous tile walk (see conv3x3_wgrad_kernel)
    int t_first, t_step, t_end;
    if ((gridDim.x & 7) == 0) {
        const int per_xcd = (total_tiles + 7) / 8;
        const int xcd = blockIdx.x & 7;
        t_first = xcd * per_xcd + ((int)blockIdx.x >> 3);
        t_step = (int)gridDim.x >> 3;
        t_end = min(total_tiles, (xcd + 1) * per_xcd);
    } else {
        t_first = blockIdx.x; t_step = gridDim.x; t_end = total_tiles;
    }
    // (the walk's (image, tile row, tile column) by scalar increments, as in conv_bwd_fused_kernel: no integer division per tile)
    const int adv_tx = t_step % tiles_x, adv_ty = (t_step / tiles_x) % tiles_y, adv_b = t_step / tpi;
    int ld_b = t_first / tpi, ld_ty = (t_first % tpi) / tiles_x, ld_tx = t_first % tiles_x;
    if (t_first < t_end) load_tile(ld_b, ld_ty, ld_tx);
    for (int t = t_first; t < t_end; t += t_step) {
        __syncthreads();
        store_tile();
        __syncthreads();
        {
            if (t + t_step < t_end) {      // (past the last tile the current one is loaded again: the loads stay unconditional)
                ld_tx += adv_tx;
                const int cx = ld_tx >= tiles_x ? 1 : 0;
                ld_tx -= cx ? tiles_x : 0;
                ld_ty += adv_ty + cx;
                const int cy = ld_ty >= tiles_y ? 1 : 0;
                ld_ty -= cy ? tiles_y : 0;
                ld_b += adv_b + cy;
            }
            load_tile(ld_b, ld_ty, ld_tx);
        }
        // double-buffered fragment registers; every read carries its offset as an immediate (~100 distinct
        // addresses would otherwise be hoisted into registers and spilled)
        i32x2 fb[2][3][2], fa[4][2][2];      // x row rho in fb[rho & 1]; g rows rho-2 .. rho+1 in a ring of 4
#define WGD_ISSUE(RHO)                                                                                   \
    {                                                                                                    \
        fb[(RHO) & 1][0][0] = tr_read_o<(((RHO) * WT + 0) * PK) * 2>(xb0);                               \
        fb[(RHO) & 1][0][1] = tr_read_o<(((RHO) * WT + 0) * PK + 16 * PK) * 2>(xb0);                     \
        fb[(RHO) & 1][1][0] = tr_read_o<(((RHO) * WT + 1) * PK) * 2>(xb0);                               \
        fb[(RHO) & 1][1][1] = tr_read_o<(((RHO) * WT + 1) * PK + 16 * PK) * 2>(xb0);                     \
        fb[(RHO) & 1][2][0] = tr_read_o<(((RHO) * WT + 2) * PK) * 2>(xb0);                               \
        fb[(RHO) & 1][2][1] = tr_read_o<(((RHO) * WT + 2) * PK + 16 * PK) * 2>(xb0);                     \
        if ((RHO) < TH) {                                                                                \
            constexpr int R_ = (RHO) < TH ? (RHO) : 0;                                                   \
            fa[R_ & 3][0][0] = tr_read_o<((R_ * TW) * PG) * 2>(gb0);                                     \
            fa[R_ & 3][0][1] = tr_read_o<((R_ * TW) * PG + 16 * PG) * 2>(gb0);                           \
            fa[R_ & 3][1][0] = tr_read_o<((R_ * TW) * PG + 16) * 2>(gb0);                                \
            fa[R_ & 3][1][1] = tr_read_o<((R_ * TW) * PG + 16 + 16 * PG) * 2>(gb0);                      \
        }                                                                                                \
    }
#define WGD_STEP(RHO)                                                                                    \
    {                                                                                                    \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                               \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        /* row rho+1 is fetched while row rho is on the matrix cores */                                  \
        if ((RHO) + 1 < HT) WGD_ISSUE(((RHO) + 1 < HT ? (RHO) + 1 : 0));                                 \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        bf16x8 bfr[3];                                                                                   \
        _Pragma("unroll") for (int tx = 0; tx < 3; ++tx) bfr[tx] = frag_from(fb[(RHO) & 1][tx][0], fb[(RHO) & 1][tx][1]); \
        _Pragma("unroll") for (int ty = 0; ty < 3; ++ty) {                                               \
            const int r = (RHO) - ty; /* tile row whose g meets x row rho at tap row ty */               \
            if (r >= 0 && r < TH) {                                                                      \
                _Pragma("unroll") for (int m = 0; m < 2; ++m) {                                          \
                    const bf16x8 af = frag_from(fa[r & 3][m][0], fa[r & 3][m][1]);                       \
                    _Pragma("unroll") for (int tx = 0; tx < 3; ++tx)                                     \
                        acc[ty][tx][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr[tx], acc[ty][tx][m], 0, 0, 0); \
                    if (ty == 0) accb[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, ones, accb[m], 0, 0, 0); /* every wave: no branch */ \
                }                                                                                        \
            }                                                                                            \
        }                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                               \
    }
        WGD_ISSUE(0);
        WGD_STEP(0) WGD_STEP(1) WGD_STEP(2) WGD_STEP(3) WGD_STEP(4)
        WGD_STEP(5) WGD_STEP(6) WGD_STEP(7) WGD_STEP(8) WGD_STEP(9)
        static_assert(HT == 10, "WGD_STEP expansion above covers 10 halo rows");
#undef WGD_STEP
#undef WGD_ISSUE
    }

    // ---- flush: the block's partial slice (9*COUT*CIN weight sums + COUT bias sums), or atomics
    const size_t pstride = (size_t)9 * a.COUT * a.CIN + a.COUT;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int co = group * CM + (wm * 2 + m) * 16 + g4 * 4 + rr;
            const int ci = chunk * CK + wc * 16 + i16;
            {
                // plain stores into this block's own slice (summed by the unpack kernel): float atomics
                // run at ~1.3 TB/s and 256 blocks x 147 KB of them cost 30-35 us per launch -- and make the sums depend on
                // the order the blocks arrive in
                float *ps = a.partials + (size_t)blockIdx.x * pstride + (size_t)co * a.CIN + ci;
#pragma unroll
                for (int ty = 0; ty < 3; ++ty)
#pragma unroll
                    for (int tx = 0; tx < 3; ++tx) {
                        float *d = ps + (size_t)(ty * 3 + tx) * a.COUT * a.CIN;
                        *d = a.acc_partials ? *d + acc[ty][tx][m][rr] : acc[ty][tx][m][rr];
                    }
            }
            if (wc == 0 && chunk == 0 && i16 == 0) {      // bias partial: last COUT floats of the slice
                float *d = a.partials + (size_t)blockIdx.x * pstride + (size_t)9 * a.COUT * a.CIN + co;
                *d = a.acc_partials ? *d + accb[m][rr] : accb[m][rr];
            }
        }
    }
}

// blocks along the pixel tiles (= partial slices of dW) of the 8-wave weight-gradient kernel
int wgrad_deep_slices(int cout, int cin, int B, int H, int W)
{
    const int tiles = ((W + TW - 1) / TW) * ((H + TH - 1) / TH) * B;
    const int chunks = cin / 64, groups = cout / 64;
    int spatial = 256 / (chunks * groups);            // one 8-wave block per CU
    spatial = spatial < 1 ? 1 : (spatial > tiles ? tiles : spatial);
    return spatial >= 8 ? (spatial & ~7) : spatial;   // multiple of 8: XCD-contiguous tile walk
}

bool wgrad_is_deep(int cout, int cin, int c1) { return cin % 64 == 0 && cout % 64 == 0 && c1 % 64 == 0; }

int launch_wgrad_deep(const WgradArgs &a, hipStream_t st)
{
    const size_t smem = ((size_t)(HT * WT + 8) * wg_pitch(64) + (size_t)(TH * TW + 8) * wg_pitch(64)) * sizeof(bf16);
    static bool attr_set[64] = {};
    int dev = 0;
    MMK_CHECK_HIP(hipGetDevice(&dev));
    if (!attr_set[dev & 63]) {
        MMK_CHECK_HIP(hipFuncSetAttribute((const void *)conv3x3_wgrad_deep_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_set[dev & 63] = true;
    }
    const int spatial = wgrad_deep_slices(a.COUT, a.CIN, a.B, a.H, a.W);
    const int chunks = a.CIN / 64, groups = a.COUT / 64;
    hipLaunchKernelGGL(conv3x3_wgrad_deep_kernel, dim3(spatial, chunks, groups), dim3(WGD_THREADS), smem, st, a);
    MMK_LAUNCH_CHECK();
    return MMK_OK;
}

int dispatch_wgrad(const WgradArgs &a, hipStream_t st)
{
    if (wgrad_is_deep(a.COUT, a.CIN, a.C1)) return launch_wgrad_deep(a, st);
    const int CK = cin_chunk(a.CIN), CM = cout_group(a.COUT);
#define MMK_WG_CASE(K, M) if (CK == K && CM == M) return (M == 16 && a.COUT == 8) ? launch_wgrad<K, M, (M == 16)>(a, st) : launch_wgrad<K, M, false>(a, st)
    MMK_WG_CASE(8, 16); MMK_WG_CASE(8, 32); MMK_WG_CASE(8, 64);
    MMK_WG_CASE(16, 16); MMK_WG_CASE(16, 32); MMK_WG_CASE(16, 64);
    MMK_WG_CASE(32, 16); MMK_WG_CASE(32, 32); MMK_WG_CASE(32, 64);
    MMK_WG_CASE(64, 16); MMK_WG_CASE(64, 32); MMK_WG_CASE(64, 64);
#undef MMK_WG_CASE
    mmk::set_error("mmk_conv3x3_wgrad: unsupported channel counts CIN=%d COUT=%d", a.CIN, a.COUT);
    return MMK_ERR_ARG;
}

// number of partial slices (= blocks along the tiles) the kernel dispatch_wgrad picks will run with
int wgrad_slices(int cout, int cin, int c1, int B, int H, int W)
{
    if (wgrad_is_deep(cout, cin, c1)) return wgrad_deep_slices(cout, cin, B, H, W);
    const int CK = cin_chunk(cin), CM = cout_group(cout);
#define MMK_WG_CASE(K, M) if (CK == K && CM == M) return (M == 16 && cout == 8) ? wgrad_spatial<K, M, (M == 16)>(cout, cin, B, H, W) : wgrad_spatial<K, M, false>(cout, cin, B, H, W)
    MMK_WG_CASE(8, 16); MMK_WG_CASE(8, 32); MMK_WG_CASE(8, 64);
    MMK_WG_CASE(16, 16); MMK_WG_CASE(16, 32); MMK_WG_CASE(16, 64);
    MMK_WG_CASE(32, 16); MMK_WG_CASE(32, 32); MMK_WG_CASE(32, 64);
    MMK_WG_CASE(64, 16); MMK_WG_CASE(64, 32); MMK_WG_CASE(64, 64);
#undef MMK_WG_CASE
    return 0;
}

struct UnpackBatch {
    const float *src[PACK_BATCH_MAX];     // dWt (9,cout,cin), or per-block partial slices (slices, 9*cout*cin + cout)
    float *dW[PACK_BATCH_MAX];
    float *db[PACK_BATCH_MAX];            // bias gradient out (partial-slice form only), may be null
    int cout[PACK_BATCH_MAX], cin[PACK_BATCH_MAX], slices[PACK_BATCH_MAX];
};

// A block's 256 threads are G groups of 256 / G lanes (G = 1, 4 or 16, picked per layer by its size: unpack_groups); a block
// covers 4 * 256 / G consecutive elements of a slice, group g sums slices g, g + G, ... (sixteen float4 loads in flight per lane),
// the groups' sums are added in group order through LDS.  Round 5: large layers read whole 4 KB runs of one slice per block
// (G = 1: longer DRAM bursts, no LDS), small ones spread their many slices over 16 groups (the 8- and 16-channel layers have
// ~2 000 slices of 0.6-2.3 K elements: at four groups a launch was bound by the serial chain over their slices) -- a fixed
// association per element either way: deterministic.  Weight sums go transposed to [co][ci][tap], bias sums to db.
__host__ __device__ inline int unpack_groups(int total) { return total < 8192 ? 16 : (total < 65536 ? 4 : 1); }

__global__ __launch_bounds__(256) void unpack_wgrad_batch_kernel(const UnpackBatch ub)
{
    __shared__ float red[16][64 * 4 + 4];             // [group][element of the block] for G = 16 (64 elements) and G = 4 (256)
    const int l = blockIdx.y;
    const int COUT = ub.cout[l], CIN = ub.cin[l];
    const int nw = COUT * CIN * 9;
    const bool sliced = ub.slices[l] > 0;
    const int S = sliced ? ub.slices[l] : 1;
    const int total = sliced ? nw + COUT : nw;       // elements of one slice
    const float *__restrict__ src = ub.src[l];
    float *__restrict__ dst = ub.dW[l];
    float *__restrict__ dbo = ub.db[l];
    const bool vec = (total & 3) == 0;                // float4 path: slices stay 16-byte aligned
    const int G = unpack_groups(total), tpg = 256 / G, epb = 4 * tpg;
    const int sg = threadIdx.x / tpg, tl = threadIdx.x % tpg;
    auto put = [&](int ee, float t) {
        if (ee < nw) {
            const int ci = ee % CIN, co = (ee / CIN) % COUT, tap = ee / (CIN * COUT);
            dst[((size_t)co * CIN + ci) * 9 + tap] = t;
        } else if (ee < total && dbo) {
            dbo[ee - nw] = t;
        }
    };
    for (int e0 = blockIdx.x * epb; e0 < total; e0 += gridDim.x * epb) {      // (block-uniform trip count: the barriers are safe)
        const int e = e0 + 4 * tl;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (vec && e + 3 < total) {
            const float *p = src + e;
            int sl = sg;
            for (; sl + 15 * G < S; sl += 16 * G) {
                float4 t[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) t[u] = *reinterpret_cast<const float4 *>(p + (size_t)(sl + u * G) * total);
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    v[0] += t[u].x; v[1] += t[u].y; v[2] += t[u].z; v[3] += t[u].w;
                }
            }
            for (; sl < S; sl += G) {
                const float4 t = *reinterpret_cast<const float4 *>(p + (size_t)sl * total);
                v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
            }
        } else {
            for (int sl = sg; sl < S; sl += G)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (e + j < total) v[j] += src[(size_t)sl * total + e + j];
        }
        if (G == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) put(e + j, v[j]);
        } else {
            // (G = 4: 256 elements per block, stored as four 64-element rows per group)
            float *row = &red[0][0] + (size_t)sg * (epb + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) row[4 * tl + j] = v[j];
            __syncthreads();
            for (int k = threadIdx.x; k < epb; k += 256) {
                float t = 0.f;
                for (int g = 0; g < G; ++g) t += (&red[0][0])[(size_t)g * (epb + 4) + k];
                put(e0 + k, t);
            }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------
// First layer: fp32 NCHW input with 1..4 channels -> 8 channels NHWC bf16 (+bias, ReLU).
// 72..288 FMAs per pixel: plain VALU, HBM-bound.
// pre (optional, 2 floats per input channel: offset, 1/scale): the input is (x - offset) * (1/scale), the
// policy's per-channel min-max normalisation (icp_weight_policy.py:151-155) applied while loading; the zero
// padding is of the normalised image.
__global__ __launch_bounds__(256) void conv_first_kernel(const float *__restrict__ x, int CIN, const float *__restrict__ Wt,
                                                         const float *__restrict__ bias, const float *__restrict__ pre, int B, int H,
                                                         int W, float slope, bf16 *__restrict__ y)
{
    __shared__ float ws[8 * 4 * 9 + 8];
    for (int i = threadIdx.x; i < 8 * CIN * 9; i += blockDim.x) ws[i] = Wt[i];
    if (threadIdx.x < 8) ws[8 * 4 * 9 + threadIdx.x] = bias ? bias[threadIdx.x] : 0.f;
    __syncthreads();
    const size_t npix = (size_t)B * H * W;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (size_t)gridDim.x * blockDim.x) {
        const int xx = (int)(p % W), yy = (int)((p / W) % H), b = (int)(p / ((size_t)W * H));
        float acc[8];
#pragma unroll
        for (int co = 0; co < 8; ++co) acc[co] = ws[8 * 4 * 9 + co];
        for (int c = 0; c < CIN; ++c) {
            const float *xc = x + ((size_t)b * CIN + c) * H * W;
            const float psub = pre ? pre[2 * c] : 0.f, prcp = pre ? pre[2 * c + 1] : 1.f;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int y2 = yy + tap / 3 - 1, x2 = xx + tap % 3 - 1;
                float v = 0.f;
                if (y2 >= 0 && y2 < H && x2 >= 0 && x2 < W) v = (xc[(size_t)y2 * W + x2] - psub) * prcp;
                // bf16 operands as on the MFMA path
                v = (float)(bf16)v;
#pragma unroll
                for (int co = 0; co < 8; ++co) acc[co] += v * ws[(co * CIN + c) * 9 + tap];
            }
        }
        bf16x8 o;
#pragma unroll
        for (int co = 0; co < 8; ++co) o[co] = (bf16)((slope > 0.f) ? act_leaky(acc[co], slope) : fmaxf(acc[co], 0.f));
        *reinterpret_cast<bf16x8 *>(y + p * 8) = o;
    }
}

// dW[8][CIN][9] += sum_p g[p][co] * x[c][p+tap], db[8] += sum_p g[p][co]  (g = grad w.r.t. pre-activation)
__global__ __launch_bounds__(256) void conv_first_wgrad_kernel(const float *__restrict__ x, int CIN, const bf16 *__restrict__ g,
                                                               const float *__restrict__ pre, int B, int H, int W,
                                                               float *__restrict__ part /*[gridDim.x][CIN][80]*/)
{
    __shared__ float red[4][80];
    const size_t npix = (size_t)B * H * W;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int c = 0; c < CIN; ++c) {
        float acc[80];
#pragma unroll
        for (int i = 0; i < 80; ++i) acc[i] = 0.f;
        for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (size_t)gridDim.x * blockDim.x) {
            const int xx = (int)(p % W), yy = (int)((p / W) % H), b = (int)(p / ((size_t)W * H));
            const bf16x8 gv = *reinterpret_cast<const bf16x8 *>(g + p * 8);
            const float *xc = x + ((size_t)b * CIN + c) * H * W;
            const float psub = pre ? pre[2 * c] : 0.f, prcp = pre ? pre[2 * c + 1] : 1.f;
            float xv[9];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int y2 = yy + tap / 3 - 1, x2 = xx + tap % 3 - 1;
                float v = 0.f;
                if (y2 >= 0 && y2 < H && x2 >= 0 && x2 < W) v = (xc[(size_t)y2 * W + x2] - psub) * prcp;
                xv[tap] = (float)(bf16)v;
            }
#pragma unroll
            for (int co = 0; co < 8; ++co) {
                const float gf = (float)gv[co];
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) acc[co * 9 + tap] += gf * xv[tap];
                acc[72 + co] += gf;
            }
        }
#pragma unroll
        for (int i = 0; i < 80; ++i) {
            float v = acc[i];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if (lane == 0) red[wv][i] = v;
        }
        __syncthreads();
        if (threadIdx.x < 80) {
            // this block's partial sums, plain stores: conv_first_wgrad_reduce_kernel adds the blocks in a fixed order
            part[((size_t)blockIdx.x * CIN + c) * 80 + threadIdx.x] =
                red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        }
        __syncthreads();
    }
}


// dW[8][CIN][9], db[8] = the per-block partial sums of the two kernels above added in block order (four interleaved
// chains per value, combined in a fixed order): no float atomics, the first layer's gradient is bit-reproducible.
// grid = CIN blocks of 320 threads: thread = (value i < 80, chain q < 4).
__global__ __launch_bounds__(320) void conv_first_wgrad_reduce_kernel(const float *__restrict__ part, int nblk, int CIN,
                                                                      float *__restrict__ dW, float *__restrict__ db)
{
    __shared__ float sh[4][80];
    const int c = blockIdx.x, i = threadIdx.x % 80, q = threadIdx.x / 80;
    // (same order of additions as a plain loop; eight loads in flight instead of one dependent load per addition, which
    // made this 160 KB reduction a 32 us kernel)
    float v = 0.f;
    int b = q;
    for (; b + 28 < nblk; b += 32) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = part[((size_t)(b + 4 * u) * CIN + c) * 80 + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) v += t[u];
    }
    for (; b < nblk; b += 4) v += part[((size_t)b * CIN + c) * 80 + i];
    sh[q][i] = v;
    __syncthreads();
    if (q == 0) {
        const float t = (sh[0][i] + sh[1][i]) + (sh[2][i] + sh[3][i]);
        if (i < 72) dW[((i / 9) * CIN + c) * 9 + (i % 9)] = t;
        else if (c == 0 && db) db[i - 72] = t;
    }
}

// Fast paths for W % 4 == 0: one thread per 4 horizontally adjacent pixels.  The three input rows
// are fetched as one 16-byte load + two edge scalars each (9 loads per 4 pixels instead of 36) and
// the index arithmetic is paid once per quad.
__device__ __forceinline__ void load_row6(const float *__restrict__ xc, int y2, int x0, int H, int W, float psub, float prcp,
                                          float (&v)[6])
{
    if (y2 < 0 || y2 >= H) {
#pragma unroll
        for (int i = 0; i < 6; ++i) v[i] = 0.f;
        return;
    }
    const float *row = xc + (size_t)y2 * W;
    const float4 m = *reinterpret_cast<const float4 *>(row + x0);
    v[0] = (x0 > 0) ? (row[x0 - 1] - psub) * prcp : 0.f;
    v[1] = (m.x - psub) * prcp; v[2] = (m.y - psub) * prcp; v[3] = (m.z - psub) * prcp; v[4] = (m.w - psub) * prcp;
    v[5] = (x0 + 4 < W) ? (row[x0 + 4] - psub) * prcp : 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i) v[i] = (float)(bf16)v[i];       // bf16 operands as on the MFMA path
}

// (LK = the LeakyReLU network, slope > 0: a compile-time switch -- as a run-time test of `slope` the activation of each of the
// 32 outputs of a thread became its own pair of branches, ~300 instructions beside the 144 packed FMAs)
template <bool LK>
__global__ __launch_bounds__(256) void conv_first_x4_kernel(const float *__restrict__ x, int CIN, const float *__restrict__ Wt,
                                                            const float *__restrict__ bias, const float *__restrict__ pre, int B,
                                                            int H, int W, float slope, bf16 *__restrict__ y)
{
    // (weights and bias are read at wave-uniform addresses: scalar loads, operands straight from SGPRs)
    float bs[8];
#pragma unroll
    for (int co = 0; co < 8; ++co) bs[co] = bias ? bias[co] : 0.f;
    const int Wq = W >> 2;
    const int nq = B * H * Wq;
    const int stride = gridDim.x * blockDim.x;
    // One input plane (the reference's default, fft only): the rows of the NEXT quad of the thread's walk are fetched raw (+ a
    // validity mask) while the current one is computed -- as in conv_first_wgrad_x4_kernel; with several planes the rows are
    // loaded where they are used (the general loop below).
    if (CIN == 1) {
        const float psub = pre ? pre[0] : 0.f, prcp = pre ? pre[1] : 1.f;
        float rv[3][6], rn[3][6];
        unsigned mv = 0, mn = 0;
        auto fetch = [&](int e, float (&ro)[3][6], unsigned &mo) {
            const int q = e % Wq, by = e / Wq;
            const int yy = by % H, b = by / H;
            const int x0 = q * 4;
            const float *xc = x + (size_t)b * H * W;
            unsigned m = 0;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int y2 = yy + dy - 1;
                const bool rok = y2 >= 0 && y2 < H;
                const float *row = xc + (size_t)(rok ? y2 : yy) * W;
                const bool lok = x0 > 0, hok = x0 + 4 < W;
                const float4 mid = *reinterpret_cast<const float4 *>(row + x0);
                ro[dy][0] = row[lok ? x0 - 1 : x0];
                ro[dy][1] = mid.x; ro[dy][2] = mid.y; ro[dy][3] = mid.z; ro[dy][4] = mid.w;
                ro[dy][5] = row[hok ? x0 + 4 : x0 + 3];
                m |= (rok ? ((lok ? 1u : 0u) | 0x1Eu | (hok ? 0x20u : 0u)) : 0u) << (6 * dy);
            }
            mo = m;
        };
        int e = blockIdx.x * blockDim.x + threadIdx.x;
        if (e < nq) fetch(e, rv, mv);
        for (; e < nq; e += stride) {
            const int en = e + stride;
            fetch(en < nq ? en : e, rn, mn);
            const int q = e % Wq, by = e / Wq;
            const int x0 = q * 4;
            float acc[4][8];
#pragma unroll
            for (int px = 0; px < 4; ++px)
#pragma unroll
                for (int co = 0; co < 8; ++co) acc[px][co] = bs[co];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                float v[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    const float t = (rv[dy][i] - psub) * prcp;
                    v[i] = ((mv >> (6 * dy + i)) & 1u) ? (float)(bf16)t : 0.f;       // bf16 operands as on the MFMA path
                }
#pragma unroll
                for (int co = 0; co < 8; ++co) {
#pragma unroll
                    for (int tx = 0; tx < 3; ++tx) {
                        const float w = Wt[co * 9 + dy * 3 + tx];
#pragma unroll
                        for (int px = 0; px < 4; ++px) acc[px][co] = __builtin_fmaf(v[px + tx], w, acc[px][co]);
                    }
                }
            }
            bf16 *dst = y + ((size_t)by * W + x0) * 8;
#pragma unroll
            for (int px = 0; px < 4; ++px) {
                bf16x8 o;
#pragma unroll
                for (int co = 0; co < 8; ++co) o[co] = (bf16)(LK ? act_leaky(acc[px][co], slope) : fmaxf(acc[px][co], 0.f));
                *reinterpret_cast<bf16x8 *>(dst + px * 8) = o;
            }
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int i = 0; i < 6; ++i) rv[dy][i] = rn[dy][i];
            mv = mn;
        }
        return;
    }
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < nq; e += gridDim.x * blockDim.x) {
        const int q = e % Wq, by = e / Wq;
        const int yy = by % H, b = by / H;
        const int x0 = q * 4;
        float acc[4][8];
#pragma unroll
        for (int px = 0; px < 4; ++px)
#pragma unroll
            for (int co = 0; co < 8; ++co) acc[px][co] = bs[co];
        for (int c = 0; c < CIN; ++c) {
            const float *xc = x + ((size_t)b * CIN + c) * H * W;
            const float psub = pre ? pre[2 * c] : 0.f, prcp = pre ? pre[2 * c + 1] : 1.f;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                float v[6];
                load_row6(xc, yy + dy - 1, x0, H, W, psub, prcp, v);
#pragma unroll
                for (int co = 0; co < 8; ++co) {
#pragma unroll
                    for (int tx = 0; tx < 3; ++tx) {
                        const float w = Wt[(co * CIN + c) * 9 + dy * 3 + tx];
#pragma unroll
                        for (int px = 0; px < 4; ++px) acc[px][co] = __builtin_fmaf(v[px + tx], w, acc[px][co]);
                    }
                }
            }
        }
        bf16 *dst = y + ((size_t)by * W + x0) * 8;
#pragma unroll
        for (int px = 0; px < 4; ++px) {
            bf16x8 o;
#pragma unroll
            for (int co = 0; co < 8; ++co) o[co] = (bf16)(LK ? act_leaky(acc[px][co], slope) : fmaxf(acc[px][co], 0.f));
            *reinterpret_cast<bf16x8 *>(dst + px * 8) = o;
        }
    }
}

__global__ __launch_bounds__(256) void conv_first_wgrad_x4_kernel(const float *__restrict__ x, int CIN, const bf16 *__restrict__ g,
                                                                  const float *__restrict__ pre, int B, int H, int W,
                                                                  float *__restrict__ part /*[gridDim.x][CIN][80]*/)
{
    __shared__ float red[4][80];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int Wq = W >> 2;
    const int nq = B * H * Wq;
    const int stride = gridDim.x * blockDim.x;
    for (int c = 0; c < CIN; ++c) {
        const float psub = pre ? pre[2 * c] : 0.f, prcp = pre ? pre[2 * c + 1] : 1.f;
        float acc[80];
#pragma unroll
        for (int i = 0; i < 80; ++i) acc[i] = 0.f;
        // Operands of the next quad are in flight while the current one is accumulated.  The image rows are fetched RAW (six
        // floats per row + a validity mask) and normalised / rounded to bf16 only when their quad is consumed: rounding them
        // inside the fetch (rounds 1-3) was a use of the loads right behind their issue -- every round waited for the memory
        // latency of its "prefetch" with 8 waves per CU to hide it.
        bf16x8 gv[4], gn[4];
        float rv[3][6], rn[3][6];
        unsigned mv = 0, mn = 0;                    // bit 6 dy + i: element i of row dy lies inside the image
        auto fetch = [&](int e, bf16x8 (&go)[4], float (&ro)[3][6], unsigned &mo) {
            const int q = e % Wq, by = e / Wq;
            const int yy = by % H, b = by / H;
            const int x0 = q * 4;
            const float *xc = x + ((size_t)b * CIN + c) * H * W;
            const bf16 *gp = g + ((size_t)by * W + x0) * 8;
#pragma unroll
            for (int px = 0; px < 4; ++px) go[px] = *reinterpret_cast<const bf16x8 *>(gp + px * 8);
            unsigned m = 0;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int y2 = yy + dy - 1;
                const bool rok = y2 >= 0 && y2 < H;
                const float *row = xc + (size_t)(rok ? y2 : yy) * W;          // (clamped: the loads stay unconditional)
                const bool lok = x0 > 0, hok = x0 + 4 < W;
                const float4 mid = *reinterpret_cast<const float4 *>(row + x0);
                ro[dy][0] = row[lok ? x0 - 1 : x0];
                ro[dy][1] = mid.x; ro[dy][2] = mid.y; ro[dy][3] = mid.z; ro[dy][4] = mid.w;
                ro[dy][5] = row[hok ? x0 + 4 : x0 + 3];
                m |= (rok ? ((lok ? 1u : 0u) | 0x1Eu | (hok ? 0x20u : 0u)) : 0u) << (6 * dy);
            }
            mo = m;
        };
        int e = blockIdx.x * blockDim.x + threadIdx.x;
        if (e < nq) fetch(e, gv, rv, mv);
        for (; e < nq; e += stride) {
            const int en = e + stride;
            fetch(en < nq ? en : e, gn, rn, mn);          // (past the end: the current quad again, unused)
            float v[3][6];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    const float t = (rv[dy][i] - psub) * prcp;
                    v[dy][i] = ((mv >> (6 * dy + i)) & 1u) ? (float)(bf16)t : 0.f;       // bf16 operands as on the MFMA path
                }
            float gf[4][8];
#pragma unroll
            for (int px = 0; px < 4; ++px)
#pragma unroll
                for (int co = 0; co < 8; ++co) gf[px][co] = (float)gv[px][co];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int co = 0; co < 8; ++co)
#pragma unroll
                    for (int tx = 0; tx < 3; ++tx)
#pragma unroll
                        for (int px = 0; px < 4; ++px)
                            acc[co * 9 + dy * 3 + tx] = __builtin_fmaf(gf[px][co], v[dy][px + tx], acc[co * 9 + dy * 3 + tx]);
#pragma unroll
            for (int co = 0; co < 8; ++co) acc[72 + co] += (gf[0][co] + gf[1][co]) + (gf[2][co] + gf[3][co]);
#pragma unroll
            for (int px = 0; px < 4; ++px) gv[px] = gn[px];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int i = 0; i < 6; ++i) rv[dy][i] = rn[dy][i];
            mv = mn;
        }
#pragma unroll
        for (int i = 0; i < 80; ++i) {
            float t = acc[i];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
            if (lane == 0) red[wv][i] = t;
        }
        __syncthreads();
        if (threadIdx.x < 80) {
            part[((size_t)blockIdx.x * CIN + c) * 80 + threadIdx.x] =
                red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// 2x2 / stride 2 max pooling (nn.MaxPool2d(2,2), icp_weight_policy.py:122-123), NHWC bf16,
// one thread per (output pixel, 8-channel granule).  Output (H/2, W/2) rounded down, as torch does
// (the polar 400 x 3360 network input reaches odd sizes: 25 x 105 at the fifth level).
// `arg` (optional): (B,H/2,W/2,C/2) bytes, one nibble per channel = position of the first maximum of the window in scan order
// | (maximum > 0) << 2 -- all the backward pass needs of the full-resolution tensor (maxpool2_bwd_arg_kernel).
__global__ void maxpool2_fwd_kernel(const bf16 *__restrict__ x, int B, int H, int W, int C, bf16 *__restrict__ y,
                                    unsigned *__restrict__ arg)
{
    const int Ho = H / 2, Wo = W / 2, G = C / 8;
    const size_t n = (size_t)B * Ho * Wo * G;
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const int gc = (int)(e % G);
    const size_t po = e / G;
    const int xo = (int)(po % Wo), yo = (int)((po / Wo) % Ho), b = (int)(po / ((size_t)Wo * Ho));
    const bf16 *src = x + ((((size_t)b * H + 2 * yo) * W + 2 * xo) * C + gc * 8);
    const bf16x8 v00 = *reinterpret_cast<const bf16x8 *>(src);
    const bf16x8 v01 = *reinterpret_cast<const bf16x8 *>(src + C);
    const bf16x8 v10 = *reinterpret_cast<const bf16x8 *>(src + (size_t)W * C);
    const bf16x8 v11 = *reinterpret_cast<const bf16x8 *>(src + (size_t)W * C + C);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j)
        o[j] = (bf16)fmaxf(fmaxf((float)v00[j], (float)v01[j]), fmaxf((float)v10[j], (float)v11[j]));
    *reinterpret_cast<bf16x8 *>(y + po * C + gc * 8) = o;
    if (arg != nullptr) {
        unsigned code = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float m = (float)v00[j];
            unsigned a = 0;
            if ((float)v01[j] > m) { m = (float)v01[j]; a = 1; }
            if ((float)v10[j] > m) { m = (float)v10[j]; a = 2; }
            if ((float)v11[j] > m) { m = (float)v11[j]; a = 3; }
            code |= (a | (m > 0.f ? 4u : 0u)) << (4 * j);
        }
        arg[e] = code;          // granule e = (pooled pixel, 8 channels) = 4 bytes of codes
    }
}

// maxpool2_bwd_kernel from the arg-max codes of the forward pass instead of the full-resolution tensor (ReLU network: the
// factor is (maximum > 0 ? scale : 0)): reads C/2 + 2 C bytes per window instead of 10 C, writes the same 8 C.
__global__ void maxpool2_bwd_arg_kernel(const unsigned *__restrict__ arg, const bf16 *__restrict__ gy, int B, int H, int W, int C,
                                        float scale, bf16 *__restrict__ gz)
{
    const int Ho = H / 2, Wo = W / 2, G = C / 8;
    const size_t n = (size_t)B * Ho * Wo * G;
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const int gc = (int)(e % G);
    const size_t po = e / G;
    const int xo = (int)(po % Wo), yo = (int)((po / Wo) % Ho), b = (int)(po / ((size_t)Wo * Ho));
    const size_t base = (((size_t)b * H + 2 * yo) * W + 2 * xo) * C + gc * 8;
    const size_t offs[4] = {0, (size_t)C, (size_t)W * C, (size_t)W * C + C};
    const unsigned code = arg[e];
    const bf16x8 g = *reinterpret_cast<const bf16x8 *>(gy + po * C + gc * 8);
    bf16x8 o[4];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const unsigned c = (code >> (4 * j)) & 0xFu;
        const float gv = (c & 4u) ? (float)g[j] * scale : 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k][j] = (bf16)(((unsigned)k == (c & 3u)) ? gv : 0.f);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) *reinterpret_cast<bf16x8 *>(gz + base + offs[k]) = o[k];
    // odd H / W (floor pooling): the last row / column belongs to no window and gets a zero gradient
    const bf16x8 z8 = {};
    const bool xr = (W & 1) && xo == Wo - 1, yr = (H & 1) && yo == Ho - 1;
    if (xr) {
        *reinterpret_cast<bf16x8 *>(gz + base + 2 * (size_t)C) = z8;
        *reinterpret_cast<bf16x8 *>(gz + base + (size_t)W * C + 2 * (size_t)C) = z8;
    }
    if (yr) {
        *reinterpret_cast<bf16x8 *>(gz + base + 2 * (size_t)W * C) = z8;
        *reinterpret_cast<bf16x8 *>(gz + base + 2 * (size_t)W * C + C) = z8;
        if (xr) *reinterpret_cast<bf16x8 *>(gz + base + 2 * (size_t)W * C + 2 * (size_t)C) = z8;
    }
}

// Backward of dropout(relu(.)) -> maxpool in one pass: the pooled gradient goes to the first
// maximal position (scan order, as torch does) times (d > 0 ? scale : 0).
__global__ void maxpool2_bwd_kernel(const bf16 *__restrict__ d, const bf16 *__restrict__ gy, int B, int H, int W, int C,
                                    float scale, float slope, bf16 *__restrict__ gz)
{
    const int Ho = H / 2, Wo = W / 2, G = C / 8;
    const size_t n = (size_t)B * Ho * Wo * G;
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const int gc = (int)(e % G);
    const size_t po = e / G;
    const int xo = (int)(po % Wo), yo = (int)((po / Wo) % Ho), b = (int)(po / ((size_t)Wo * Ho));
    const size_t base = (((size_t)b * H + 2 * yo) * W + 2 * xo) * C + gc * 8;
    const size_t offs[4] = {0, (size_t)C, (size_t)W * C, (size_t)W * C + C};
    bf16x8 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const bf16x8 *>(d + base + offs[k]);
    const bf16x8 g = *reinterpret_cast<const bf16x8 *>(gy + po * C + gc * 8);
    bf16x8 o[4];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float m = (float)v[0][j];
        int arg = 0;
#pragma unroll
        for (int k = 1; k < 4; ++k)
            if ((float)v[k][j] > m) { m = (float)v[k][j]; arg = k; }
        const float gv = (slope > 0.f) ? (float)g[j] * bwd_factor_leaky(m, scale, slope) : ((m > 0.f) ? (float)g[j] * scale : 0.f);
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k][j] = (bf16)((k == arg) ? gv : 0.f);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) *reinterpret_cast<bf16x8 *>(gz + base + offs[k]) = o[k];
    // odd H / W (floor pooling): the last row / column belongs to no window and gets a zero gradient
    const bf16x8 z8 = {};
    const bool xr = (W & 1) && xo == Wo - 1, yr = (H & 1) && yo == Ho - 1;
    if (xr) {
        *reinterpret_cast<bf16x8 *>(gz + base + 2 * (size_t)C) = z8;
        *reinterpret_cast<bf16x8 *>(gz + base + (size_t)W * C + 2 * (size_t)C) = z8;
    }
    if (yr) {
        *reinterpret_cast<bf16x8 *>(gz + base + 2 * (size_t)W * C) = z8;
        *reinterpret_cast<bf16x8 *>(gz + base + 2 * (size_t)W * C + C) = z8;
        if (xr) *reinterpret_cast<bf16x8 *>(gz + base + 2 * (size_t)W * C + 2 * (size_t)C) = z8;
    }
}

// ------------------------------------------------------------------------------------------
// Bilinear up-sampling with align_corners=True (nn.UpsamplingBilinear2d, icp_weight_policy.py:175-176).
__device__ __forceinline__ void up_coord(int o, float r, int n_src, int &i0, int &i1, float &l0, float &l1)
{
    const float f = r * (float)o;
    i0 = (int)f;
    i1 = i0 + ((i0 < n_src - 1) ? 1 : 0);
    l1 = f - (float)i0;
    l0 = 1.0f - l1;
}

// (element index -> (pixel, 8-channel granule): a shift when the granule count is a power of two)
template <bool POW2>
__device__ __forceinline__ void split_granule(int e, int G, int lg, int &pix, int &gc)
{
    if constexpr (POW2) {
        gc = e & (G - 1);
        pix = e >> lg;
    } else {
        gc = e % G;
        pix = e / G;
    }
}

// rh / rw = (n_src - 1) / (n_out - 1), worked out once on the host.  A thread owns one (column, granule)
// of UP_ROWS consecutive output rows: the column taps and weights are formed once, every source row is
// fetched and interpolated along x once (kept for the next output row, which shares it at factors >= 2),
// and an output row is one multiply and one fused multiply-add per channel.  Row bases are
// wave-uniform, per-lane offsets 32-bit.
constexpr int UP_ROWS = 4;

template <bool POW2>
__global__ __launch_bounds__(256) void upsample_fwd_kernel(const bf16 *__restrict__ x, int Hs, int Ws, int C, int Ho, int Wo,
                                                           float rh, float rw, int lg, bf16 *__restrict__ y)
{
    // grid: x = (output pixel, granule) of one output row, y = group of UP_ROWS output rows, z = image
    const int G = C >> 3;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= Wo * G) return;
    int gc, xo;
    split_granule<POW2>(e, G, lg, xo, gc);
    const int b = blockIdx.z;
    int x0, x1;
    float lx0, lx1;
    up_coord(xo, rw, Ws, x0, x1, lx0, lx1);
    const int o0 = x0 * C + gc * 8, o1 = x1 * C + gc * 8;
    const bf16 *img = x + (size_t)b * Hs * Ws * C;
    auto hrow = [&](int ysrc, float (&h)[8]) {
        const bf16 *r = img + (size_t)ysrc * Ws * C;
        float a0[8], a1[8];
        unpack8(*reinterpret_cast<const u32x4 *>(r + o0), a0);
        unpack8(*reinterpret_cast<const u32x4 *>(r + o1), a1);
#pragma unroll
        for (int j = 0; j < 8; ++j) h[j] = __builtin_fmaf(lx1, a1[j], lx0 * a0[j]);
    };
    float h0[8], h1[8];
    int hy0 = -1, hy1 = -1;                // source rows held in h0 / h1 (uniform over the block)
#pragma unroll
    for (int r = 0; r < UP_ROWS; ++r) {
        const int yo = blockIdx.y * UP_ROWS + r;
        if (yo >= Ho) break;
        int y0, y1;
        float ly0, ly1;
        up_coord(yo, rh, Hs, y0, y1, ly0, ly1);
        if (y0 != hy0) {
            if (y0 == hy1) {
#pragma unroll
                for (int j = 0; j < 8; ++j) h0[j] = h1[j];
            } else {
                hrow(y0, h0);
            }
            hy0 = y0;
        }
        if (y1 != hy1) {
            if (y1 == hy0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) h1[j] = h0[j];
            } else {
                hrow(y1, h1);
            }
            hy1 = y1;
        }
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = __builtin_fmaf(ly1, h1[j], ly0 * h0[j]);
        *reinterpret_cast<u32x4 *>(y + ((size_t)b * Ho + yo) * Wo * C + (size_t)e * 8) = pack8(o);
    }
}

// Gather form of the adjoint (deterministic, no atomics): each source pixel sums the output
// pixels it was interpolated into; optional ReLU/dropout factor of the source activation.
// A thread owns one (column, granule) of UP_ROWS consecutive source rows.  The weights along x do not
// depend on the row: the candidate output columns are scanned once per thread (exactly the forward's
// up_coord arithmetic) and the run of non-zero weights (at most NW wide for up-sampling factors >= 2) is
// compacted to the front.  Every output row in reach is then summed along x once (NW unconditional
// 16-byte loads) and added to the one or two source rows it was interpolated from.  Wider runs
// (factors < 2) take the generic loop.
template <bool POW2, bool LK>
__global__ __launch_bounds__(256) void upsample_bwd_kernel(const bf16 *__restrict__ gy, int Hs, int Ws, int C, int Ho, int Wo,
                                                           float rh, float rw, int lg, const bf16 *__restrict__ relu_src,
                                                           float scale, float slope, bf16 *__restrict__ gx)
{
    // grid: x = (source pixel, granule) of one source row, y = group of UP_ROWS source rows, z = image
    constexpr int MAXW = 8, NW = 5, R = UP_ROWS;
    const int G = C >> 3;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= Ws * G) return;
    int gc, xs;
    split_granule<POW2>(e, G, lg, xs, gc);
    const int ys0 = blockIdx.y * R, b = blockIdx.z;
    const int ys1 = min(ys0 + R, Hs) - 1;             // last source row of the group
    const int ylo = (rh > 0.f) ? max(0, (int)floorf((float)(ys0 - 1) / rh) - 1) : 0;
    const int yhi = (rh > 0.f) ? min(Ho - 1, (int)ceilf((float)(ys1 + 1) / rh) + 1) : Ho - 1;
    const int xlo = (rw > 0.f) ? max(0, (int)floorf((float)(xs - 1) / rw) - 1) : 0;
    const int xhi = (rw > 0.f) ? min(Wo - 1, (int)ceilf((float)(xs + 1) / rw) + 1) : Wo - 1;
    float acc[R][8];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[r][j] = 0.f;
    float wxs[MAXW];
    int k0 = MAXW, k1 = -1;
#pragma unroll
    for (int k = 0; k < MAXW; ++k) {
        const int xo = xlo + k;
        int x0, x1;
        float m0, m1;
        up_coord(xo, rw, Ws, x0, x1, m0, m1);
        const float wx = ((x0 == xs) ? m0 : 0.f) + ((x1 == xs) ? m1 : 0.f);
        wxs[k] = (xo <= xhi) ? wx : 0.f;
        if (wxs[k] != 0.f) {
            k0 = min(k0, k);
            k1 = k;
        }
    }
    k0 = (k1 < 0) ? 0 : k0;
    // every lane of the wave must fit the compact form (the loops below are wave-uniform)
    const bool compact = __all((xhi - xlo + 1) <= MAXW && k0 <= MAXW - NW && k1 - k0 < NW);
    const bf16 *gimg = gy + (size_t)b * Ho * Wo * C + gc * 8;
    float wn[NW];
    int on[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        float w = wxs[i];                       // k0 == 0
#pragma unroll
        for (int q = 1; q <= MAXW - NW; ++q) w = (k0 == q) ? wxs[i + q] : w;
        wn[i] = w;
        on[i] = min(xlo + k0 + i, Wo - 1) * C;  // (past the run the weight is zero)
    }
    for (int yo = ylo; yo <= yhi; ++yo) {
        int y0, y1;
        float l0, l1;
        up_coord(yo, rh, Hs, y0, y1, l0, l1);
        if (y1 < ys0 || y0 > ys1) continue;                 // (uniform: the rows are the block's)
        const bf16 *grow = gimg + (size_t)yo * Wo * C;
        float hs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // the output row summed along x for this column
        if (compact) {
            u32x4 v[NW];
#pragma unroll
            for (int i = 0; i < NW; ++i) v[i] = *reinterpret_cast<const u32x4 *>(grow + on[i]);
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                float g[8];
                unpack8(v[i], g);
#pragma unroll
                for (int j = 0; j < 8; ++j) hs[j] = __builtin_fmaf(wn[i], g[j], hs[j]);
            }
        } else {
            for (int xo = xlo; xo <= xhi; ++xo) {
                int x0, x1;
                float m0, m1;
                up_coord(xo, rw, Ws, x0, x1, m0, m1);
                const float wx = ((x0 == xs) ? m0 : 0.f) + ((x1 == xs) ? m1 : 0.f);
                if (wx == 0.f) continue;
                float g[8];
                unpack8(*reinterpret_cast<const u32x4 *>(grow + (size_t)xo * C), g);
#pragma unroll
                for (int j = 0; j < 8; ++j) hs[j] = __builtin_fmaf(wx, g[j], hs[j]);
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int ys = ys0 + r;
            const float wy = ((y0 == ys) ? l0 : 0.f) + ((y1 == ys) ? l1 : 0.f);
            if (wy != 0.f) {
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[r][j] = __builtin_fmaf(wy, hs[j], acc[r][j]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int ys = ys0 + r;
        if (ys >= Hs) break;
        const size_t po = (((size_t)b * Hs + ys) * Ws) * C + (size_t)e * 8;
        if (relu_src) {
            float sv[8];
            unpack8(*reinterpret_cast<const u32x4 *>(relu_src + po), sv);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                acc[r][j] = LK ? acc[r][j] * bwd_factor_leaky(sv[j], scale, slope) : ((sv[j] > 0.f) ? acc[r][j] * scale : 0.f);
        }
        *reinterpret_cast<u32x4 *>(gx + po) = pack8(acc[r]);
    }
}

// ------------------------------------------------------------------------------------------
// Final layer: Conv2d(8 -> 1, 1x1) + Sigmoid (icp_weight_policy.py:96-99,184), fp32 mask out.
__global__ void final_fwd_kernel(const bf16 *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                                 size_t npix, float *__restrict__ mask)
{
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    const bf16x8 v = *reinterpret_cast<const bf16x8 *>(x + p * 8);
    float acc = bias[0];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += (float)v[j] * (float)(bf16)w[j];
    mask[p] = 1.0f / (1.0f + expf(-acc));
}

// grid: x strides over the pixels of one image, y = image.  coef (optional, 2 floats per image) carries the
// adjoint of the mask's amax normalisation (see mask_norm_*): the gradient w.r.t. the raw sigmoid output is
// g * (1 / a) + (m == a ? t : 0)  with a = amax, t = -(sum g m_n) / a / count(m == a).
template <bool LK>        // LK: the LeakyReLU network (slope > 0), a compile-time switch (no branch per value)
__global__ __launch_bounds__(256) void final_bwd_kernel(const bf16 *__restrict__ x, const float *__restrict__ w,
                                                        const float *__restrict__ mask, const float *__restrict__ gmask,
                                                        size_t npix_per, const float *__restrict__ coef, float scale, float slope,
                                                        bf16 *__restrict__ gx, float *__restrict__ part /*[blocks][9]*/)
{
    __shared__ float red[4][9];
    float acc[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float wv8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) wv8[j] = (float)(bf16)w[j];
    const size_t base = (size_t)blockIdx.y * npix_per;
    const float na = coef ? coef[2 * blockIdx.y] : 1.0f, nt = coef ? coef[2 * blockIdx.y + 1] : 0.f;
    const float rna = 1.0f / na;
    // Four pixels of the thread's stride walk per round, their twelve loads issued before the first use: with ~512 blocks
    // (the ordered reduction's workspace bounds the count) a CU holds 8 waves, and one dependent load -> compute -> store chain
    // per wave leaves the memory system idle most of the time.  The sums are still taken pixel by pixel in walk order.
    constexpr int U = 4;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t q0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q0 < npix_per; q0 += U * stride) {
        float m[U], gmv[U];
        bf16x8 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t q = q0 + u * stride;
            const size_t p = base + (q < npix_per ? q : q0);           // (clamped: the loads stay unconditional)
            m[u] = mask[p];
            gmv[u] = gmask[p];
            v[u] = *reinterpret_cast<const bf16x8 *>(x + p * 8);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t q = q0 + u * stride;
            if (q >= npix_per) break;
            const size_t p = base + q;
            float gm = gmv[u];
            if (coef) gm = gm * rna + ((m[u] == na) ? nt : 0.f);
            const float gl = gm * m[u] * (1.0f - m[u]);
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xv = (float)v[u][j];
                o[j] = (bf16)(LK ? gl * wv8[j] * bwd_factor_leaky(xv, scale, slope) : ((xv > 0.f) ? gl * wv8[j] * scale : 0.f));
                acc[j] += gl * xv;
            }
            acc[8] += gl;
            *reinterpret_cast<bf16x8 *>(gx + p * 8) = o;
        }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        float v = acc[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0) red[wv][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < 9)      // this block's partial sums (plain stores; final_bwd_reduce_kernel adds the blocks in a fixed order)
        part[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 9 + threadIdx.x] =
            red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// dW[8], db[1] = the block partials of final_bwd_kernel added in block order (32 interleaved chains per value, combined by a
// fixed butterfly): no float atomics.  One block of 9 x 32 threads.
__global__ __launch_bounds__(288) void final_bwd_reduce_kernel(const float *__restrict__ part, int nblk, float *__restrict__ dW,
                                                               float *__restrict__ db)
{
    __shared__ float sh[9][32];
    const int i = threadIdx.x / 32, q = threadIdx.x % 32;
    float v = 0.f;
    for (int b = q; b < nblk; b += 32) v += part[(size_t)b * 9 + i];
    sh[i][q] = v;
    __syncthreads();
    for (int off = 16; off > 0; off >>= 1) {
        if (q < off) sh[i][q] += sh[i][q + off];
        __syncthreads();
    }
    if (q == 0) {
        if (i < 8) dW[i] = sh[i][0];
        else db[0] = sh[8][0];
    }
}

// ------------------------------------------------------------------------------------------
// mask / amax(mask over H,W) per image (icp_weight_policy.py:192-193) and its adjoint, fused around the
// final layer: the forward is two passes (segment maxima, then the division), the backward one reduction
// (sum of g * m_n and the number of maximal pixels: torch.amax spreads its gradient evenly over ties)
// whose result final_bwd_kernel applies on the fly.
constexpr int MASK_SEG = 64;           // segments (blocks) per image

__device__ __forceinline__ float block_reduce_max_256(float v, float *sh /*4*/)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_down(v, off, 64));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}

__global__ __launch_bounds__(256) void mask_segmax_kernel(const float *__restrict__ mask, size_t npix_per, float *__restrict__ part)
{
    __shared__ float sh[4];
    const float *m = mask + (size_t)blockIdx.y * npix_per;
    float v = -INFINITY;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < npix_per; q += (size_t)gridDim.x * blockDim.x)
        v = fmaxf(v, m[q]);
    v = block_reduce_max_256(v, sh);
    if (threadIdx.x == 0) part[(size_t)blockIdx.y * MASK_SEG + blockIdx.x] = v;
}

__global__ __launch_bounds__(256) void mask_scale_kernel(const float *__restrict__ mask, size_t npix_per, const float *__restrict__ part,
                                                         float *__restrict__ mask_n, float *__restrict__ amax)
{
    __shared__ float sh[4];
    const float pv = (threadIdx.x < MASK_SEG) ? part[(size_t)blockIdx.y * MASK_SEG + threadIdx.x] : -INFINITY;
    const float a = block_reduce_max_256(pv, sh);
    if (blockIdx.x == 0 && threadIdx.x == 0) amax[blockIdx.y] = a;
    const size_t base = (size_t)blockIdx.y * npix_per;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < npix_per; q += (size_t)gridDim.x * blockDim.x)
        mask_n[base + q] = mask[base + q] / a;
}

__global__ __launch_bounds__(256) void mask_norm_bwd_partial_kernel(const float *__restrict__ g, const float *__restrict__ mask_n,
                                                                    size_t npix_per, float *__restrict__ part /*B*SEG*2*/)
{
    __shared__ float red[4][2];
    const size_t base = (size_t)blockIdx.y * npix_per;
    float s = 0.f, c = 0.f;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < npix_per; q += (size_t)gridDim.x * blockDim.x) {
        const float mn = mask_n[base + q];
        s = __builtin_fmaf(g[base + q], mn, s);
        c += (mn == 1.0f) ? 1.f : 0.f;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_down(s, off, 64);
        c += __shfl_down(c, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        red[threadIdx.x >> 6][0] = s;
        red[threadIdx.x >> 6][1] = c;
    }
    __syncthreads();
    if (threadIdx.x < 2)
        part[((size_t)blockIdx.y * MASK_SEG + blockIdx.x) * 2 + threadIdx.x] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

__global__ void mask_norm_bwd_final_kernel(const float *__restrict__ part, const float *__restrict__ amax, int B,
                                           float *__restrict__ coef)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float s = 0.f, c = 0.f;
    for (int k = 0; k < MASK_SEG; ++k) {
        s += part[((size_t)b * MASK_SEG + k) * 2];
        c += part[((size_t)b * MASK_SEG + k) * 2 + 1];
    }
    const float a = amax[b];
    coef[2 * b] = a;
    coef[2 * b + 1] = (c > 0.f) ? -(s / a) / c : 0.f;
}

// Per-channel minimum / maximum of an fp32 NCHW tensor over (B,H,W) -> pre[2c] = min, pre[2c+1] = 1 / (max - min):
// the offset and reciprocal scale of the policy's min-max normalisation (icp_weight_policy.py:151-155),
// consumed by the first-layer kernels while they load.
constexpr int MM_BLOCKS = 1024;       // blocks (and partial results) per channel

__global__ __launch_bounds__(256) void channel_minmax_partial_kernel(const float *__restrict__ x, int B, int C, size_t hw,
                                                                     float *__restrict__ part /*C*MM_BLOCKS*2*/)
{
    __shared__ float shn[4], shx[4];
    const int c = blockIdx.y;
    float mn = INFINITY, mx = -INFINITY;
    const bool vec = (hw & 3) == 0;
    for (int b = 0; b < B; ++b) {
        const float *xc = x + ((size_t)b * C + c) * hw;
        if (vec) {
            const float4 *x4 = reinterpret_cast<const float4 *>(xc);
            for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < hw / 4; q += (size_t)gridDim.x * blockDim.x) {
                const float4 v = x4[q];
                mn = fminf(fminf(mn, v.x), fminf(v.y, fminf(v.z, v.w)));
                mx = fmaxf(fmaxf(mx, v.x), fmaxf(v.y, fmaxf(v.z, v.w)));
            }
        } else {
            for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < hw; q += (size_t)gridDim.x * blockDim.x) {
                mn = fminf(mn, xc[q]);
                mx = fmaxf(mx, xc[q]);
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        mn = fminf(mn, __shfl_down(mn, off, 64));
        mx = fmaxf(mx, __shfl_down(mx, off, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        shn[threadIdx.x >> 6] = mn;
        shx[threadIdx.x >> 6] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[((size_t)c * MM_BLOCKS + blockIdx.x) * 2] = fminf(fminf(shn[0], shn[1]), fminf(shn[2], shn[3]));
        part[((size_t)c * MM_BLOCKS + blockIdx.x) * 2 + 1] = fmaxf(fmaxf(shx[0], shx[1]), fmaxf(shx[2], shx[3]));
    }
}

__global__ __launch_bounds__(256) void channel_minmax_final_kernel(const float *__restrict__ part, int C, float *__restrict__ pre,
                                                                   float *__restrict__ minmax)
{
    __shared__ float shn[4], shx[4];
    const int c = blockIdx.x;
    float mn = INFINITY, mx = -INFINITY;
    for (int k = threadIdx.x; k < MM_BLOCKS; k += blockDim.x) {
        mn = fminf(mn, part[((size_t)c * MM_BLOCKS + k) * 2]);
        mx = fmaxf(mx, part[((size_t)c * MM_BLOCKS + k) * 2 + 1]);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        mn = fminf(mn, __shfl_down(mn, off, 64));
        mx = fmaxf(mx, __shfl_down(mx, off, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        shn[threadIdx.x >> 6] = mn;
        shx[threadIdx.x >> 6] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        mn = fminf(fminf(shn[0], shn[1]), fminf(shn[2], shn[3]));
        mx = fmaxf(fmaxf(shx[0], shx[1]), fmaxf(shx[2], shx[3]));
        pre[2 * c] = mn;
        pre[2 * c + 1] = 1.0f / (mx - mn);
        if (minmax) {
            minmax[2 * c] = mn;
            minmax[2 * c + 1] = mx;
        }
    }
}

// ------------------------------------------------------------------------------------------
// nn.BatchNorm2d of the network's batch-norm variant (params["batch_norm"], icp_weight_policy.py:108-113: it
// follows the ReLU of each convolution), on NHWC bf16 tensors.  Training mode: batch statistics over (B,H,W) per
// channel, fp32 partial sums per block in a fixed order + an fp64 final sum (deterministic), the running
// statistics updated as torch does (momentum, unbiased variance).  The affine form y = a * scale + shift is applied
// by a separate streaming pass that also draws the block's dropout (the Dropout module follows the second
// BatchNorm).  A value that is kept but exactly zero is stored as -0.0 and a dropped one as +0.0, the convention of
// the LeakyReLU variant: the backward kernels recover the dropout mask from the stored tensor alone.
constexpr int BN_BLOCKS = 512;

__global__ __launch_bounds__(256) void bn_stats_kernel(const bf16 *__restrict__ a, size_t npix, int C, float *__restrict__ part)
{
    __shared__ float red[256][17];
    const int G = C >> 3, per = 256 / G;                  // granules per pixel, pixels per block pass
    const int gc = threadIdx.x % G, pl = threadIdx.x / G;
    float s[8], q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = q[j] = 0.f;
    for (size_t p = (size_t)blockIdx.x * per + pl; p < npix; p += (size_t)gridDim.x * per) {
        float v[8];
        unpack8(*reinterpret_cast<const u32x4 *>(a + p * C + gc * 8), v);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            s[j] += v[j];
            q[j] = __builtin_fmaf(v[j], v[j], q[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        red[threadIdx.x][j] = s[j];
        red[threadIdx.x][8 + j] = q[j];
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < G * 16; idx += blockDim.x) {
        const int g2 = idx / 16, k = idx % 16;
        float t = 0.f;
        for (int r = 0; r < per; ++r) t += red[r * G + g2][k];
        const int c = g2 * 8 + (k & 7);
        part[((size_t)blockIdx.x * C + c) * 2 + (k >> 3)] = t;
    }
}

// stat[c] = (mean, invstd); affine[c] = (scale, shift) = (gamma invstd, beta - mean gamma invstd)
__global__ void bn_finalize_kernel(const float *__restrict__ part, int nblk, int C, double n, const float *__restrict__ gamma,
                                   const float *__restrict__ beta, float eps, float momentum, float *__restrict__ running_mean,
                                   float *__restrict__ running_var, float *__restrict__ stat, float *__restrict__ affine)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0.0, q = 0.0;
    for (int b = 0; b < nblk; ++b) {
        s += (double)part[((size_t)b * C + c) * 2];
        q += (double)part[((size_t)b * C + c) * 2 + 1];
    }
    const double mean = s / n;
    double var = q / n - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const double invstd = 1.0 / sqrt(var + (double)eps);
    stat[2 * c] = (float)mean;
    stat[2 * c + 1] = (float)invstd;
    const double sc = (double)gamma[c] * invstd;
    affine[2 * c] = (float)sc;
    affine[2 * c + 1] = (float)((double)beta[c] - mean * sc);
    if (running_mean != nullptr) {
        running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
        const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
        running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
    }
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const bf16 *__restrict__ a, size_t ngran, int C, const float *__restrict__ affine,
                                                       float drop_p, unsigned seed, bf16 *__restrict__ y)
{
    const DropoutParams dp = dropout_params(drop_p);
    const int G = C >> 3;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < ngran; e += (size_t)gridDim.x * blockDim.x) {
        const int gc = (int)(e % G);
        float v[8];
        unpack8(*reinterpret_cast<const u32x4 *>(a + e * 8), v);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float t = __builtin_fmaf(v[j], affine[2 * (gc * 8 + j)], affine[2 * (gc * 8 + j) + 1]);
            v[j] = (t == 0.f) ? -0.f : t;
        }
        if (drop_p > 0.f) {
            float sc[4];
            dropout_scale4(seed, (unsigned)(e * 8), dp, sc);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = drop_leaky(v[j], sc[j]);
            dropout_scale4(seed, (unsigned)(e * 8 + 4), dp, sc);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[4 + j] = drop_leaky(v[4 + j], sc[j]);
        }
        *reinterpret_cast<u32x4 *>(y + e * 8) = pack8(v);
    }
}

// gy = gd * (d != +0 ? drop_scale : 0) when the block's dropout followed this BatchNorm (d given), else gd
__device__ __forceinline__ void bn_gy8(const bf16 *__restrict__ gd, const bf16 *__restrict__ d, size_t off, float drop_scale, float (&g)[8])
{
    unpack8(*reinterpret_cast<const u32x4 *>(gd + off), g);
    if (d != nullptr) {
        const u32x4 dv = *reinterpret_cast<const u32x4 *>(d + off);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            g[2 * j] = ((dv[j] & 0xffffu) != 0u) ? g[2 * j] * drop_scale : 0.f;
            g[2 * j + 1] = ((dv[j] >> 16) != 0u) ? g[2 * j + 1] * drop_scale : 0.f;
        }
    }
}

// partial sums of gy and gy * xhat per channel (xhat = (a - mean) invstd)
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const bf16 *__restrict__ gd, const bf16 *__restrict__ d, float drop_scale,
                                                            const bf16 *__restrict__ a, size_t npix, int C,
                                                            const float *__restrict__ stat, float *__restrict__ part)
{
    __shared__ float red[256][17];
    const int G = C >> 3, per = 256 / G;
    const int gc = threadIdx.x % G, pl = threadIdx.x / G;
    float mean[8], istd[8], s[8], q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        mean[j] = stat[2 * (gc * 8 + j)];
        istd[j] = stat[2 * (gc * 8 + j) + 1];
        s[j] = q[j] = 0.f;
    }
    for (size_t p = (size_t)blockIdx.x * per + pl; p < npix; p += (size_t)gridDim.x * per) {
        float g[8], v[8];
        bn_gy8(gd, d, p * C + gc * 8, drop_scale, g);
        unpack8(*reinterpret_cast<const u32x4 *>(a + p * C + gc * 8), v);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            s[j] += g[j];
            q[j] = __builtin_fmaf(g[j], (v[j] - mean[j]) * istd[j], q[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        red[threadIdx.x][j] = s[j];
        red[threadIdx.x][8 + j] = q[j];
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < G * 16; idx += blockDim.x) {
        const int g2 = idx / 16, k = idx % 16;
        float t = 0.f;
        for (int r = 0; r < per; ++r) t += red[r * G + g2][k];
        const int c = g2 * 8 + (k & 7);
        part[((size_t)blockIdx.x * C + c) * 2 + (k >> 3)] = t;
    }
}

// coef[c] = (gamma invstd, sum gy / n, sum gy xhat / n); dgamma / dbeta written or accumulated
__global__ void bn_bwd_finalize_kernel(const float *__restrict__ part, int nblk, int C, double n, const float *__restrict__ gamma,
                                       const float *__restrict__ stat, int accumulate, float *__restrict__ coef,
                                       float *__restrict__ dgamma, float *__restrict__ dbeta)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0.0, q = 0.0;
    for (int b = 0; b < nblk; ++b) {
        s += (double)part[((size_t)b * C + c) * 2];
        q += (double)part[((size_t)b * C + c) * 2 + 1];
    }
    coef[3 * c] = gamma[c] * stat[2 * c + 1];
    coef[3 * c + 1] = (float)(s / n);
    coef[3 * c + 2] = (float)(q / n);
    dbeta[c] = accumulate ? dbeta[c] + (float)s : (float)s;
    dgamma[c] = accumulate ? dgamma[c] + (float)q : (float)q;
}

__global__ void bn_eval_coef_kernel(const float *__restrict__ affine, int C, float *__restrict__ coef)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) coef[3 * c] = affine[2 * c];
}

// gz = [k (gy - m1 - xhat m2)] * act'(a): the gradient w.r.t. the convolution's pre-activation.  Evaluation mode
// (stat == nullptr): gz = gy * scale * act'(a), scale = coef[3c].
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const bf16 *__restrict__ gd, const bf16 *__restrict__ d, float drop_scale,
                                                           const bf16 *__restrict__ a, size_t ngran, int C,
                                                           const float *__restrict__ stat, const float *__restrict__ coef, float slope,
                                                           bf16 *__restrict__ gz)
{
    const int G = C >> 3;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < ngran; e += (size_t)gridDim.x * blockDim.x) {
        const int gc = (int)(e % G);
        float g[8], v[8], o[8];
        bn_gy8(gd, d, e * 8, drop_scale, g);
        unpack8(*reinterpret_cast<const u32x4 *>(a + e * 8), v);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = gc * 8 + j;
            float ga;
            if (stat != nullptr) {
                const float xh = (v[j] - stat[2 * c]) * stat[2 * c + 1];
                ga = coef[3 * c] * ((g[j] - coef[3 * c + 1]) - xh * coef[3 * c + 2]);
            } else {
                ga = g[j] * coef[3 * c];
            }
            // a is the ReLU / LeakyReLU output: for the leaky network a kept zero is -0.0 (negative side)
            const float f = (slope > 0.f) ? ((v[j] > 0.f) ? 1.f : slope) : ((v[j] > 0.f) ? 1.f : 0.f);
            o[j] = ga * f;
        }
        *reinterpret_cast<u32x4 *>(gz + e * 8) = pack8(o);
    }
}

bool chan_ok(int c) { return c == 8 || c == 16 || c == 32 || (c >= 64 && c % 64 == 0); }

}  // namespace

#ifdef MMK_DEEP_STAMPS
extern "C" int mmk_debug_deep_stamps(unsigned long long *buf)
{
    MMK_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_deep_stamp_buf), &buf, sizeof(buf)));
    return MMK_OK;
}
#endif

// ================================================================================== C ABI
extern "C" size_t mmk_conv3x3_packed_elems(int32_t cout, int32_t cin, int32_t transposed)
{
    const int co_n = transposed ? cin : cout, ci_n = transposed ? cout : cin;
    if (!chan_ok(co_n) || !chan_ok(ci_n)) return 0;
    const int CK = conv_ck(ci_n), CM = conv_cm(ci_n, co_n);
    const int groups = (co_n + CM - 1) / CM, chunks = ci_n / CK;
    return (size_t)groups * chunks * ksteps(CK) * (CM / 16) * 512;
}

extern "C" int mmk_conv3x3_pack_weights(const float *W, int32_t cout, int32_t cin, int32_t transposed, void *packed,
                                        void *stream)
{
    MMK_REQUIRE(W && packed, "mmk_conv3x3_pack_weights: NULL pointer");
    const size_t total = mmk_conv3x3_packed_elems(cout, cin, transposed);
    MMK_REQUIRE(total > 0, "mmk_conv3x3_pack_weights: unsupported channel counts %d -> %d", cin, cout);
    hipLaunchKernelGGL(pack_conv_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W,
                       cout, cin, transposed, (bf16 *)packed, (int)total);
    MMK_LAUNCH_CHECK();
    return MMK_OK;
}

extern "C" int mmk_conv3x3_pack_weights_batch(int32_t n, const float *const *W, const int32_t *cout, const int32_t *cin,
                                              int32_t transposed, void *const *packed, void *stream)
{
    MMK_REQUIRE(n >= 1 && W && cout && cin && packed, "mmk_conv3x3_pack_weights_batch: bad argument");
    for (int base = 0; base < n; base += PACK_BATCH_MAX) {
        PackBatch pb;
        const int m = std::min(PACK_BATCH_MAX, n - base);
        size_t most = 0;
        for (int i = 0; i < PACK_BATCH_MAX; ++i) {
            const int k = base + (i < m ? i : 0);
            MMK_REQUIRE(W[k] && packed[k], "mmk_conv3x3_pack_weights_batch: NULL pointer (layer %d)", k);
            const size_t total = mmk_conv3x3_packed_elems(cout[k], cin[k], transposed);
            MMK_REQUIRE(total > 0 && total < ((size_t)1 << 31), "mmk_conv3x3_pack_weights_batch: unsupported channel counts %d -> %d",
                        cin[k], cout[k]);
            pb.W[i] = W[k]; pb.out[i] = (bf16 *)packed[k]; pb.cout[i] = cout[k]; pb.cin[i] = cin[k];
            pb.total[i] = (int)total;
            most = std::max(most, total);
        }
        pb.transposed = transposed;
        const unsigned bx = (unsigned)std::min<size_t>((most + 255) / 256, 256);
        hipLaunchKernelGGL(pack_conv_weights_batch_kernel, dim3(bx, m), dim3(256), 0, (hipStream_t)stream, pb);
        MMK_LAUNCH_CHECK();
    }
    return MMK_OK;
}

extern "C" int mmk_conv3x3(const mmk_conv_desc *d, void *stream)
{
    MMK_REQUIRE(d != nullptr, "mmk_conv3x3: NULL descriptor");
    MMK_REQUIRE(d->x1 && d->wpack && (d->y1 || (d->pool_y && d->pool_arg)), "mmk_conv3x3: NULL pointer");
    MMK_REQUIRE(d->pool_arg == nullptr || d->pool_y != nullptr, "mmk_conv3x3: pool_arg needs pool_y");
    MMK_REQUIRE(d->B >= 1 && d->H >= 1 && d->W >= 1, "mmk_conv3x3: bad shape");
    const int cin = d->C1 + d->C2, cout = d->O1 + d->O2;
    MMK_REQUIRE(d->C1 % 8 == 0 && d->C2 % 8 == 0 && (d->C2 == 0 || d->x2), "mmk_conv3x3: bad input split %d+%d", d->C1, d->C2);
    MMK_REQUIRE(d->O1 % 8 == 0 && d->O2 % 8 == 0 && (d->O2 == 0 || d->y2), "mmk_conv3x3: bad output split %d+%d", d->O1, d->O2);
    MMK_REQUIRE(chan_ok(cin) && chan_ok(cout), "mmk_conv3x3: unsupported channel counts %d -> %d", cin, cout);
    MMK_REQUIRE(d->drop_p >= 0.f && d->drop_p < 1.f, "mmk_conv3x3: dropout probability out of range");
    ConvArgs a;
    a.x1 = (const bf16 *)d->x1; a.x2 = (const bf16 *)d->x2; a.C1 = d->C1; a.C2 = d->C2;
    a.wpack = (const bf16 *)d->wpack; a.bias = d->bias;
    a.o1 = {(bf16 *)d->y1, (const bf16 *)d->relu_src1, d->O1, d->accumulate1, d->scale1};
    a.o2 = {(bf16 *)d->y2, (const bf16 *)d->relu_src2, d->O2, d->accumulate2, d->scale2};
    a.B = d->B; a.H = d->H; a.W = d->W; a.CIN = cin; a.COUT = cout;
    MMK_REQUIRE(d->leaky_slope >= 0.f && d->leaky_slope <= 1.f, "mmk_conv3x3: leaky_slope must be in [0, 1]");
    a.relu = d->relu; a.slope = d->leaky_slope; a.drop_p = d->drop_p; a.seed = d->seed;
    a.pool_y = (bf16 *)d->pool_y;
    a.pool_arg = (unsigned char *)d->pool_arg;
    return dispatch_conv(a, (hipStream_t)stream);
}

extern "C" int32_t mmk_conv3x3_pool_fusable(int32_t cin, int32_t cout, int32_t B, int32_t H, int32_t W)
{
    return pool_fusable(cin, cout, B, H, W) ? 1 : 0;
}

extern "C" int mmk_conv3x3_wgrad_unpack_batch(int32_t n, const float *const *src, const int32_t *slices, const int32_t *cout,
                                              const int32_t *cin, float *const *dW, float *const *db, void *stream)
{
    MMK_REQUIRE(n >= 1 && src && cout && cin && dW, "mmk_conv3x3_wgrad_unpack_batch: bad argument");
    for (int base = 0; base < n; base += PACK_BATCH_MAX) {
        UnpackBatch ub;
        const int m = std::min(PACK_BATCH_MAX, n - base);
        int most = 0;       // blocks the layer with the most of them needs
        for (int i = 0; i < PACK_BATCH_MAX; ++i) {
            const int k = base + (i < m ? i : 0);
            MMK_REQUIRE(src[k] && dW[k] && cout[k] >= 1 && cin[k] >= 1, "mmk_conv3x3_wgrad_unpack_batch: bad layer %d", k);
            ub.src[i] = src[k]; ub.dW[i] = dW[k]; ub.cout[i] = cout[k]; ub.cin[i] = cin[k];
            ub.slices[i] = slices ? slices[k] : 0;
            ub.db[i] = (db && ub.slices[i] > 0) ? db[k] : nullptr;
            const int total = cout[k] * cin[k] * 9 + (ub.slices[i] > 0 ? cout[k] : 0);
            const int epb = 4 * (256 / unpack_groups(total));
            most = std::max(most, (total + epb - 1) / epb);
        }
        const unsigned bx = (unsigned)std::min(most, 2048);
        hipLaunchKernelGGL(unpack_wgrad_batch_kernel, dim3(bx, m), dim3(256), 0, (hipStream_t)stream, ub);
        MMK_LAUNCH_CHECK();
    }
    return MMK_OK;
}

template <int CX, int CG>
int launch_bwd_fused(const BwdFusedArgs &a, int c1, hipStream_t st)
{
    const int spatial = wgrad_slices(CG, CX, c1, a.B, a.H, a.W);     // the partial slices of the two-kernel path: same layout, same sums
    if (spatial < 1) {
        mmk::set_error("fused backward: occupancy query failed");
        return MMK_ERR_HIP;
    }
    hipLaunchKernelGGL((conv_bwd_fused_kernel<CX, CG>), dim3(spatial), dim3(CONV_THREADS), 0, st, a);
    MMK_LAUNCH_CHECK();
    return MMK_OK;
}

extern "C" int mmk_conv_bwd_fused(const void *x, const void *g, const void *wpack_t, float scale, int32_t B, int32_t H, int32_t W,
                                  int32_t C, void *dx, float *partials, int32_t accumulate, void *stream)
{
    MMK_REQUIRE(x && g && wpack_t && dx && partials, "mmk_conv_bwd_fused: NULL pointer");
    MMK_REQUIRE(B >= 1 && H >= 2 && W >= 2 && (C == 8 || C == 16), "mmk_conv_bwd_fused: bad shape (8 or 16 channels)");
    MMK_REQUIRE((size_t)B * H * W * C < ((size_t)1 << 31), "mmk_conv_bwd_fused: tensor too large for 32-bit offsets");
    BwdFusedArgs a = {};
    a.x = (const bf16 *)x; a.g = (const bf16 *)g; a.wpack_t = (const bf16 *)wpack_t; a.dx = (bf16 *)dx; a.scale = scale; a.masked = 1;
    a.B = B; a.H = H; a.W = W; a.partials = partials; a.acc_partials = accumulate;
    return C == 8 ? launch_bwd_fused<8, 8>(a, 8, (hipStream_t)stream) : launch_bwd_fused<16, 16>(a, 16, (hipStream_t)stream);
}

extern "C" int mmk_conv8x16_bwd_fused(const void *x, const void *g, const void *wpack_t, float scale, int32_t B, int32_t H, int32_t W,
                                      void *dx, float *partials, int32_t accumulate, void *stream)
{
    MMK_REQUIRE(x && g && wpack_t && dx && partials, "mmk_conv8x16_bwd_fused: NULL pointer");
    MMK_REQUIRE(B >= 1 && H >= 2 && W >= 2, "mmk_conv8x16_bwd_fused: bad shape");
    MMK_REQUIRE((size_t)B * H * W * 16 < ((size_t)1 << 31), "mmk_conv8x16_bwd_fused: tensor too large for 32-bit offsets");
    BwdFusedArgs a = {};
    a.x = (const bf16 *)x; a.g = (const bf16 *)g; a.wpack_t = (const bf16 *)wpack_t; a.dx = (bf16 *)dx; a.scale = scale; a.masked = 1;
    a.accumulate_dx = 1;
    a.B = B; a.H = H; a.W = W; a.partials = partials; a.acc_partials = accumulate;
    return launch_bwd_fused<8, 16>(a, 8, (hipStream_t)stream);
}

extern "C" int mmk_conv16x8_bwd_fused(const void *x1, const void *x2, const void *g, const void *wpack_t, float scale, int32_t B,
                                      int32_t H, int32_t W, void *dx1, void *dx2, float *partials, int32_t accumulate, void *stream)
{
    MMK_REQUIRE(x1 && g && wpack_t && dx1 && partials, "mmk_conv16x8_bwd_fused: NULL pointer");
    MMK_REQUIRE((x2 == nullptr) == (dx2 == nullptr), "mmk_conv16x8_bwd_fused: x2 and dx2 go together (both NULL: one 16-channel input / unmasked output)");
    MMK_REQUIRE(B >= 1 && H >= 2 && W >= 2, "mmk_conv16x8_bwd_fused: bad shape");
    MMK_REQUIRE((size_t)B * H * W * 16 < ((size_t)1 << 31), "mmk_conv16x8_bwd_fused: tensor too large for 32-bit offsets");
    BwdFusedArgs a = {};
    a.x = (const bf16 *)x1; a.x2 = (const bf16 *)x2; a.g = (const bf16 *)g; a.wpack_t = (const bf16 *)wpack_t;
    a.dx = (bf16 *)dx1; a.dx2 = (bf16 *)dx2; a.scale = scale; a.masked = x2 != nullptr;
    a.B = B; a.H = H; a.W = W; a.partials = partials; a.acc_partials = accumulate;
    return launch_bwd_fused<16, 8>(a, x2 ? 8 : 16, (hipStream_t)stream);
}

extern "C" int32_t mmk_conv3x3_wgrad_slices(int32_t cout, int32_t cin, int32_t c1, int32_t B, int32_t H, int32_t W)
{
    if (B < 1 || H < 1 || W < 1 || !chan_ok(cin) || !chan_ok(cout)) return 0;
    const int ns = wgrad_slices(cout, cin, c1, B, H, W);
    return ns < 0 ? 0 : ns;
}

extern "C" int mmk_conv3x3_wgrad_partial(const void *x1, const void *x2, int32_t C1, int32_t C2, const void *g, int32_t cout,
                                         int32_t B, int32_t H, int32_t W, float *partials, int32_t accumulate, void *stream)
{
    MMK_REQUIRE(x1 && g && partials, "mmk_conv3x3_wgrad_partial: NULL pointer");
    MMK_REQUIRE(B >= 1 && H >= 1 && W >= 1, "mmk_conv3x3_wgrad_partial: bad shape");
    const int cin = C1 + C2;
    MMK_REQUIRE(C1 % 8 == 0 && C2 % 8 == 0 && (C2 == 0 || x2), "mmk_conv3x3_wgrad_partial: bad input split %d+%d", C1, C2);
    MMK_REQUIRE(chan_ok(cin) && chan_ok(cout), "mmk_conv3x3_wgrad_partial: unsupported channel counts %d -> %d", cin, cout);
    WgradArgs a;
    a.x1 = (const bf16 *)x1; a.x2 = (const bf16 *)x2; a.C1 = C1; a.C2 = C2; a.g = (const bf16 *)g;
    a.B = B; a.H = H; a.W = W; a.CIN = cin; a.COUT = cout;
    a.partials = partials; a.acc_partials = accumulate;
    return dispatch_wgrad(a, (hipStream_t)stream);
}

extern "C" int mmk_bn_forward_stats(const void *a, int64_t npix, int32_t C, const float *gamma, const float *beta, float eps,
                                    float momentum, float *running_mean, float *running_var, float *part, float *stat, float *affine,
                                    void *stream)
{
    MMK_REQUIRE(a && gamma && beta && part && stat && affine, "mmk_bn_forward_stats: NULL pointer");
    MMK_REQUIRE(npix >= 1 && chan_ok(C) && C <= 2048, "mmk_bn_forward_stats: bad shape (npix %lld, C %d)", (long long)npix, C);
    MMK_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "mmk_bn_forward_stats: running_mean / running_var go together");
    hipStream_t st = (hipStream_t)stream;
    const int per = 256 / (C / 8);
    const int nblk = (int)std::min<int64_t>(BN_BLOCKS, (npix + per - 1) / per);
    hipLaunchKernelGGL(bn_stats_kernel, dim3(nblk), dim3(256), 0, st, (const bf16 *)a, (size_t)npix, C, part);
    MMK_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 63) / 64), dim3(64), 0, st, part, nblk, C, (double)npix, gamma, beta, eps, momentum,
                       running_mean, running_var, stat, affine);
    MMK_LAUNCH_CHECK();
    return MMK_OK;
}

extern "C" int mmk_bn_apply(const void *a, int64_t npix, int32_t C, const float *affine, float drop_p, uint32_t seed, void *y,
                            void *stream)
{
    MMK_REQUIRE(a && affine && y, "mmk_bn_apply: NULL pointer");
    MMK_REQUIRE(npix >= 1 && chan_ok(C), "mmk_bn_apply: bad shape");
    MMK_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "mmk_bn_apply: dropout probability out of range");
    const size_t ngran = (size_t)npix * (C / 8);
    hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)std::min<size_t>((ngran + 255) / 256, 8192)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16 *)a, ngran, C, affine, drop_p, seed, (bf16 *)y);
    MMK_LAUNCH_CHECK();
    return MMK_OK;
}

extern "C" int mmk_bn_backward(const void *gd, const void *d, float drop_scale, const void *a, int64_t npix, int32_t C,
                               const float *stat, const float *affine, const float *gamma, float leaky_slope, int32_t accumulate,
                               float *part, float *coef, float *dgamma, float *dbeta, void *gz, void *stream)
{
    MMK_REQUIRE(gd && a && coef && gz, "mmk_bn_backward: NULL pointer");
    MMK_REQUIRE(npix >= 1 && chan_ok(C) && C <= 2048, "mmk_bn_backward: bad shape");
    hipStream_t st = (hipStream_t)stream;
    const size_t ngran = (size_t)npix * (C / 8);
    const unsigned gb = (unsigned)std::min<size_t>((ngran + 255) / 256, 8192);
    if (stat != nullptr) {          // training mode: batch statistics
        MMK_REQUIRE(gamma && part && dgamma && dbeta, "mmk_bn_backward: NULL pointer (training mode)");
        const int per = 256 / (C / 8);
        const int nblk = (int)std::min<int64_t>(BN_BLOCKS, (npix + per - 1) / per);
        hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(nblk), dim3(256), 0, st, (const bf16 *)gd, (const bf16 *)d, drop_scale,
                           (const bf16 *)a, (size_t)npix, C, stat, part);
        MMK_LAUNCH_CHECK();
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 63) / 64), dim3(64), 0, st, part, nblk, C, (double)npix, gamma, stat,
                           accumulate, coef, dgamma, dbeta);
        MMK_LAUNCH_CHECK();
    } else {
        MMK_REQUIRE(affine != nullptr, "mmk_bn_backward: evaluation mode needs the affine (scale, shift) pairs");
        hipLaunchKernelGGL(bn_eval_coef_kernel, dim3((C + 63) / 64), dim3(64), 0, st, affine, C, coef);
        MMK_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(gb), dim3(256), 0, st, (const bf16 *)gd, (const bf16 *)d, drop_scale, (const bf16 *)a,
                       ngran, C, stat, coef, leaky_slope, (bf16 *)gz);
    MMK_LAUNCH_CHECK();
    return MMK_OK;
}

static unsigned nblk(size_t n, int t) { return (unsigned)((n + t - 1) / t); }

extern "C" int mmk_channel_minmax(const float *x, int32_t B, int32_t C, int64_t hw, float *part, float *pre, float *minmax,
                                  void *stream)
{
    MMK_REQUIRE(x && part && pre && B >= 1 && C >= 1 && C <= 65535 && hw >= 1, "mmk_channel_minmax: bad argument");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(channel_minmax_partial_kernel, dim3(MM_BLOCKS, C), dim3(256), 0, st, x, B, C, (size_t)hw, part);
    MMK_LAUNCH_CHECK();
    hipLaunchKernelGGL(channel_minmax_final_kernel, dim3(C), dim3(256), 0, st, part, C, pre, minmax);
    MMK_LAUNCH_CHECK();
    return MMK_OK;
}

extern "C" int mmk_conv_first(const float *x, int32_t cin, const float *W, const float *bias, const float *pre, int32_t B,
                              int32_t H, int32_t Wd, float leaky_slope, void *y, void *stream)
{
    MMK_REQUIRE(x && W && y, "mmk_conv_first: NULL pointer");
    MMK_REQUIRE(leaky_slope >= 0.f && leaky_slope < 1.f, "mmk_conv_first: leaky_slope must be in [0, 1)");
    MMK_REQUIRE(cin >= 1 && cin <= 4 && B >= 1 && H >= 1 && Wd >= 1, "mmk_conv_first: bad shape (cin must be 1..4)");
    const size_t npix = (size_t)B * H * Wd;
    if (Wd % 4 == 0 && npix < (1u << 31)) {
        const unsigned blocks = (unsigned)std::min<size_t>((npix / 4 + 255) / 256, 4096);
        if (leaky_slope > 0.f)
            hipLaunchKernelGGL(conv_first_x4_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, cin, W, bias, pre, B, H, Wd,
                               leaky_slope, (bf16 *)y);
        else
            hipLaunchKernelGGL(conv_first_x4_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, cin, W, bias, pre, B, H, Wd,
                               leaky_slope, (bf16 *)y);
        MMK_LAUNCH_CHECK();
        return MMK_OK;
    }
    const unsigned blocks = (unsigned)std::min<size_t>((npix + 255) / 256, 8192);
    hipLaunchKernelGGL(conv_first_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, cin, W, bias, pre, B, H, Wd, leaky_slope,
                       (bf16 *)y);
    MMK_LAUNCH_CHECK();
    return MMK_OK;
}

extern "C" size_t mmk_conv_first_wgrad_ws_bytes(int32_t cin) { return (size_t)1024 * (cin < 1 ? 1 : cin) * 80 * sizeof(float); }

extern "C" int mmk_conv_first_wgrad(const float *x, int32_t cin, const void *g, const float *pre, int32_t B, int32_t H,
                                    int32_t Wd, float *dW, float *db, float *ws, size_t ws_bytes, void *stream)
{
    MMK_REQUIRE(x && g && dW && ws, "mmk_conv_first_wgrad: NULL pointer");
    MMK_REQUIRE(cin >= 1 && cin <= 4 && B >= 1 && H >= 1 && Wd >= 1, "mmk_conv_first_wgrad: bad shape");
    MMK_REQUIRE(ws_bytes >= mmk_conv_first_wgrad_ws_bytes(cin), "mmk_conv_first_wgrad: workspace too small (%zu < %zu bytes)", ws_bytes,
                mmk_conv_first_wgrad_ws_bytes(cin));
    const size_t npix = (size_t)B * H * Wd;
    unsigned blocks;
    if (Wd % 4 == 0 && npix < (1u << 31)) {
        blocks = (unsigned)std::min<size_t>((npix / 4 + 255) / 256, 512);       // (1 024 blocks: 133-138 us instead of 110-114)
        hipLaunchKernelGGL(conv_first_wgrad_x4_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, cin, (const bf16 *)g, pre,
                           B, H, Wd, ws);
    } else {
        blocks = (unsigned)std::min<size_t>((npix + 255) / 256, 1024);
        hipLaunchKernelGGL(conv_first_wgrad_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, cin, (const bf16 *)g, pre, B,
                           H, Wd, ws);
    }
    MMK_LAUNCH_CHECK();
    hipLaunchKernelGGL(conv_first_wgrad_reduce_kernel, dim3(cin), dim3(320), 0, (hipStream_t)stream, ws, (int)blocks, cin, dW, db);
    MMK_LAUNCH_CHECK();
    return MMK_OK;
}

extern "C" int mmk_maxpool2_fwd(const void *x, int32_t B, int32_t H, int32_t W, int32_t C, void *y, void *stream)
{
    MMK_REQUIRE(x && y && B >= 1 && H >= 2 && W >= 2 && C % 8 == 0, "mmk_maxpool2_fwd: bad argument");
    const size_t n = (size_t)B * (H / 2) * (W / 2) * (C / 8);
    hipLaunchKernelGGL(maxpool2_fwd_kernel, dim3(nblk(n, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16 *)x, B, H, W, C,
                       (bf16 *)y, (unsigned *)nullptr);
    MMK_LAUNCH_CHECK();
    return MMK_OK;
}

extern "C" int mmk_maxpool2_fwd_arg(const void *x, int32_t B, int32_t H, int32_t W, int32_t C, void *y, void *arg, void *stream)
{
    MMK_REQUIRE(x && y && arg && B >= 1 && H >= 2 && W >= 2 && C % 8 == 0, "mmk_maxpool2_fwd_arg: bad argument");
    const size_t n = (size_t)B * (H / 2) * (W / 2) * (C / 8);
    hipLaunchKernelGGL(maxpool2_fwd_kernel, dim3(nblk(n, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16 *)x, B, H, W, C,
                       (bf16 *)y, (unsigned *)arg);
    MMK_LAUNCH_CHECK();
    return MMK_OK;
}

extern "C" int mmk_maxpool2_bwd_arg(const void *arg, const void *gy, int32_t B, int32_t H, int32_t W, int32_t C, float scale,
                                    void *gz, void *stream)
{
    MMK_REQUIRE(arg && gy && gz && B >= 1 && H >= 2 && W >= 2 && C % 8 == 0, "mmk_maxpool2_bwd_arg: bad argument");
    const size_t n = (size_t)B * (H / 2) * (W / 2) * (C / 8);
    hipLaunchKernelGGL(maxpool2_bwd_arg_kernel, dim3(nblk(n, 256)), dim3(256), 0, (hipStream_t)stream, (const unsigned *)arg,
                       (const bf16 *)gy, B, H, W, C, scale, (bf16 *)gz);
    MMK_LAUNCH_CHECK();
    return MMK_OK;
}

extern "C" int mmk_maxpool2_bwd(const void *d, const void *gy, int32_t B, int32_t H, int32_t W, int32_t C, float scale,
                                float leaky_slope, void *gz, void *stream)
{
    MMK_REQUIRE(d && gy && gz && B >= 1 && H >= 2 && W >= 2 && C % 8 == 0, "mmk_maxpool2_bwd: bad argument");
    const size_t n = (size_t)B * (H / 2) * (W / 2) * (C / 8);
    hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(nblk(n, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16 *)d,
                       (const bf16 *)gy, B, H, W, C, scale, leaky_slope, (bf16 *)gz);
    MMK_LAUNCH_CHECK();
    return MMK_OK;
}

namespace {
struct UpGeom {
    float rh, rw;
    int lg;          // log2 of the granule count, or -1
};
UpGeom up_geom(int Hs, int Ws, int C, int Ho, int Wo)
{
    UpGeom g;
    g.rh = (Ho > 1) ? (float)(Hs - 1) / (float)(Ho - 1) : 0.f;
    g.rw = (Wo > 1) ? (float)(Ws - 1) / (float)(Wo - 1) : 0.f;
    const int G = C / 8;
    g.lg = -1;
    for (int l = 0; l < 16; ++l)
        if ((1 << l) == G) g.lg = l;
    return g;
}
}  // namespace

extern "C" int mmk_upsample_fwd(const void *x, int32_t B, int32_t Hs, int32_t Ws, int32_t C, int32_t Ho, int32_t Wo, void *y,
                                void *stream)
{
    MMK_REQUIRE(x && y && B >= 1 && Hs >= 1 && Ws >= 1 && Ho >= 1 && Wo >= 1 && C % 8 == 0, "mmk_upsample_fwd: bad argument");
    MMK_REQUIRE(Ho <= 65535 && B <= 65535, "mmk_upsample_fwd: shape exceeds the launch grid");
    const UpGeom ug = up_geom(Hs, Ws, C, Ho, Wo);
    const dim3 grid(nblk((size_t)Wo * (C / 8), 256), (Ho + UP_ROWS - 1) / UP_ROWS, B);
    if (ug.lg >= 0)
        hipLaunchKernelGGL(upsample_fwd_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16 *)x, Hs, Ws, C, Ho, Wo,
                           ug.rh, ug.rw, ug.lg, (bf16 *)y);
    else
        hipLaunchKernelGGL(upsample_fwd_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16 *)x, Hs, Ws, C, Ho, Wo,
                           ug.rh, ug.rw, 0, (bf16 *)y);
    MMK_LAUNCH_CHECK();
    return MMK_OK;
}

extern "C" int mmk_upsample_bwd(const void *gy, int32_t B, int32_t Hs, int32_t Ws, int32_t C, int32_t Ho, int32_t Wo,
                                const void *relu_src, float scale, float leaky_slope, void *gx, void *stream)
{
    MMK_REQUIRE(gy && gx && B >= 1 && Hs >= 1 && Ws >= 1 && Ho >= 1 && Wo >= 1 && C % 8 == 0, "mmk_upsample_bwd: bad argument");
    MMK_REQUIRE(Hs <= 65535 && B <= 65535, "mmk_upsample_bwd: shape exceeds the launch grid");
    const UpGeom ug = up_geom(Hs, Ws, C, Ho, Wo);
    const dim3 grid(nblk((size_t)Ws * (C / 8), 256), (Hs + UP_ROWS - 1) / UP_ROWS, B);
#define MMK_UPB(P2, LKV, LG)                                                                                                   \
    hipLaunchKernelGGL((upsample_bwd_kernel<P2, LKV>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16 *)gy, Hs, Ws, C, Ho, Wo, \
                       ug.rh, ug.rw, LG, (const bf16 *)relu_src, scale, leaky_slope, (bf16 *)gx)
    const bool lkv = leaky_slope > 0.f;
    if (ug.lg >= 0) {
        if (lkv) MMK_UPB(true, true, ug.lg); else MMK_UPB(true, false, ug.lg);
    } else {
        if (lkv) MMK_UPB(false, true, 0); else MMK_UPB(false, false, 0);
    }
#undef MMK_UPB
    MMK_LAUNCH_CHECK();
    return MMK_OK;
}

extern "C" int mmk_final_fwd(const void *x, const float *w, const float *bias, int64_t npix, float *mask, void *stream)
{
    MMK_REQUIRE(x && w && bias && mask && npix >= 1, "mmk_final_fwd: bad argument");
    hipLaunchKernelGGL(final_fwd_kernel, dim3(nblk((size_t)npix, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16 *)x, w,
                       bias, (size_t)npix, mask);
    MMK_LAUNCH_CHECK();
    return MMK_OK;
}

extern "C" int mmk_final_bwd(const void *x, const float *w, const float *mask, const float *gmask, int64_t npix, float scale,
                             float leaky_slope, void *gx, float *dW, float *db, float *ws /* MMK_FINAL_BWD_WS_FLOATS */, void *stream)
{
    MMK_REQUIRE(x && w && mask && gmask && gx && dW && db && ws && npix >= 1, "mmk_final_bwd: bad argument");
    const unsigned blocks = (unsigned)std::min<size_t>(((size_t)npix + 255) / 256, 512);
    if (leaky_slope > 0.f)
        hipLaunchKernelGGL(final_bwd_kernel<true>, dim3(blocks, 1), dim3(256), 0, (hipStream_t)stream, (const bf16 *)x, w, mask, gmask,
                           (size_t)npix, (const float *)nullptr, scale, leaky_slope, (bf16 *)gx, ws);
    else
        hipLaunchKernelGGL(final_bwd_kernel<false>, dim3(blocks, 1), dim3(256), 0, (hipStream_t)stream, (const bf16 *)x, w, mask, gmask,
                           (size_t)npix, (const float *)nullptr, scale, leaky_slope, (bf16 *)gx, ws);
    MMK_LAUNCH_CHECK();
    hipLaunchKernelGGL(final_bwd_reduce_kernel, dim3(1), dim3(288), 0, (hipStream_t)stream, ws, (int)blocks, dW, db);
    MMK_LAUNCH_CHECK();
    return MMK_OK;
}

extern "C" int mmk_mask_normalize(const float *mask, int32_t B, int64_t npix_per, float *part, float *mask_n, float *amax,
                                  void *stream)
{
    MMK_REQUIRE(mask && part && mask_n && amax && B >= 1 && B <= 65535 && npix_per >= 1, "mmk_mask_normalize: bad argument");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(mask_segmax_kernel, dim3(MASK_SEG, B), dim3(256), 0, st, mask, (size_t)npix_per, part);
    MMK_LAUNCH_CHECK();
    const unsigned bx = (unsigned)std::min<size_t>(((size_t)npix_per + 1023) / 1024, 256);
    hipLaunchKernelGGL(mask_scale_kernel, dim3(bx, B), dim3(256), 0, st, mask, (size_t)npix_per, part, mask_n, amax);
    MMK_LAUNCH_CHECK();
    return MMK_OK;
}

extern "C" int mmk_final_bwd_normalized(const void *x, const float *w, const float *mask, const float *mask_n, const float *amax,
                                        const float *gmask_n, int32_t B, int64_t npix_per, float scale, float leaky_slope,
                                        float *part, float *coef, void *gx, float *dW, float *db,
                                        float *ws /* MMK_FINAL_BWD_WS_FLOATS */, void *stream)
{
    MMK_REQUIRE(x && w && mask && mask_n && amax && gmask_n && part && coef && gx && dW && db && ws, "mmk_final_bwd_normalized: NULL pointer");
    MMK_REQUIRE(B >= 1 && B <= 65535 && npix_per >= 1, "mmk_final_bwd_normalized: bad shape");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(mask_norm_bwd_partial_kernel, dim3(MASK_SEG, B), dim3(256), 0, st, gmask_n, mask_n, (size_t)npix_per, part);
    MMK_LAUNCH_CHECK();
    hipLaunchKernelGGL(mask_norm_bwd_final_kernel, dim3((B + 63) / 64), dim3(64), 0, st, part, amax, B, coef);
    MMK_LAUNCH_CHECK();
    // ~512 blocks in all (9 partial sums per block; per * B <= 512 + B <= MMK_FINAL_BWD_WS_FLOATS / 9 for B <= 1200)
    const unsigned per = (unsigned)std::max<size_t>(1, std::min<size_t>(((size_t)npix_per + 255) / 256, (512 + B - 1) / B));
    MMK_REQUIRE((size_t)per * B * 9 <= MMK_FINAL_BWD_WS_FLOATS, "mmk_final_bwd_normalized: batch too large for the reduction workspace");
    if (leaky_slope > 0.f)
        hipLaunchKernelGGL(final_bwd_kernel<true>, dim3(per, B), dim3(256), 0, st, (const bf16 *)x, w, mask, gmask_n, (size_t)npix_per, coef,
                           scale, leaky_slope, (bf16 *)gx, ws);
    else
        hipLaunchKernelGGL(final_bwd_kernel<false>, dim3(per, B), dim3(256), 0, st, (const bf16 *)x, w, mask, gmask_n, (size_t)npix_per, coef,
                           scale, leaky_slope, (bf16 *)gx, ws);
    MMK_LAUNCH_CHECK();
    hipLaunchKernelGGL(final_bwd_reduce_kernel, dim3(1), dim3(288), 0, st, ws, (int)(per * B), dW, db);
    MMK_LAUNCH_CHECK();
    return MMK_OK;
}
