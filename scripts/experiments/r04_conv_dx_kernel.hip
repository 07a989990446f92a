// EXPERIMENT, NOT PART OF THE BUILD (round 4).  Kept as the record of the LDS-DMA rebuild of conv3x3_deep_kernel; HISTORY.md section 10
// has the measurements: bit-identical to the deep kernel at p = 0, within -10 % ... +9 % of its time (it wins at 160 x 160 forward only),
// because what bounds these launches is not the staging path.  To build it again: add it to _lib.SOURCES, declare
// conv_dx_dispatch() / dropout_draws4_fast() (= dropout_draws4) in mmk_unet_shared.h and call it from dispatch_conv_deep().
// 3x3 convolution of the >= 64-channel U-Net levels (forward and data-gradient roles), second generation:
// the same implicit GEMM as conv3x3_deep_kernel (mmk_unet.hip: weights = MFMA A operand, 16 pixels of a row = B operand,
// v_mfma_f32_16x16x32_bf16, packed weights in fragment order, 8 waves per block, input channels in chunks of 32), with
// the staging rebuilt around LDS-DMA (global_load_lds_dwordx4) so that nothing of it runs on the vector pipe or through
// registers and every load stays in flight across the workgroup barriers:
//
//   * stage = (tile, 32-channel chunk), cut into three SUB-STAGES, one per tap row.  The weights of a sub-stage (3 taps x
//     BM output channels x 32 input channels, 12 / 24 KB) sit in a ring of three LDS slots and are issued two sub-stages
//     ahead; the chunk's halo tile (64 bytes per pixel) is double buffered and issued one stage ahead, in the first two
//     sub-stages of the stage before.  One raw s_barrier per sub-stage, preceded by a COUNTED s_waitcnt vmcnt(N) that
//     retires exactly the pieces the next sub-stage reads (N = the pieces issued since, a compile-time constant: every
//     wave issues its pieces in the same fixed order; waves that own one piece less use the same N, which waits for more,
//     never for less).  A piece is read one barrier after the wait that retired it (cdna_hip_programming.md §5).
//   * LDS-DMA writes 1 KB per wave-instruction at "wave-uniform base + 16 bytes x lane": the halo image is therefore
//     unpadded (64 bytes per pixel) and bank conflicts of the ds_read_b128 fragment reads are avoided by a swizzle instead:
//     16-byte granule g of halo pixel q sits at cell 4q + (g ^ 2 ((q >> 2) & 1)); the DMA's per-lane SOURCE address
//     applies the same involution.  (Enumerated over the real lane groups of ds_read_b128 for every alignment of a
//     16-pixel fragment: conflict-free.)  The packed weights are copied as they are (lane-linear fragments).
//   * out-of-image granules read a 16-byte zero word; nothing in the loop is conditional on data.
//
// Roles, arguments and results are those of conv3x3_deep_kernel (the epilogue is the same code, with unconditional
// stores): outputs are bit-identical to it (tests/test_gpu_unet_kernels.py::test_conv_dx_bit_identical_to_deep).
// Reference: the nn.Conv2d 3x3 layers of mm_masking/icp_weight_policy.py:104-125 at >= 64 channels and their backward.
#include <stdlib.h>

#include "mmk_unet_shared.h"

namespace {

using namespace mmku;

__device__ u32x4 dx_zero16;     // zero-initialised, never written
__device__ u32x4 dx_sink16[8];  // write-only: where lanes without an output pixel store

constexpr int DX_THREADS = 512;

// pieces w, w + 8, w + 16, ... < total that wave w owns among k0 <= k < k1
constexpr int dx_owned(int w, int total, int k0, int k1)
{
    int n = 0;
    for (int k = k0; k < k1; ++k)
        if (w + 8 * k < total) ++n;
    return n;
}

template <int BM, int NT>
struct DxCfg {
    static constexpr int WM = BM >= 128 ? 2 : 1;          // waves along the output channels
    static constexpr int WN = 8 / WM;                     // waves along the tile rows
    static constexpr int MT = BM / 16 / WM;               // 16-channel tiles per wave
    static constexpr int MTB = BM / 16;                   // ... per block
    static constexpr int TH = WN, TWD = NT * 16;
    static constexpr int HT = TH + 2, WT = TWD + 2;
    static constexpr int NPIX = HT * WT;
    static constexpr int NPI = (NPIX * 4 + 63) / 64;      // 1-KB pieces of a halo buffer
    static constexpr int PIXB = NPI * 1024;
    static constexpr int NWI = 3 * MTB;                   // 1-KB pieces of a weight sub-stage (one tap row)
    static constexpr int WSUB = NWI * 1024;
    static constexpr int PW = (NPI + 7) / 8;              // halo pieces per wave and stage (pieces w, w + 8, ...: beyond NPI dummies)
    static constexpr int PW0 = (PW + 1) / 2, PW1 = PW - PW0;     // ... issued in sub-stage 0 / sub-stage 1
    static constexpr int WW = (NWI + 7) / 8;              // weight pieces per wave and sub-stage (beyond NWI: dummies)
    static constexpr int RING = 4;                        // weight slots: a sub-stage's weights are issued three sub-stages ahead
    static constexpr int OFF_W = 2 * PIXB;
    static constexpr int OFF_BIAS = OFF_W + RING * WSUB;
    static constexpr int OFF_DUMMY = OFF_BIAS + BM * 4;   // 1 KB: where the dummy pieces land (every wave issues the same count)
    static constexpr size_t SMEM = (size_t)OFF_DUMMY + 1024;
    static constexpr int ST = NT * 2;                     // 16-byte stores per lane of the epilogue
    static_assert(MT == 4, "conv3x3_dx_kernel: 64 output channels per wave");
    static_assert(SMEM <= 160 * 1024, "LDS");
};

template <int N>
__device__ __forceinline__ void dx_wait_vm()
{
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ void dx_barrier()
{
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ void dx_dma16(const void *gsrc, unsigned char *lds_piece)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                     (__attribute__((address_space(3))) void *)lds_piece, 16, 0, 0);
}

// ABL (development ablations, MMK_DX_ABL): bit 0 = no DMA after the prologue, bit 1 = no fragment reads / MFMAs, bit 2 = no epilogue
template <int BM, int NT, int ABL = 0>
__global__ __launch_bounds__(DX_THREADS) void conv3x3_dx_kernel(const ConvArgs a, int total_tiles, int tiles_per_xcd)
{
    using C = DxCfg<BM, NT>;
    constexpr int MT = C::MT, MTB = C::MTB, WT = C::WT, PW = C::PW, PW0 = C::PW0, WW = C::WW;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv / C::WN, wn = wv % C::WN;
    const int tiles_x = (a.W + C::TWD - 1) / C::TWD, tiles_y = (a.H + C::TH - 1) / C::TH;
    const int tpi = tiles_x * tiles_y;
    const int group = blockIdx.y;
    const int nchunk = a.CIN / 32;

    const int xcd = blockIdx.x & 7, nb = gridDim.x >> 3;
    const int t_begin = xcd * tiles_per_xcd;
    const int t_end = (t_begin + tiles_per_xcd < total_tiles) ? t_begin + tiles_per_xcd : total_tiles;
    int tile = t_begin + (blockIdx.x >> 3);
    if (tile >= t_end) return;

    // ---- this lane's granule of each halo piece the wave owns: (row, column) inside the halo tile, pixel offset from the
    // halo origin, channel offset inside the chunk (the swizzle's involution applied to the cell the lane writes)
    bool st_abl = false;      // (ablation: set once the prologue is issued)
    int p_rc[PW], p_off[PW];
#pragma unroll
    for (int k = 0; k < PW; ++k) {
        const int G = (wv + 8 * k) * 64 + lane;            // cell of the halo image
        const int q = G >> 2;
        const int g = (G & 3) ^ (((q >> 2) & 1) << 1);
        const bool valid = q < C::NPIX;
        const int r = valid ? q / WT : 0x7ff, c = q % WT; // (an invalid row fails every bounds test)
        p_rc[k] = (r << 20) | (c << 8) | (g * 8);
        p_off[k] = r * a.W + c;
    }
    // halo pieces of stage (t, chunk) with first <= k < last, into the halo buffer at byte offset `buf`
    auto issue_pix = [&](int t, int chunk, int buf, int first, int last, int phase = -1) {
        const int b = t / tpi, tr = t - b * tpi;
        const int tyi = tr / tiles_x;
        const int tx0 = (tr - tyi * tiles_x) * C::TWD, ty0 = tyi * C::TH;
        const int c0 = chunk * 32;
        const bool in1 = c0 < a.C1;                        // a 32-channel chunk lies entirely in one of the two concatenated inputs
        const bf16 *xb = in1 ? a.x1 : a.x2;
        const int xc = in1 ? a.C1 : a.C2, cb = in1 ? c0 : c0 - a.C1;
        const long org = ((long)b * a.H + ty0 - 1) * a.W + tx0 - 1;
        const bf16 *base = xb + org * xc + cb;
        if ((ABL & 1) && buf >= 0 && st_abl) return;
#pragma unroll
        for (int k = 0; k < PW; ++k) {
            if (k < first || k >= last) continue;
            if (phase >= 0 && (k - first) % 3 != phase) continue;
            if (wv + 8 * k >= C::NPI) {                    // (wave-uniform) a dummy piece: every wave issues the same number of
                dx_dma16(&dx_zero16, smem + C::OFF_DUMMY); // vector-memory operations per sub-stage, so the counted waits are exact
                continue;
            }
            const int r = p_rc[k] >> 20, c = (p_rc[k] >> 8) & 0xfff, gch = p_rc[k] & 0xff;
            const bool ok = (unsigned)(ty0 - 1 + r) < (unsigned)a.H && (unsigned)(tx0 - 1 + c) < (unsigned)a.W;
            const unsigned off = (unsigned)(p_off[k] * xc + gch);
            const void *sp = ok ? static_cast<const void *>(base + off) : static_cast<const void *>(&dx_zero16);
            dx_dma16(sp, smem + buf + (wv + 8 * k) * 1024);
        }
    };
    // weights of (chunk, tap row) into ring slot `slot`
    auto issue_w = [&](int chunk, int taprow, int slot, int phase = -1) {
        if ((ABL & 1) && st_abl) return;
        const bf16 *wsrc = a.wpack + ((size_t)(group * nchunk + chunk) * 9 + taprow * 3) * MTB * 512;
#pragma unroll
        for (int k = 0; k < WW; ++k) {
            if (phase >= 0 && k % 3 != phase) continue;
            if (wv + 8 * k >= C::NWI) {
                dx_dma16(&dx_zero16, smem + C::OFF_DUMMY);
                continue;
            }
            dx_dma16(wsrc + (size_t)(wv + 8 * k) * 512 + lane * 8, smem + C::OFF_W + slot * C::WSUB + (wv + 8 * k) * 1024);
        }
    };

    float *bias_lds = reinterpret_cast<float *>(smem + C::OFF_BIAS);
    if (tid < BM) {
        const int c = group * BM + tid;
        bias_lds[tid] = (a.bias && c < a.COUT) ? a.bias[c] : 0.f;
    }
    f32x4 acc[MT][NT];
    auto reset_acc = [&]() {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    reset_acc();
    const DropoutParams dp = dropout_params(a.drop_p);
    const bool lean = a.o1.relu_src == nullptr && a.o2.relu_src == nullptr && a.o1.accumulate == 0 && a.o2.accumulate == 0;

    // ---- fragment addresses.  B: lane l reads granule l / 16 of halo pixel q = (wn + ty) WT + 16 n + tx + (l & 15); the
    // swizzle bit of q depends on (q_lane + delta) with delta = ty WT + 16 n + tx known at compile time: one base per
    // class delta % 8, everything else in the read's immediate offset.
    unsigned bq[8];
    {
        const int ql = wn * WT + (lane & 15), g = lane >> 4;
#pragma unroll
        for (int cl = 0; cl < 8; ++cl) bq[cl] = (unsigned)(ql * 64 + ((g ^ ((((ql + cl) >> 2) & 1) << 1)) * 16));
    }
    const unsigned a_off = (unsigned)(C::OFF_W + ((wm * MT) * 64 + lane) * 16);

    int chunk = 0, st = 0;
    bool after_epilogue = false;
    constexpr bool SPREAD = (ABL & 8) == 0;   // (ABL bit 3: the block-wise issue orders of the first cut, for comparison)
    const bool ga = wv < 4;         // waves 0-3: issue, then compute; waves 4-7 (their SIMD partners): compute, then issue
    // ---- prologue: weights of sub-stages 0 .. 2, the first stage's halo tile
    issue_w(0, 0, 0);
    issue_w(0, 1, 1);
    issue_w(0, 2, 2);
    issue_pix(tile, 0, 0, 0, PW);
    dx_wait_vm<0>();
    __syncthreads();
    st_abl = true;

    while (true) {
        int ntile = tile, nck = chunk + 1;
        if (nck == nchunk) {
            nck = 0;
            ntile = tile + nb;
        }
        const bool has_next = ntile < t_end;
        const int lt = has_next ? ntile : tile;            // (clamped: the DMA schedule stays unconditional)
        const int lck = has_next ? nck : chunk;
        const int cur = (st & 1) * C::PIXB, oth = C::PIXB - cur;
        const bool tile_end = chunk == nchunk - 1;
        const int ubase = (3 * st) & 3;                    // ring slot of this stage's first sub-stage
        unsigned bcur[8];
#pragma unroll
        for (int cl = 0; cl < 8; ++cl) bcur[cl] = bq[cl] + (unsigned)cur;

#define DX_ISSUE(TR)                                                                                                 \
    {                                                                                                                \
        /* weights three sub-stages ahead (= the next stage's same tap row) into the slot the previous sub-stage read, */ \
        /* then the next stage's halo tile (sub-stages 0 and 1) */                                                   \
        issue_w(lck, TR, (ubase + (TR) + 3) & 3);                                                                     \
        if (TR == 0) issue_pix(lt, lck, oth, 0, PW0);                                                                \
        if (TR == 1) issue_pix(lt, lck, oth, PW0, PW);                                                               \
    }
#define DX_MFMA(TR)                                                                                                  \
    if (!(ABL & 2)) {                                                                                                \
        const unsigned a_cur = a_off + (unsigned)(((ubase + (TR)) & 3) * C::WSUB);                                   \
        _Pragma("unroll") for (int tl = 0; tl < 3; ++tl) {                                                           \
            bf16x8 bf[NT];                                                                                           \
            _Pragma("unroll") for (int n = 0; n < NT; ++n) {                                                         \
                const int delta = (TR) * WT + n * 16 + tl;                                                           \
                bf[n] = *reinterpret_cast<const bf16x8 *>(smem + bcur[delta & 7] + delta * 64);                      \
            }                                                                                                        \
            _Pragma("unroll") for (int m = 0; m < MT; ++m) {                                                         \
                const bf16x8 af = *reinterpret_cast<const bf16x8 *>(smem + a_cur + (tl * MTB + m) * 1024);           \
                _Pragma("unroll") for (int n = 0; n < NT; ++n)                                                       \
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf[n], acc[m][n], 0, 0, 0);              \
            }                                                                                                        \
            if (SPREAD) {   /* a third of the sub-stage's pieces behind each tap: the memory pipe's queue never backs up */ \
                __builtin_amdgcn_sched_barrier(0);                                                                   \
                issue_w(lck, TR, (ubase + (TR) + 3) & 3, tl);                                                        \
                if (TR == 0) issue_pix(lt, lck, oth, 0, PW0, tl);                                                    \
                if (TR == 1) issue_pix(lt, lck, oth, PW0, PW, tl);                                                   \
                __builtin_amdgcn_sched_barrier(0);                                                                   \
            }                                                                                                        \
        }                                                                                                            \
    }
        // The two waves of a SIMD (w and w + 4) run the sub-stage in opposite orders: waves 0-3 issue their DMA pieces first and
        // then compute, waves 4-7 compute first and issue last -- one wave's issue block (~100 cycles per piece, plus its address
        // arithmetic) and, at a tile's end, its epilogue run beside the partner's MFMAs instead of beside the partner's own
        // issue block / epilogue (in lockstep the three phases simply added up: scripts/abl_dx.py).  The counted waits
        // differ accordingly: for waves 4-7 the pieces of the CURRENT sub-stage are the youngest operations.
        // ---- sub-stage 0: the weights of (st, 1) are needed next
        if (!SPREAD && ga) DX_ISSUE(0)
        DX_MFMA(0)
        if (!SPREAD && !ga) DX_ISSUE(0)
        // (younger than the weights of the next sub-stage, issued two sub-stages ago: the halo pieces of that sub-stage, the two
        // issue blocks since, and the stores of an epilogue in between)
        if (after_epilogue && !(ABL & 4)) dx_wait_vm<2 * WW + PW + C::ST>();
        else dx_wait_vm<2 * WW + PW>();
        dx_barrier();
        // ---- sub-stage 1: the weights of (st, 2) are needed next
        if (!SPREAD && ga) DX_ISSUE(1)
        DX_MFMA(1)
        if (!SPREAD && !ga) DX_ISSUE(1)
        if ((ga || SPREAD) && after_epilogue && !(ABL & 4)) dx_wait_vm<2 * WW + PW + C::ST>();
        else dx_wait_vm<2 * WW + PW>();
        dx_barrier();
        // ---- sub-stage 2: the next stage's halo tile and the weights of (st + 1, 0) are needed next
        if (!SPREAD && ga) DX_ISSUE(2)
        DX_MFMA(2)
        if (tile_end && (ABL & 4)) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n) asm volatile("" ::"v"(acc[m][n]));
            reset_acc();
        } else if (tile_end && lean) {
            // ---- forward role (no ReLU-backward source, no accumulate target): bias and 1 / keep in one fma per value
            // (exact for p = 0: fma(acc, 1, bias) = acc + bias, bit-identical to conv3x3_deep_kernel there), ReLU as one max
            // against 0 or -inf, the dropout draws from the 24-bit-multiply hash, one select per value
            const int b = tile / tpi, tr = tile - b * tpi;
            const int tyi = tr / tiles_x;
            const int tx0 = (tr - tyi * tiles_x) * C::TWD, yy = tyi * C::TH + wn;
            int lv = lane;
            asm volatile("" : "+v"(lv));
            const int g4 = lv >> 4;
            const int lane_ch = 16 * (g4 & 1) + 8 * (g4 >> 1);
            bf16 *t_y[2];
            int t_C[2], t_cl[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int ct = group * BM + wm * 64 + 32 * c + lane_ch;
                const bool fpt = ct < a.o1.C;
                t_y[c] = fpt ? a.o1.y : a.o2.y;
                t_C[c] = fpt ? a.o1.C : a.o2.C;
                t_cl[c] = fpt ? ct : ct - a.o1.C;
            }
            auto swap16 = [](unsigned (&x)[4]) {
                auto r01 = __builtin_amdgcn_permlane16_swap(x[0], x[1], false, false);
                auto r23 = __builtin_amdgcn_permlane16_swap(x[2], x[3], false, false);
                x[0] = r01[0]; x[1] = r01[1]; x[2] = r23[0]; x[3] = r23[1];
            };
            const bool drop = a.drop_p > 0.f;
            const float ik = dp.inv_keep;
            const float floor_v = a.relu ? 0.f : -INFINITY;
            f32x4 bvk[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const f32x4 bv = *reinterpret_cast<const f32x4 *>(bias_lds + (wm * 4 + m) * 16 + g4 * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) bvk[m][r] = bv[r] * ik;
            }
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int xx = tx0 + n * 16 + (lv & 15);
                const bool okp = yy < a.H && xx < a.W;
                const size_t p = ((size_t)b * a.H + yy) * a.W + xx;
                unsigned lo[4], hi[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaxf(__builtin_fmaf(acc[m][n][r], ik, bvk[m][r]), floor_v);
                    if (drop) {
                        unsigned dr[4];
                        const int c0 = group * BM + (wm * 4 + m) * 16 + g4 * 4;
                        dropout_draws4_fast(a.seed, (unsigned)(p * a.COUT + c0), dr);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = (dr[r] >= dp.thr) ? v[r] : 0.f;
                    }
                    bf16x4 outv;
#pragma unroll
                    for (int r = 0; r < 4; ++r) outv[r] = (bf16)v[r];
                    const unsigned long long pk = __builtin_bit_cast(unsigned long long, outv);
                    lo[m] = (unsigned)pk;
                    hi[m] = (unsigned)(pk >> 32);
                }
                swap16(lo);
                swap16(hi);
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const u32x4 w = {lo[2 * c], hi[2 * c], lo[2 * c + 1], hi[2 * c + 1]};
                    u32x4 *dst = okp ? reinterpret_cast<u32x4 *>(t_y[c] + p * t_C[c] + t_cl[c]) : &dx_sink16[lv & 7];
                    *dst = w;
                }
            }
            reset_acc();
        } else if (tile_end) {
            const int b = tile / tpi, tr = tile - b * tpi;
            const int tyi = tr / tiles_x;
            const int tx0 = (tr - tyi * tiles_x) * C::TWD, yy = tyi * C::TH + wn;
            // The wave's four 16-channel tiles: the MFMA layout gives lane group g = 2a+b (16 lanes each) 4 channels of each
            // tile m = 2c+d, i.e. four 8-byte pieces 32 bytes apart.  One v_permlane16_swap per dword and tile pair trades lane
            // bit b for tile bit d; the lane then owns 16 contiguous bytes at channel 32c + 16b + 8a and one store instruction
            // covers 64 contiguous bytes per pixel (conv3x3_deep_kernel's wide path; same arithmetic, same order).
            int lv = lane;
            asm volatile("" : "+v"(lv));
            const int g4 = lv >> 4;
            const int lane_ch = 16 * (g4 & 1) + 8 * (g4 >> 1);
            bf16 *t_y[2];
            const bf16 *t_src[2];
            int t_C[2], t_cl[2];
            bool t_acc[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int ct = group * BM + wm * 64 + 32 * c + lane_ch;
                const bool fpt = ct < a.o1.C;
                t_y[c] = fpt ? a.o1.y : a.o2.y;
                t_src[c] = fpt ? a.o1.relu_src : a.o2.relu_src;
                t_C[c] = fpt ? a.o1.C : a.o2.C;
                t_acc[c] = (fpt ? a.o1.accumulate : a.o2.accumulate) != 0;
                t_cl[c] = fpt ? ct : ct - a.o1.C;
            }
            bool m_has_src[4];
            float m_scale[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int c0 = group * BM + (wm * 4 + m) * 16 + g4 * 4;
                const bool firstp = c0 < a.o1.C;
                m_has_src[m] = (firstp ? a.o1.relu_src : a.o2.relu_src) != nullptr;
                m_scale[m] = firstp ? a.o1.scale : a.o2.scale;
            }
            auto swap16 = [](unsigned (&x)[4]) {
                auto r01 = __builtin_amdgcn_permlane16_swap(x[0], x[1], false, false);
                auto r23 = __builtin_amdgcn_permlane16_swap(x[2], x[3], false, false);
                x[0] = r01[0]; x[1] = r01[1]; x[2] = r23[0]; x[3] = r23[1];
            };
            const bool any_src = a.o1.relu_src != nullptr || a.o2.relu_src != nullptr;
            const bool any_acc = a.o1.accumulate != 0 || a.o2.accumulate != 0;
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int xx = tx0 + n * 16 + (lv & 15);
                const bool okp = yy < a.H && xx < a.W;
                const size_t p = ((size_t)b * a.H + yy) * a.W + xx;
                unsigned slo[4] = {0, 0, 0, 0}, shi[4] = {0, 0, 0, 0}, alo[4] = {0, 0, 0, 0}, ahi[4] = {0, 0, 0, 0};
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    if (t_src[c] != nullptr && okp) {
                        const u32x4 v = *reinterpret_cast<const u32x4 *>(t_src[c] + p * t_C[c] + t_cl[c]);
                        slo[2 * c] = v[0]; shi[2 * c] = v[1]; slo[2 * c + 1] = v[2]; shi[2 * c + 1] = v[3];
                    }
                    if (t_acc[c] && okp) {
                        const u32x4 v = *reinterpret_cast<const u32x4 *>(t_y[c] + p * t_C[c] + t_cl[c]);
                        alo[2 * c] = v[0]; ahi[2 * c] = v[1]; alo[2 * c + 1] = v[2]; ahi[2 * c + 1] = v[3];
                    }
                }
                if (any_src) { swap16(slo); swap16(shi); }
                if (any_acc) { swap16(alo); swap16(ahi); }
                unsigned lo[4], hi[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    float v[4];
                    const f32x4 bv = *reinterpret_cast<const f32x4 *>(bias_lds + (wm * 4 + m) * 16 + g4 * 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[r] = acc[m][n][r] + bv[r];
                        if (a.relu) v[r] = fmaxf(v[r], 0.f);
                    }
                    if (a.drop_p > 0.f) {
                        float sc[4];
                        const int c0 = group * BM + (wm * 4 + m) * 16 + g4 * 4;
                        dropout_scale4(a.seed, (unsigned)(p * a.COUT + c0), dp, sc);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = v[r] * sc[r];
                    }
                    if (m_has_src[m]) {
                        const bf16x4 sv = __builtin_bit_cast(bf16x4, (unsigned long long)slo[m] | ((unsigned long long)shi[m] << 32));
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = ((float)sv[r] > 0.f) ? v[r] * m_scale[m] : 0.f;
                    }
                    if (any_acc) {
                        const bf16x4 ov = __builtin_bit_cast(bf16x4, (unsigned long long)alo[m] | ((unsigned long long)ahi[m] << 32));
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += (float)ov[r];
                    }
                    bf16x4 outv;
#pragma unroll
                    for (int r = 0; r < 4; ++r) outv[r] = (bf16)v[r];
                    const unsigned long long pk = __builtin_bit_cast(unsigned long long, outv);
                    lo[m] = (unsigned)pk;
                    hi[m] = (unsigned)(pk >> 32);
                }
                swap16(lo);
                swap16(hi);
                // unconditional stores (a lane without a pixel writes the sink): the count of vector-memory operations per
                // sub-stage is a compile-time constant, which the counted waits rely on
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const u32x4 w = {lo[2 * c], hi[2 * c], lo[2 * c + 1], hi[2 * c + 1]};
                    u32x4 *dst = okp ? reinterpret_cast<u32x4 *>(t_y[c] + p * t_C[c] + t_cl[c]) : &dx_sink16[lv & 7];
                    *dst = w;
                }
            }
            reset_acc();
        }
        if (!SPREAD && !ga) DX_ISSUE(2)
        // (both orders: the stores of the epilogue and this sub-stage's weight pieces are younger than the halo pieces of
        // sub-stage 1, the youngest thing the next sub-stage reads)
        if (tile_end && !(ABL & 4)) dx_wait_vm<WW + C::ST>();
        else dx_wait_vm<WW>();
        if (!has_next) break;
        dx_barrier();
        after_epilogue = tile_end;
        tile = ntile;
        chunk = nck;
        ++st;
    }
#undef DX_ISSUE
#undef DX_MFMA
    dx_wait_vm<0>();       // (the clamped pieces issued by the last sub-stages)
}

template <int BM, int NT, int ABL = 0>
int launch_conv_dx(const ConvArgs &a, hipStream_t st)
{
    using C = DxCfg<BM, NT>;
    static bool attr_set[64] = {};
    int dev = 0;
    MMK_CHECK_HIP(hipGetDevice(&dev));
    if (!attr_set[dev & 63]) {
        MMK_CHECK_HIP(hipFuncSetAttribute((const void *)conv3x3_dx_kernel<BM, NT, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::SMEM));
        attr_set[dev & 63] = true;
    }
    const int tiles = ((a.W + C::TWD - 1) / C::TWD) * ((a.H + C::TH - 1) / C::TH);
    const int groups = (a.COUT + BM - 1) / BM;
    const int total = tiles * a.B;
    const int per_xcd = (total + 7) / 8;
    int nb = 32 / groups;                                    // one 8-wave block per CU, 32 CUs per XCD
    nb = nb < 1 ? 1 : (nb > per_xcd ? per_xcd : nb);
    hipLaunchKernelGGL((conv3x3_dx_kernel<BM, NT, ABL>), dim3(8 * nb, groups), dim3(DX_THREADS), C::SMEM, st, a, total, per_xcd);
    MMK_LAUNCH_CHECK();
    return MMK_OK;
}

}  // namespace

namespace mmku {

int conv_dx_dispatch(const ConvArgs &a, hipStream_t st)
{
    // the ReLU network's layers with a whole number of 64-channel wave tiles on the output side
    if (a.slope > 0.f || a.pool_y != nullptr || a.CIN % 32 != 0 || a.COUT % 64 != 0 || a.C1 % 32 != 0 || a.C2 % 32 != 0) return 1;
    if (a.o1.C % 64 != 0 || a.o2.C % 64 != 0) return 1;
    if ((size_t)a.B * a.H * a.W * (size_t)(a.CIN > a.COUT ? a.CIN : a.COUT) >= ((size_t)1 << 31)) return 1;
    {
        const char *e = getenv("MMK_CONV_DX");          // MMK_CONV_DX=0: conv3x3_deep_kernel everywhere (A/B, bit-identity test)
        if (e && e[0] == '0') return 1;
    }
    const bool narrow = a.W <= 48;
    const bool wide_bm = a.COUT >= 128;
#ifdef MMK_DX_ABLATIONS
    {
        const char *e = getenv("MMK_DX_ABL");
        const int abl = e ? atoi(e) : 0;
#define DX_ABL_CASE(A)                                                                                         \
    if (abl == A) {                                                                                            \
        if (wide_bm) return narrow ? launch_conv_dx<128, 3, A>(a, st) : launch_conv_dx<128, 5, A>(a, st);     \
        return narrow ? launch_conv_dx<64, 3, A>(a, st) : launch_conv_dx<64, 5, A>(a, st);                    \
    }
        DX_ABL_CASE(1) DX_ABL_CASE(2) DX_ABL_CASE(3) DX_ABL_CASE(4) DX_ABL_CASE(5) DX_ABL_CASE(6) DX_ABL_CASE(7) DX_ABL_CASE(8) DX_ABL_CASE(12)
#undef DX_ABL_CASE
    }
#endif
    if (wide_bm) return narrow ? launch_conv_dx<128, 3>(a, st) : launch_conv_dx<128, 5>(a, st);
    return narrow ? launch_conv_dx<64, 3>(a, st) : launch_conv_dx<64, 5>(a, st);
}

}  // namespace mmku
