// Differentiable ICP on gfx950: brute-force nearest neighbour (LDS-tiled), robust
// weighted normal-equation accumulation (wave64 shuffle reduction), 3x3 / 6x6
// Gauss-Newton solve + SE(2)/SE(3) exponential update, and the reverse sweep.
//
// Replaces dICP.ICP.ICP(...).icp(...) as called from the reference at
// mm_masking/icp_weight_policy.py:281-287 (external/dICP itself is absent from the
// reference tree).  The arithmetic is the normative spec of DESIGN.md §3 and is
// restated on the CPU in oracle/dicp_ref.py + oracle/nn_search.c; per-point fp32
// operations are written in the same order as there and this file MUST be built
// with -ffp-contract=off so that only the explicit fmaf calls fuse.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>

#include "mmk_common.h"

namespace {

constexpr int NN_THREADS = 256;
constexpr int NN_TILE = 1024;  // target points staged in LDS per tile
constexpr float NN_PAD = 3.0e18f;
constexpr int ACC_THREADS = 256;
constexpr int PART_ROWS = 32;         // rows of block partials staged in LDS per round by the per-pair kernels

__host__ __device__ constexpr int nacc(int dim) { return dim == 2 ? 9 : 27; }
__host__ __device__ constexpr int npose(int dim) { return dim == 2 ? 6 : 12; }

// ------------------------------------------------------------------------------------------
// I1: p = R s + t, fp32, individually rounded left to right (oracle: transform_points).
template <int DIM>
__device__ __forceinline__ void transform_point(const float *T, const float s[3], float p[DIM])
{
    if constexpr (DIM == 2) {
        p[0] = (T[0] * s[0] + T[1] * s[1]) + T[3];
        p[1] = (T[4] * s[0] + T[5] * s[1]) + T[7];
    } else {
        p[0] = ((T[0] * s[0] + T[1] * s[1]) + T[2] * s[2]) + T[3];
        p[1] = ((T[4] * s[0] + T[5] * s[1]) + T[6] * s[2]) + T[7];
        p[2] = ((T[8] * s[0] + T[9] * s[1]) + T[10] * s[2]) + T[11];
    }
}

// Squared distance in the fmaf form of oracle/nn_search.c.
template <int DIM>
__device__ __forceinline__ float nn_dist(float tx, float ty, float tz, const float p[DIM])
{
    float dx = tx - p[0];
    float dy = ty - p[1];
    float d = __builtin_fmaf(dy, dy, dx * dx);
    if constexpr (DIM == 3) {
        float dz = tz - p[2];
        d = __builtin_fmaf(dz, dz, d);
    }
    return d;
}

// ------------------------------------------------------------------------------------------
// Planar, padded copy of the target coordinates: (B,M,cols) AoS -> (B,DIM,Mpad).
// `dec` (optional, (B, dim, Mpad / dec_stride)): every dec_stride-th target again, densely -- the samples of the first
// iteration's coarse seed pass (nn_coarse_seed_kernel), which would otherwise gather them 4 bytes at a time.
__global__ void pack_target_kernel(const float *__restrict__ tgt, int M, int cols, int dim, int Mpad,
                                   float *__restrict__ out, float *__restrict__ dec, int dec_stride)
{
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    int b = blockIdx.y;
    if (j >= Mpad) return;
    for (int c = 0; c < dim; ++c) {
        float v = NN_PAD;
        if (j < M) v = tgt[((size_t)b * M + j) * cols + c];
        out[((size_t)b * dim + c) * Mpad + j] = v;
        if (dec != nullptr && j % dec_stride == 0) dec[((size_t)b * dim + c) * (Mpad / dec_stride) + j / dec_stride] = v;
    }
}

// ------------------------------------------------------------------------------------------
// I2: brute-force NN with an exact pre-filter.  The derivation below is the vector-pipe form of rounds 2-3 (kernel removed in
// round 5: HISTORY.md 5, 9.1); the kernel of this file prices the pairs on the matrix cores (nn_mfma_kernel, further down) with
// the same unit decomposition, seeds and exact re-scan.  Work is cut into units (pair, source block, target range); a persistent 1-D grid walks them with
// stride gridDim.x.  Units are numbered so that unit % 8 == pair % 8: blocks of one pair share an XCD (blockIdx % 8 label) and
// its target planes stay in that L2.  The partial results of the target ranges meet in one 64-bit atomic min per source point
// on the key (float bits of d2) << 32 | index: d2 >= 0, so unsigned order == float order, and equal distances resolve to the
// lowest index; min is order independent, so the result is deterministic.
// A plain scan spends 4.5 vector lane-operations per (point, target) pair on d = fma(dy, dy, dx * dx) (round 1's kernel:
// bound by the fp32 VALU).  Here every pair is first priced with the expanded form
//      e_j = fma(tx_j, a, fma(ty_j, b, tn_j)),   a = -2 px, b = -2 py,  tn_j = |t_j|^2 (formed once per target while
//      its tile is staged into LDS)              -> 2 lane-operations (+ 1/2 for the running minimum)
// which equals D_j - |p|^2 up to rounding, D_j the true squared distance.  Still exhaustive: every target is
// visited by every point; only targets that can be the argmin are evaluated with the normative formula.
//
// Exactness.  u = 2^-24.  The normative value obeys |d_j - D_j| <= 5u D_j (four relative roundings).  For a
// target with |t_j| <= |p| + sqrt(D_j):  |e_j - (D_j - |p|^2)| <= G(D_j), G(D) = u (7.1 |p|^2 + 10.1 |p| sqrt(D) +
// 5.1 D)  [tn: 2 roundings; each fma one rounding of a partial sum bounded by |t|^2 + 2 |p||t|].  Let j* be the
// normative argmin and m the argmin of e, X = D_m.  d_j* <= d_k for all k gives D_j* <= X (1 + 11u), hence
//      e_j* - e_m <= 11u X + G(X (1 + 11u)) + G(X) <= u (24.3 |p|^2 + 31.4 X),   X <= (e_m + |p|^2)(1 + 13u).
// So j* lies among the targets with e_j <= thr(e_m), thr(b) = b + kappa (|p|^2 + max(0, b + |p|^2)), kappa = 48u
// (dim 2; 64u for dim 3 where the same derivation gives u (38 |p|^2 + 41 X)): the constants leave > 30 % of slack
// for the fp32 evaluation of thr itself.  thr is monotone in b, so testing a chunk of 32 targets against the
// RUNNING minimum b_run >= e_m flags a superset of the chunks that hold such a target; the flagged chunks (one
// bit each) are then re-scanned with nn_dist() -- from the LDS tile, before the next tile is staged -- in ascending
// order with a strict '<', which yields exactly the lowest-index argmin of the oracle (oracle/nn_search.c):
// tests/test_gpu_icp.py::test_nn_bit_exact.  b_run starts from e of the previous iteration's correspondent (any
// e_j bounds the minimum), so after the first iteration little more than the chunks that matter is flagged.
constexpr int PF_CH = 32;            // targets per flag bit (32 chunks per LDS tile: one flag word per point and tile)

// Identical source rows have identical nearest neighbours.  The reference pads every scan with all-zero rows up to the
// largest cloud of the data set (icp_weight_dataset.py:377-381: about half of the 5 120 rows of a synthetic scan), and all
// of them map to the same transformed point.  Once per icp() call this kernel finds, per pair, the first all-zero row
// (zrep) and the source blocks that hold nothing else (allzero); a block that is all zero and lies behind zrep is not
// scanned at all -- its rows take the key of row zrep, which an earlier block computed (icp_accumulate_kernel) --
// exactly what the scan would have returned for them.
__global__ __launch_bounds__(NN_THREADS) void src_zero_scan_kernel(const float *__restrict__ src, int N, int pts, int nsb,
                                                                  int32_t *__restrict__ allzero, int32_t *__restrict__ zrep)
{
    const int sb = blockIdx.x, b = blockIdx.y;
    int first = N;
    int nonzero = 0;
    for (int q = threadIdx.x; q < pts; q += blockDim.x) {
        const int i = sb * pts + q;
        if (i < N) {
            const float *sp = src + ((size_t)b * N + i) * 3;
            const bool z = sp[0] == 0.f && sp[1] == 0.f && sp[2] == 0.f;
            nonzero |= z ? 0 : 1;
            if (z && i < first) first = i;
        }
    }
    const int any = __syncthreads_or(nonzero);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) first = min(first, __shfl_down(first, off, 64));
    if ((threadIdx.x & 63) == 0 && first < N) atomicMin(&zrep[b], first);
    if (threadIdx.x == 0) allzero[(size_t)b * nsb + sb] = any ? 0 : 1;
}

// The source blocks that are scanned, as dense lists: one list per class b % 8 (the class picks the XCD whose L2 holds the
// pair's target, as in the kernels' unit numbering), entries (b << 12 | sb) in ascending (b, sb); ucnt[c] = length of
// list c, ucnt[8] = the longest.  With the skipped blocks gone from the numbering the persistent blocks of the NN kernel
// all get the same amount of work (with `continue` on a skipped unit some blocks had two units and others none).
__global__ void src_units_kernel(const int32_t *__restrict__ allzero, const int32_t *__restrict__ zrep, int B, int nsb, int pts,
                                 int cap, int32_t *__restrict__ ulist, int32_t *__restrict__ ucnt)
{
    const int c = threadIdx.x;
    int n = 0;
    if (c < 8)
        for (int b = c; b < B; b += 8)
            for (int sb = 0; sb < nsb; ++sb)
                if (!(allzero[(size_t)b * nsb + sb] != 0 && zrep[b] < sb * pts)) ulist[(size_t)c * cap + n++] = (b << 12) | sb;
    if (c < 8) ucnt[c] = n;
    int mx = n;
#pragma unroll
    for (int off = 4; off > 0; off >>= 1) mx = max(mx, __shfl_down(mx, off, 64));
    if (c == 0) ucnt[8] = mx;
}

// ------------------------------------------------------------------------------------------
// I2, the brute-force engine: the exact pre-filter above with every (point, target) pair priced on the MATRIX cores.  The fp32 VALU is what bounds the scan above (2 FMAs + 1/2 min per pair); the
// bf16 matrix pipe runs beside the vector pipe and prices 32 x 32 pairs per v_mfma_f32_32x32x16_bf16 (32 cycles per SIMD),
// which leaves the vector pipe the chunk minima and the flags only: ~0.8 vector instructions per 64 pairs instead of 3.3.
//
// How a bf16 product gives an fp32-grade filter.  An fp32 number splits EXACTLY into three bf16 pieces, x = x1 + x2 + x3
// (x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2): 8 + 8 + 8 significant bits, |x2| <= 2^-9 |x|, |x3| <= 2^-18 |x|),
// products of pieces are exact in fp32, and the matrix core accumulates in fp32.  With a = -2p (exact) and n = fl(|t|^2)
// (formed in fp32 while the tile is staged: |n - |t|^2| <= 2u |t|^2, u = 2^-24), the 16 k-slots of one instruction hold
//      k0..5   tx1 ax1, tx2 ax1, tx1 ax2, tx3 ax1, tx2 ax2, tx1 ax3        (= tx ax up to 2^-26 |tx ax|: the three
//      k6..11  ty1 ay1, ty2 ay1, ty1 ay2, ty3 ay1, ty2 ay2, ty1 ay3         products left out are <= 2^-27, 2^-27, 2^-36)
//      k12..14 n1 * 1, n2 * 1, n3 * 1;  k15 = 0
// so that E_j = sum_k A[j][k] B[k][i] equals e_j = |t_j|^2 - 2 p_i . t_j = D_j - |p|^2 up to
//      |E_j - e_j| <= 2u |t|^2 + u/2 |t||p| + 32.7u (|t|^2 + 2 |t||p|)
// -- the last term for the accumulation of the <= 16 addends in WHATEVER order the hardware adds them, each addition charged
// 2u of the sum of magnitudes S <= 1.02 (|t|^2 + 2|t||p|) (twice the rounding unit: also covers truncating adders).  With
// |t| <= |p| + sqrt(D) and 2 |p| sqrt(D) <= |p|^2 + D:   |E_j - e_j| <= G(D_j),  G(D) = u (168.3 |p|^2 + 102.4 D).
// The fp32 seed of the running bound (e of the previous iteration's correspondent, the fma chain e = fma(tx, a, fma(ty, b, tn)))
// errs by less than that.  Exactness then follows as above: j* the normative argmin, m the target whose computed value is
// the running bound b, X = D_m: D_j* <= X (1 + 11u), so
//      E_j* <= e_m + 11u X + G(X (1 + 11u)) <= b + G(X) + 11u X + G(X(1 + 11u)) <= b + u (336.6 |p|^2 + 216 X),
//      X <= (b + |p|^2)(1 + 103u) + 169u |p|^2,
// i.e. j* lies in a chunk whose minimum is <= thr(b), thr(b) = b + kappa (|p|^2 + (b + |p|^2)) = b (1 + kappa) + 2 kappa |p|^2,
// kappa = 352u (15u |p|^2 of slack for the fp32 evaluation of thr; b + |p|^2 < 0 happens by rounding only, where X is
// O(u |p|^2) and the term it multiplies is immaterial).  The kernel tests min < thr with kappa = 360u (strict: the chain of
// v_min3 starts from thr itself), which flags whatever min <= thr flags at 352u: thr_360 - thr_352 = 8u (|p|^2 + D) > 0,
// and c2 carries an absolute 1e-30 for the point that sits on the origin.  thr is monotone in b, every lane tests the 16 values it holds of a
// 32-target chunk against ITS OWN running minimum (an upper bound of the true one: a superset of the chunks is flagged), the
// two lanes that share a point OR their flag words, and the flagged chunks are re-scanned with nn_dist() from the fp32
// planes kept in LDS next to the fragments -- ascending, strict '<': exactly the oracle's lowest-index argmin, as before
// (tests/test_gpu_icp.py::test_nn_bit_exact and every ICP test; kappa is 7x the VALU filter's, which costs re-scans --
// 0.03 m^2 of margin at 35 m from the sensor -- not correctness).
// Layout: a block of 4 waves takes a source block of 512 points (the unit decomposition, XCD classes, zero-row lists and
// 64-bit atomic-min merge are those of the kernels above); wave w holds 4 groups of 32 points as B operands (columns);
// a tile of 1 024 targets sits in LDS as 32 A fragments [chunk][lane][8 bf16] (lane l: target l & 31, k = 8 (l >> 5) ...),
// written by the thread that staged the target; the result tile has the point on the lane (column l & 31) and 16 targets
// in the lane's registers.  For the re-scan and the output lane l owns the points of groups 2 (l >> 5) + {0, 1}.
typedef __attribute__((ext_vector_type(8))) __bf16 nn_bf16x8;
typedef __attribute__((ext_vector_type(16))) float nn_f32x16;
constexpr int NNM_GROUPS = 4;                      // 32-point groups per wave
constexpr int NNM_PTS = NN_THREADS / 64 * NNM_GROUPS * 32;    // 512 points per block

// D = A B (C = 0).  The compiler builtin, NOT inline asm: hipcc then keeps the result tile in VGPRs (gfx950's register file is
// unified; the vector pipe consumes the tile at once) and pads the MFMA -> VALU read hazard itself (s_nop 8-10 behind each pair
// of MFMAs; nothing is padded around asm statements).  Measured equal to a hand-scheduled asm form (134 us per launch either way).
__device__ __forceinline__ nn_f32x16 mfma_32x32x16_bf16_c0(nn_bf16x8 a, nn_bf16x8 b)
{
    const nn_f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, z, 0, 0, 0);
}
__device__ __forceinline__ void split3(float x, __bf16 &x1, __bf16 &x2, __bf16 &x3)
{
    x1 = (__bf16)x;
    const float r = x - (float)x1;
    x2 = (__bf16)r;
    x3 = (__bf16)(r - (float)x2);
}

// dim 3 (round 4): 21 k-slots -- 6 products per coordinate and the three pieces of |t|^2 -- fill two k-slices, i.e. two chained
// MFMAs per (32 targets x 32 points); the target tile is staged 512 targets at a time so that the fragments still take 32 KB
// (four blocks per CU by LDS; three by registers).  Error bound, same derivation with three coordinates: n = fl(|t|^2) carries
// three roundings (3u |t|^2), the nine dropped products u/2 |t||p| (Cauchy-Schwarz over the coordinates), the 21 addends and the
// chaining addition are charged 2u S each, S <= 1.02 (|t|^2 + 2 |t||p|): |E_j - e_j| <= 48u |t|^2 + 90.5u |t||p| <= G3(D) =
// u (231.8 |p|^2 + 141.3 D).  The normative distance has one more fma: |d - D| <= 6u D, D_j* <= X (1 + 13u).  Hence
// E_j* <= b + u (463.5 |p|^2 + 296 X), X <= (b + |p|^2)(1 + 145u) + 232u |p|^2, and thr(b) = b (1 + kappa) + 2 kappa |p|^2 with
// kappa = 496u covers it with 15u |p|^2 to spare for its own evaluation (launch_nn passes 496u for dim 3, 360u for dim 2).
template <int DIM>
__global__ __launch_bounds__(NN_THREADS, DIM == 2 ? 4 : 3) void nn_mfma_kernel(
    const float *__restrict__ src, const float *__restrict__ tgtp,
    const float *__restrict__ Tk, const int32_t *__restrict__ active, const int32_t *__restrict__ prev_idx,
    const int32_t *__restrict__ ulist, const int32_t *__restrict__ ucnt, int ucap, int B,
    int N, int Mpad, int nsb, int ntu, int tiles_per_unit, int total_units, float kappa,
    unsigned long long *__restrict__ packed)
{
    static_assert((DIM == 2 || DIM == 3) && NNM_GROUPS == 4, "the matrix-core filter: 16 (dim 2) or 2 x 16 (dim 3) k-slots, 4 point groups per wave");
    constexpr int KS = DIM == 2 ? 1 : 2;             // k-slices = chained MFMAs per chunk
    constexpr int TT = NN_TILE / KS;                 // targets staged per sub-tile
    constexpr int NCH = TT / 32;                     // chunks per sub-tile (one flag bit each)
    constexpr int TPT = TT / NN_THREADS;             // targets a thread stages: 4 (dim 2) or 2 (dim 3)
    __shared__ __attribute__((aligned(16))) uint4 frag[KS][NCH][64];             // 32 KB: A fragments of the sub-tile
    // the fp32 planes (exact re-scan), 8 / 6 KB, laid out plainly.  In the re-scan every lane reads the eight float4s of ITS OWN
    // flagged chunk, so chunks of one parity share eight 16-byte bank slots and 45 % of the kernel's LDS cycles are conflict
    // cycles (profiles/r03_nn_pmc_counters.json) -- which costs no time: a conflict-free image (float4 h of chunk c at position
    // h ^ ((c >> 1) & 7)) was measured 1-3 us per launch SLOWER (131-134 against 130-131 us: eight more vector instructions
    // per re-scanned chunk on a loop the vector pipe binds, round 4), so the plain image stays.
    __shared__ __attribute__((aligned(16))) float lt[DIM][TT];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int col = lane & 31, half = lane >> 5;
    const int ntiles = Mpad / NN_TILE;
    if (ulist != nullptr) total_units = ucnt[8] * ntu * 8;
    const float kp1 = 1.0f + kappa;

    // (static stride over the units, as in the kernels above.  A work queue -- one atomic draw per unit from a head per XCD
    // class, issued a unit ahead -- was tried and lost 25-50 us per launch: the block-wide hand-over of the drawn unit costs
    // more than the imbalance it removes.)
    for (int u = blockIdx.x; u < total_units; u += gridDim.x) {
        const int xcd = u & 7;
        int rest = u >> 3;
        const int tu = rest % ntu;
        rest /= ntu;
        int sb, b;
        if (ulist != nullptr) {
            if (rest >= ucnt[xcd]) continue;
            const int code = ulist[(size_t)xcd * ucap + rest];
            sb = code & 0xfff;
            b = code >> 12;
        } else {
            sb = rest % nsb;
            b = (rest / nsb) * 8 + xcd;
        }
        if (b >= B) continue;
        if (active != nullptr && active[b] == 0) continue;

        float T[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) T[i] = Tk[(size_t)b * 16 + i];
        const float *tb = tgtp + (size_t)b * DIM * Mpad;

        // ---- the wave's 4 x 32 points: this lane's column of every group (both lanes l, l + 32 of a column do the same)
        // Coordinates as separate scalars, selected with v_cndmask.  As an array p[g][c] handed to transform_point() and then
        // read as `half ? p[2 + q] : p[q]` they became a 48-byte scratch-memory array read back at a run-time offset; with that
        // build 16-lane groups of a wave now and then worked on wrong coordinates (correspondences that changed from run to
        // run; a NaN coordinate finds nothing, so keys that nobody armed).  The round trip is sound in isolation
        // (scripts/ubench/scratch_roundtrip.hip) and so is the code object (HISTORY.md section 10); what failed in the full
        // process is not established.  The library uses no scratch memory anywhere:
        // tests/test_code_objects.py::test_no_kernel_uses_scratch_memory.
        float px[NNM_GROUPS], py[NNM_GROUPS], pz[NNM_GROUPS], c2[NNM_GROUPS], brun[NNM_GROUPS], thr[NNM_GROUPS], dseed[NNM_GROUPS];
        int jseed[NNM_GROUPS];
        nn_bf16x8 bfr[KS][NNM_GROUPS];
        unsigned w[NNM_GROUPS];
#pragma unroll
        for (int g = 0; g < NNM_GROUPS; ++g) {
            const int i = sb * NNM_PTS + wv * (NNM_GROUPS * 32) + g * 32 + col;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f;
            if (i < N) {
                const float *sp = src + ((size_t)b * N + i) * 3;
                s0 = sp[0];
                s1 = sp[1];
                s2 = sp[2];
            }
            const float s[3] = {s0, s1, s2};
            float tp[DIM];
            transform_point<DIM>(T, s, tp);
            px[g] = tp[0];
            py[g] = tp[1];
            pz[g] = DIM == 3 ? tp[DIM - 1] : 0.f;
            float pn = tp[0] * tp[0] + tp[1] * tp[1];
            if constexpr (DIM == 3) pn += tp[DIM - 1] * tp[DIM - 1];
            c2[g] = 2.0f * kappa * pn + 1e-30f;
            const float ax = -2.0f * tp[0], ay = -2.0f * tp[1], az = -2.0f * pz[g];
            __bf16 ax1, ax2, ax3, ay1, ay2, ay3;
            split3(ax, ax1, ax2, ax3);
            split3(ay, ay1, ay2, ay3);
            const __bf16 one = (__bf16)1.0f, zero = (__bf16)0.0f;
            const nn_bf16x8 lo = {ax1, ax1, ax2, ax1, ax2, ax3, ay1, ay1};
            if constexpr (DIM == 2) {
                const nn_bf16x8 hi = {ay2, ay1, ay2, ay3, one, one, one, zero};
                bfr[0][g] = half ? hi : lo;
            } else {
                __bf16 az1, az2, az3;
                split3(az, az1, az2, az3);
                const nn_bf16x8 hi = {ay2, ay1, ay2, ay3, az1, az1, az2, az1};
                const nn_bf16x8 lo1 = {az2, az3, one, one, one, zero, zero, zero};
                const nn_bf16x8 hi1 = {zero, zero, zero, zero, zero, zero, zero, zero};
                bfr[0][g] = half ? hi : lo;
                bfr[KS - 1][g] = half ? hi1 : lo1;
            }
            // any e_j bounds the minimum: start from the previous iteration's correspondent (fp32 chain of the filter above)
            float b0 = INFINITY;
            dseed[g] = INFINITY;
            jseed[g] = -1;
            if (prev_idx != nullptr && i < N) {
                const int j = prev_idx[(size_t)b * N + i];
                if (j >= 0 && j < Mpad) {
                    const float jx = tb[j], jy = tb[(size_t)Mpad + j];
                    float jz = 0.f;
                    float jn = __builtin_fmaf(jy, jy, jx * jx);
                    float e = 0.f;
                    if constexpr (DIM == 3) {
                        jz = tb[(size_t)2 * Mpad + j];
                        jn = __builtin_fmaf(jz, jz, jn);
                        e = __builtin_fmaf(jz, az, __builtin_fmaf(jy, ay, jn));
                    } else {
                        e = __builtin_fmaf(jy, ay, jn);
                    }
                    b0 = __builtin_fmaf(jx, ax, e);
                    b0 = (b0 == b0) ? b0 : INFINITY;
                    const float dj = nn_dist<DIM>(jx, jy, jz, tp);       // the normative distance to the seed target
                    dseed[g] = (dj == dj) ? dj : INFINITY;
                    jseed[g] = j;
                }
            }
            brun[g] = b0;
            thr[g] = __builtin_fmaf(b0, kp1, c2[g]);
        }
        // ---- the two points this lane owns for the exact part
        float pox[2], poy[2], poz[2], cur[2], dso[2];
        int pidx[2], jj[2], jso[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            pox[q] = half ? px[2 + q] : px[q];
            poy[q] = half ? py[2 + q] : py[q];
            poz[q] = half ? pz[2 + q] : pz[q];
            pidx[q] = sb * NNM_PTS + wv * (NNM_GROUPS * 32) + (2 * half + q) * 32 + col;
            cur[q] = INFINITY;
            dso[q] = half ? dseed[2 + q] : dseed[q];
            jso[q] = half ? jseed[2 + q] : jseed[q];
        }

        const int t0 = tu * tiles_per_unit;
        const int t1 = min(ntiles, t0 + tiles_per_unit);
        jj[0] = jj[1] = t0 * NN_TILE;

        for (int t = t0 * KS; t < t1 * KS; ++t) {          // sub-tiles of TT targets
            __syncthreads();
            {   // stage: this thread's TPT targets -> the fp32 planes and rows TPT (tid % (32 / TPT)) .. of fragment tid / (32 / TPT)
                float xs[TPT], ys[TPT], zs[TPT];
                const int ch = tid / (32 / TPT), p4 = (tid % (32 / TPT)) * TPT;      // chunk, first target inside the chunk
                const int lpos = ch * 32 + p4;
                if constexpr (TPT == 4) {
                    const float4 vx = *reinterpret_cast<const float4 *>(tb + (size_t)t * TT + tid * 4);
                    const float4 vy = *reinterpret_cast<const float4 *>(tb + (size_t)Mpad + (size_t)t * TT + tid * 4);
                    *reinterpret_cast<float4 *>(&lt[0][lpos]) = vx;
                    *reinterpret_cast<float4 *>(&lt[1][lpos]) = vy;
                    xs[0] = vx.x; xs[1] = vx.y; xs[2] = vx.z; xs[3] = vx.w;
                    ys[0] = vy.x; ys[1] = vy.y; ys[2] = vy.z; ys[3] = vy.w;
#pragma unroll
                    for (int r = 0; r < TPT; ++r) zs[r] = 0.f;
                } else {
                    const float2 vx = *reinterpret_cast<const float2 *>(tb + (size_t)t * TT + tid * 2);
                    const float2 vy = *reinterpret_cast<const float2 *>(tb + (size_t)Mpad + (size_t)t * TT + tid * 2);
                    const float2 vz = *reinterpret_cast<const float2 *>(tb + (size_t)(DIM - 1) * Mpad + (size_t)t * TT + tid * 2);
                    *reinterpret_cast<float2 *>(&lt[0][lpos]) = vx;
                    *reinterpret_cast<float2 *>(&lt[1][lpos]) = vy;
                    *reinterpret_cast<float2 *>(&lt[DIM - 1][lpos]) = vz;
                    xs[0] = vx.x; xs[TPT - 1] = vx.y;
                    ys[0] = vy.x; ys[TPT - 1] = vy.y;
                    zs[0] = vz.x; zs[TPT - 1] = vz.y;
                }
                uint4 *fr = &frag[0][ch][p4];
#pragma unroll
                for (int r = 0; r < TPT; ++r) {
                    float n = __builtin_fmaf(ys[r], ys[r], xs[r] * xs[r]);
                    if constexpr (DIM == 3) n = __builtin_fmaf(zs[r], zs[r], n);
                    __bf16 x1, x2, x3, y1, y2, y3, n1, n2, n3;
                    split3(xs[r], x1, x2, x3);
                    split3(ys[r], y1, y2, y3);
                    split3(n, n1, n2, n3);
                    const nn_bf16x8 lo = {x1, x2, x1, x3, x2, x1, y1, y2};
                    fr[r] = __builtin_bit_cast(uint4, lo);
                    if constexpr (DIM == 2) {
                        const nn_bf16x8 hi = {y1, y3, y2, y1, n1, n2, n3, (__bf16)0.0f};
                        fr[32 + r] = __builtin_bit_cast(uint4, hi);
                    } else {
                        __bf16 z1, z2, z3;
                        split3(zs[r], z1, z2, z3);
                        const __bf16 zero = (__bf16)0.0f;
                        const nn_bf16x8 hi = {y1, y3, y2, y1, z1, z2, z1, z3};
                        const nn_bf16x8 lo1 = {z2, z1, n1, n2, n3, zero, zero, zero};
                        const nn_bf16x8 hi1 = {zero, zero, zero, zero, zero, zero, zero, zero};
                        fr[32 + r] = __builtin_bit_cast(uint4, hi);
                        uint4 *fr1 = &frag[KS - 1][ch][p4];
                        fr1[r] = __builtin_bit_cast(uint4, lo1);
                        fr1[32 + r] = __builtin_bit_cast(uint4, hi1);
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int g = 0; g < NNM_GROUPS; ++g) w[g] = 0u;
            // (tried, no gain: a software pipeline one chunk deep -- the MFMAs of chunk c + 1 issued between the reductions of
            // chunk c, pinned with sched_group_barrier: 173-199 VGPRs, two waves per SIMD instead of three, 20 % slower)
            // t = min(thr, chunk minimum) by a tree of v_min3 that starts from thr (a canonical fp32 value, so that no input of the
            // tree needs quieting); the chunk is flagged when t < thr -- the sign bit of t - thr, shifted into the flag word by one
            // v_alignbit (a compare + select through a lane mask costs 3 ns per MFMA more: scripts/ubench/mfma_valu.hip) -- and
            // min(brun, t) is the new bound either way.  The threshold follows the bound every OTHER chunk: a stale threshold is
            // a larger one (thr is monotone in the bound), which flags a superset, and the fma is one vector instruction less
            // per MFMA on a loop the vector pipe binds (the matrix pipe does NOT hide it: 17.7 ns per MFMA and SIMD alone,
            // 28.8 with the eight v_min3, 36.4 / 30.9 with the bookkeeping before / after this change).
            auto reduce = [&](const nn_f32x16 &acc, int g, bool refresh) {
                auto m3 = [](float a, float b, float c) { return __builtin_fminf(__builtin_fminf(a, b), c); };
                const float m0 = m3(thr[g], acc[0], acc[1]), m1 = m3(acc[2], acc[3], acc[4]), m2 = m3(acc[5], acc[6], acc[7]);
                const float m4 = m3(acc[8], acc[9], acc[10]), m5 = m3(acc[11], acc[12], acc[13]);
                const float n0 = m3(m0, m1, m2), n1 = m3(m4, m5, acc[14]);
                const float t = m3(n0, n1, acc[15]);
                w[g] = __builtin_amdgcn_alignbit(w[g], __float_as_uint(t - thr[g]), 31);      // chunk c ends up in bit 31 - c
                brun[g] = __builtin_fminf(brun[g], t);
                if (refresh) thr[g] = __builtin_fmaf(brun[g], kp1, c2[g]);
            };
            uint4 af_next[KS];
#pragma unroll
            for (int k = 0; k < KS; ++k) af_next[k] = frag[k][0][lane];
#pragma unroll 2
            for (int c = 0; c < NCH; ++c) {
                nn_bf16x8 af[KS];
#pragma unroll
                for (int k = 0; k < KS; ++k) {
                    af[k] = __builtin_bit_cast(nn_bf16x8, af_next[k]);
                    af_next[k] = frag[k][(c + 1) & (NCH - 1)][lane];            // next chunk's fragment in flight behind this chunk's MFMAs
                }
                const bool refresh = (c & 1) != 0;
                // two (chains of) MFMAs in flight per wave (32 accumulator registers)
#pragma unroll
                for (int h2 = 0; h2 < NNM_GROUPS; h2 += 2) {
                    nn_f32x16 acc0 = mfma_32x32x16_bf16_c0(af[0], bfr[0][h2]);
                    nn_f32x16 acc1 = mfma_32x32x16_bf16_c0(af[0], bfr[0][h2 + 1]);
                    if constexpr (KS == 2) {
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[KS - 1], bfr[KS - 1][h2], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[KS - 1], bfr[KS - 1][h2 + 1], acc1, 0, 0, 0);
                    }
                    reduce(acc0, h2, refresh);
                    reduce(acc1, h2 + 1, refresh);
                }
            }
            // ---- flags of the two lanes that share a point, then the exact re-scan of this sub-tile's flagged chunks while
            // it is in LDS (chunks ascending, tiles ascending, strict '<': the lowest index among equal distances)
            unsigned wo[2];
            {
                unsigned wf[NNM_GROUPS];
#pragma unroll
                for (int g = 0; g < NNM_GROUPS; ++g) {
                    // (NCH < 32: the flags sit in the top NCH bits, in chunk order from bit 31 down, once the word has been shifted NCH times)
                    const unsigned wg = NCH == 32 ? w[g] : (w[g] << (32 - NCH));
                    wf[g] = wg | (unsigned)__shfl_xor((int)wg, 32, 64);
                }
                wo[0] = half ? wf[2] : wf[0];
                wo[1] = half ? wf[3] : wf[1];
            }
            // (one loop over the flags of both points a lane owns: the wave leaves it after max over lanes of (flags of point 0 +
            // flags of point 1) rounds instead of the sum of the two maxima)
            while (__any((wo[0] | wo[1]) != 0u)) {
                if ((wo[0] | wo[1]) != 0u) {
                    const bool second = wo[0] == 0u;
                    const unsigned ww = second ? wo[1] : wo[0];
                    const int c = __clz((int)ww);
                    const unsigned rest = ww & ~(0x80000000u >> c);
                    wo[0] = second ? wo[0] : rest;
                    wo[1] = second ? rest : wo[1];
                    float pq[DIM];
                    pq[0] = second ? pox[1] : pox[0];
                    pq[1] = second ? poy[1] : poy[0];
                    if constexpr (DIM == 3) pq[DIM - 1] = second ? poz[1] : poz[0];
                    const int o0 = c * 32;
                    float best = INFINITY;
                    int bj = 0;
#pragma unroll
                    for (int h = 0; h < 8; ++h) {
                        const float4 vx = *reinterpret_cast<const float4 *>(&lt[0][o0 + h * 4]);
                        const float4 vy = *reinterpret_cast<const float4 *>(&lt[1][o0 + h * 4]);
                        float4 vz = make_float4(0.f, 0.f, 0.f, 0.f);
                        if constexpr (DIM == 3) vz = *reinterpret_cast<const float4 *>(&lt[DIM - 1][o0 + h * 4]);
                        const float d0 = nn_dist<DIM>(vx.x, vy.x, vz.x, pq), d1 = nn_dist<DIM>(vx.y, vy.y, vz.y, pq);
                        const float d2 = nn_dist<DIM>(vx.z, vy.z, vz.z, pq), d3 = nn_dist<DIM>(vx.w, vy.w, vz.w, pq);
                        if (d0 < best) { best = d0; bj = h * 4 + 0; }
                        if (d1 < best) { best = d1; bj = h * 4 + 1; }
                        if (d2 < best) { best = d2; bj = h * 4 + 2; }
                        if (d3 < best) { best = d3; bj = h * 4 + 3; }
                    }
                    const int jn = t * TT + o0 + bj;
                    if (!second && best < cur[0]) { cur[0] = best; jj[0] = jn; }
                    if (second && best < cur[1]) { cur[1] = best; jj[1] = jn; }
                }
            }
        }

#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int i = pidx[q];
            if (i >= N) continue;
            if (!(cur[q] < INFINITY) && tu != 0) continue;
            // A seeded point's answer is never farther than its seed target, whose own unit finds and writes it: a unit that
            // holds neither the seed target nor anything as close has nothing to add to the atomic minimum ('<=': an equally
            // distant target with a lower index must still get its say).  13.7 -> 1.5 MB of 8-byte atomics per launch.
            const bool holds_seed = jso[q] >= t0 * NN_TILE && jso[q] < t1 * NN_TILE;
            if (jso[q] >= 0 && !holds_seed && !(cur[q] <= dso[q])) continue;
            if (holds_seed && !(cur[q] <= dso[q])) {       // (cannot happen while the filter holds; keeps the key armed whatever happens)
                cur[q] = dso[q];
                jj[q] = jso[q];
            }
            const unsigned long long key =
                ((unsigned long long)__float_as_uint(cur[q]) << 32) | (unsigned long long)(unsigned)jj[q];
            atomicMin(&packed[(size_t)b * N + i], key);
        }
    }
}

// The first ICP iteration has no previous correspondent to start the filter's bound from, and a bound that starts at infinity
// flags a "record" sequence of chunks in every unit (189 us per launch against 130 for a seeded one).  This pass gives it a
// seed: the nearest of every 64th target.  A seed only bounds the search -- the correspondences stay the exhaustive scan's --
// so its own arithmetic is free: packed fp32 (two samples per instruction), and the sample's number in the low 10 bits of
// the distance so that one unsigned minimum tracks value and index (distances are >= 0: their bit patterns order like the
// values).  Source blocks the scan skips (all-zero rows behind the first zero row) are skipped here too.  Measured (B = 32,
// survey density): 18 us for the pass, the first launch's scan 189 -> ~150 us behind it: 189 -> 180 us in all (every 32nd target
// with exact nn_dist: 35 us for the pass, no gain; 64-thread blocks: 21 us).
constexpr int NN_COARSE_STRIDE = 64, NN_COARSE_TILE = 1024, NN_COARSE_THREADS = 256;
typedef __attribute__((ext_vector_type(2))) float nn_f32x2;
template <int DIM>
__global__ __launch_bounds__(NN_COARSE_THREADS) void nn_coarse_seed_kernel(const float *__restrict__ src, const float *__restrict__ tgtp,
                                                             const float *__restrict__ Tk, const int32_t *__restrict__ active,
                                                             const int32_t *__restrict__ allzero, const int32_t *__restrict__ zrep,
                                                             int nn_pts, int nsb, int N, int Mpad, const float *__restrict__ dec,
                                                             int32_t *__restrict__ seed)
{
    __shared__ __attribute__((aligned(16))) float st[DIM][NN_COARSE_TILE];
    const int b = blockIdx.y;
    if (active != nullptr && active[b] == 0) return;
    if (allzero != nullptr) {
        const int sb = (blockIdx.x * NN_COARSE_THREADS) / nn_pts;
        if (sb < nsb && allzero[(size_t)b * nsb + sb] != 0 && zrep[b] < sb * nn_pts) return;
    }
    const int i = blockIdx.x * NN_COARSE_THREADS + threadIdx.x;
    float T[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) T[q] = Tk[(size_t)b * 16 + q];
    float p[DIM] = {};
    if (i < N) {
        const float *sp = src + ((size_t)b * N + i) * 3;
        const float s[3] = {sp[0], sp[1], sp[2]};
        transform_point<DIM>(T, s, p);
    }
    const float *tb = tgtp + (size_t)b * DIM * Mpad;
    const int ns = Mpad / NN_COARSE_STRIDE;
    unsigned best = 0x7f800000u;          // (+inf: nothing found)
    int best_tile = 0;
    const nn_f32x2 npx = {-p[0], -p[0]}, npy = {-p[1], -p[1]}, npz = {-p[DIM - 1], -p[DIM - 1]};
    for (int s0 = 0; s0 < ns; s0 += NN_COARSE_TILE) {
        __syncthreads();
        for (int q = threadIdx.x; q < NN_COARSE_TILE; q += NN_COARSE_THREADS) {
            const int sidx = s0 + q;
#pragma unroll
            for (int c = 0; c < DIM; ++c)
                st[c][q] = sidx >= ns ? 3e18f
                                      : (dec != nullptr ? dec[((size_t)b * DIM + c) * ns + sidx] : tb[(size_t)c * Mpad + (size_t)sidx * NN_COARSE_STRIDE]);
        }
        __syncthreads();
        const int cnt = min(NN_COARSE_TILE, ns - s0);
        unsigned tb_best = 0x7f800000u;
        for (int q = 0; q < cnt; q += 4) {
            const float4 vx = *reinterpret_cast<const float4 *>(&st[0][q]);
            const float4 vy = *reinterpret_cast<const float4 *>(&st[1][q]);
            const nn_f32x2 dx0 = nn_f32x2{vx.x, vx.y} + npx, dx1 = nn_f32x2{vx.z, vx.w} + npx;
            const nn_f32x2 dy0 = nn_f32x2{vy.x, vy.y} + npy, dy1 = nn_f32x2{vy.z, vy.w} + npy;
            nn_f32x2 d0 = dx0 * dx0, d1 = dx1 * dx1;
            d0 = __builtin_elementwise_fma(dy0, dy0, d0);
            d1 = __builtin_elementwise_fma(dy1, dy1, d1);
            if (DIM == 3) {
                const float4 vz = *reinterpret_cast<const float4 *>(&st[DIM - 1][q]);
                const nn_f32x2 dz0 = nn_f32x2{vz.x, vz.y} + npz, dz1 = nn_f32x2{vz.z, vz.w} + npz;
                d0 = __builtin_elementwise_fma(dz0, dz0, d0);
                d1 = __builtin_elementwise_fma(dz1, dz1, d1);
            }
            const unsigned k0 = (__float_as_uint(d0[0]) & 0xfffffc00u) | (unsigned)q;
            const unsigned k1 = (__float_as_uint(d0[1]) & 0xfffffc00u) | (unsigned)(q + 1);
            const unsigned k2 = (__float_as_uint(d1[0]) & 0xfffffc00u) | (unsigned)(q + 2);
            const unsigned k3 = (__float_as_uint(d1[1]) & 0xfffffc00u) | (unsigned)(q + 3);
            tb_best = min(min(tb_best, min(k0, k1)), min(k2, k3));       // (a NaN's pattern is above +inf: never the minimum)
        }
        if (tb_best < (best & 0xfffffc00u)) {
            best = tb_best;
            best_tile = s0;
        }
    }
    if (i < N) seed[(size_t)b * N + i] = best >= 0x7f800000u ? -1 : (best_tile + (int)(best & 0x3ffu)) * NN_COARSE_STRIDE;
}

constexpr unsigned long long NN_KEY_INIT = ~0ull;

__global__ void nn_unpack_kernel(const unsigned long long *__restrict__ packed, int n, int32_t *__restrict__ idx,
                                 float *__restrict__ d2)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long key = packed[i];
    idx[i] = (int32_t)(unsigned)(key & 0xffffffffull);
    d2[i] = __uint_as_float((unsigned)(key >> 32));
}


// ------------------------------------------------------------------------------------------
// I2, accelerated form: EXACT nearest neighbour through a uniform 2-D grid
// over the target's x,y.  It returns bit for bit what the brute-force scan returns — the same
// nn_dist() arithmetic on the same coordinates, ties to the lowest original index — but only
// visits the cells that can hold the answer: rings of cells around the query are scanned until
// the best distance is provably smaller than anything an unvisited ring can offer (a point in a
// cell at Chebyshev ring r+1 is at least r cells away; targets/queries outside the grid are
// clamped to the border cells, which only makes them farther than the bound).  A query that is
// still unresolved after GRID_RMAX rings scans every target.  The grid is built once per icp()
// call (the target does not move between iterations).
constexpr int GRID_N = 128;
constexpr int GRID_NC = GRID_N * GRID_N;
constexpr float GRID_CELL = 2.0f;
constexpr float GRID_ORG = -128.0f;
constexpr int GRID_RMAX = 24;

__device__ __forceinline__ int grid_coord(float v)
{
    const float f = floorf((v - GRID_ORG) / GRID_CELL);
    return (int)fminf(fmaxf(f, 0.f), (float)(GRID_N - 1));
}

__global__ void grid_count_kernel(const float *__restrict__ tgt, int M, int cols, int32_t *__restrict__ counts)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (j >= M) return;
    const float *t = tgt + ((size_t)b * M + j) * cols;
    const int cell = grid_coord(t[1]) * GRID_N + grid_coord(t[0]);
    atomicAdd(&counts[(size_t)b * GRID_NC + cell], 1);
}

__global__ __launch_bounds__(256) void grid_scan_kernel(const int32_t *__restrict__ counts, int32_t *__restrict__ starts)
{
    __shared__ int sm[4];
    const int b = blockIdx.x;
    constexpr int L = GRID_NC / 256;
    const int32_t *c = counts + (size_t)b * GRID_NC + threadIdx.x * L;
    int s = 0;
    for (int i = 0; i < L; ++i) s += c[i];
    // block exclusive scan of 256 partial sums
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int inc = s;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(inc, off, 64);
        if (lane >= off) inc += o;
    }
    if (lane == 63) sm[wv] = inc;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wv; ++w) base += sm[w];
    int run = base + inc - s;
    int32_t *o = starts + (size_t)b * (GRID_NC + 1) + threadIdx.x * L;
    for (int i = 0; i < L; ++i) {
        o[i] = run;
        run += c[i];
    }
    if (threadIdx.x == 255) starts[(size_t)b * (GRID_NC + 1) + GRID_NC] = run;
}

__global__ void grid_fill_kernel(const float *__restrict__ tgt, int M, int cols, int dim, const int32_t *__restrict__ starts,
                                 int32_t *__restrict__ cursor, float4 *__restrict__ sorted)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (j >= M) return;
    const float *t = tgt + ((size_t)b * M + j) * cols;
    const int cell = grid_coord(t[1]) * GRID_N + grid_coord(t[0]);
    const int pos = starts[(size_t)b * (GRID_NC + 1) + cell] + atomicAdd(&cursor[(size_t)b * GRID_NC + cell], 1);
    sorted[(size_t)b * M + pos] = make_float4(t[0], t[1], dim == 3 ? t[2] : 0.f, __int_as_float(j));
}

template <int DIM>
__global__ __launch_bounds__(256) void grid_nn_kernel(const float *__restrict__ src, const float *__restrict__ Tk,
                                                      const int32_t *__restrict__ active,
                                                      const int32_t *__restrict__ starts,
                                                      const float4 *__restrict__ sorted, int N, int M,
                                                      const float *__restrict__ tgt, int cols,
                                                      const int32_t *__restrict__ prev_idx,
                                                      unsigned long long *__restrict__ packed)
{
    const int b = blockIdx.y;
    if (active != nullptr && active[b] == 0) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float T[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) T[q] = Tk[(size_t)b * 16 + q];
    const float *sp = src + ((size_t)b * N + i) * 3;
    const float s[3] = {sp[0], sp[1], sp[2]};
    float p[DIM];
    transform_point<DIM>(T, s, p);
    const int cx = grid_coord(p[0]), cy = grid_coord(p[1]);
    const int32_t *st = starts + (size_t)b * (GRID_NC + 1);
    const float4 *pts = sorted + (size_t)b * M;
    float best = INFINITY;
    int bi = 0x7fffffff;

    // candidates are fetched eight at a time (independent 16-byte loads) before they are compared:
    // the search is latency-bound, not ALU-bound
    auto scan = [&](int k0, int k1) {
        for (int k = k0; k < k1; k += 8) {
            float4 t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = pts[min(k + u, k1 - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float d = nn_dist<DIM>(t[u].x, t[u].y, t[u].z, p);
                const int j = __float_as_int(t[u].w);
                if (d < best || (d == best && j < bi)) {
                    best = d;
                    bi = j;
                }
            }
        }
    };

    auto scan_square = [&](int r) {
        const int y0 = max(cy - r, 0), y1 = min(cy + r, GRID_N - 1);
        const int x0 = max(cx - r, 0), x1 = min(cx + r, GRID_N - 1);
        int k0 = st[y0 * GRID_N + x0], k1 = st[y0 * GRID_N + x1 + 1];
        for (int y = y0; y <= y1; ++y) {
            const int yn = min(y + 1, y1);
            const int n0 = st[yn * GRID_N + x0], n1 = st[yn * GRID_N + x1 + 1];   // next row's range in flight
            scan(k0, k1);
            k0 = n0;
            k1 = n1;
        }
    };

    bool done = false;
    if (prev_idx != nullptr) {
        // the previous iteration's correspondent bounds the search: every target at most that far
        // away (the answer and all its ties) lies within r cells of the query's cell
        const float *tp = tgt + ((size_t)b * M + prev_idx[(size_t)b * N + i]) * cols;
        const float u = nn_dist<DIM>(tp[0], tp[1], DIM == 3 ? tp[2] : 0.f, p);
        const int r = (int)floorf((sqrtf(u) + 1e-3f) / GRID_CELL) + 1;
        if (r <= GRID_RMAX) {
            scan_square(r);
            done = true;
        }
    }
    // growing squares of cells around the query (inner cells are simply seen again: harmless);
    // after the square of radius r everything unvisited is at least r cells away
    const int radii[8] = {1, 2, 3, 5, 8, 12, 17, GRID_RMAX};
    for (int ri = 0; ri < 8 && !done; ++ri) {
        const int r = radii[ri];
        scan_square(r);
        const float lim = (float)r * GRID_CELL - 1e-3f;   // 1 mm of slack for the float cell assignment
        done = best < lim * lim;
    }
    if (!done) {   // far from every target: exhaustive scan (still exact)
        best = INFINITY;
        bi = 0x7fffffff;
        scan(0, M);
    }
    packed[(size_t)b * N + i] = ((unsigned long long)__float_as_uint(best) << 32) | (unsigned long long)(unsigned)bi;
}

// ------------------------------------------------------------------------------------------
// I3 + I4 per-point terms (oracle: per_point_terms).
template <int DIM, int TYPE>
struct PointTerms {
    static constexpr int P = (DIM == 2) ? 3 : 6;
    static constexpr int NR = (TYPE == MMK_ICP_PT2PT) ? DIM : 1;
    float s[3];
    float p[DIM];
    float ev[DIM];
    float n[DIM];
    float J[NR][P];
    float er[NR];
    float d2, r2, keep, rho, w, omega;
};

template <int DIM, int TYPE>
__device__ __forceinline__ void point_terms(PointTerms<DIM, TYPE> &t, const float *T, const float *srow,
                                            const float *trow, float omega, int loss, float k, float k2,
                                            float trim2)
{
    constexpr int P = PointTerms<DIM, TYPE>::P;
    constexpr int NR = PointTerms<DIM, TYPE>::NR;
    t.s[0] = srow[0];
    t.s[1] = srow[1];
    t.s[2] = srow[2];
    transform_point<DIM>(T, t.s, t.p);
#pragma unroll
    for (int c = 0; c < DIM; ++c) t.ev[c] = trow[c] - t.p[c];
    float d2 = t.ev[0] * t.ev[0] + t.ev[1] * t.ev[1];
    if constexpr (DIM == 3) d2 = d2 + t.ev[2] * t.ev[2];
    t.d2 = d2;
    t.keep = (d2 < trim2) ? 1.f : 0.f;
    if (TYPE == MMK_ICP_PT2PL) {
#pragma unroll
        for (int c = 0; c < DIM; ++c) t.n[c] = trow[3 + c];
        float e = t.n[0] * t.ev[0] + t.n[1] * t.ev[1];
        if constexpr (DIM == 3) e = e + t.n[2] * t.ev[2];
        t.er[0] = e;
        t.r2 = e * e;
        if (DIM == 2) {
            t.J[0][0] = t.n[0];
            t.J[0][1] = t.n[1];
            t.J[0][2] = t.n[1] * t.p[0] - t.n[0] * t.p[1];
        } else {
            t.J[0][0] = t.n[0];
            t.J[0][1] = t.n[1];
            t.J[0][2] = t.n[DIM - 1];
            t.J[0][P - 3] = t.p[1] * t.n[DIM - 1] - t.p[DIM - 1] * t.n[1];
            t.J[0][P - 2] = t.p[DIM - 1] * t.n[0] - t.p[0] * t.n[DIM - 1];
            t.J[0][P - 1] = t.p[0] * t.n[1] - t.p[1] * t.n[0];
        }
    } else {
        t.r2 = d2;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            t.er[r] = t.ev[r];
#pragma unroll
            for (int c = 0; c < P; ++c) t.J[r][c] = 0.f;
            t.J[r][r] = 1.f;
        }
        if (DIM == 2) {
            t.J[0][2] = -t.p[1];
            t.J[NR - 1][2] = t.p[0];
        } else {
            const float px = t.p[0], py = t.p[1], pz = t.p[DIM - 1];
            t.J[0][P - 2] = pz;
            t.J[0][P - 1] = -py;
            t.J[1 % NR][P - 3] = -pz;
            t.J[1 % NR][P - 1] = px;
            t.J[2 % NR][P - 3] = py;
            t.J[2 % NR][P - 2] = -px;
        }
    }
    float rho = 1.f;
    if (loss == MMK_LOSS_CAUCHY) {
        rho = 1.0f / (1.0f + t.r2 / k2);
    } else if (loss == MMK_LOSS_HUBER) {
        const float r = sqrtf(t.r2);
        rho = (r <= k) ? 1.f : k / r;
    }
    t.rho = rho;
    t.omega = omega;
    t.w = (omega * t.keep) * rho;
}

// Accumulate A (upper triangle, row-major) and b for one pair; one point per thread.
template <int DIM, int TYPE>
__global__ __launch_bounds__(ACC_THREADS) void icp_accumulate_kernel(
    const float *__restrict__ src, const float *__restrict__ tgt, int tgt_cols,
    const float *__restrict__ weight, const float *__restrict__ Tk, const int32_t *__restrict__ active,
    const unsigned long long *__restrict__ packed, unsigned long long *__restrict__ packed_next,
    const int32_t *__restrict__ allzero, const int32_t *__restrict__ zrep, int nn_pts, int nn_nsb,
    int32_t *__restrict__ idx_out, int N, int M, int loss, float k,
    float k2, float trim2, double *__restrict__ partials, int32_t *__restrict__ status)
{
    constexpr int P = PointTerms<DIM, TYPE>::P;
    constexpr int NR = PointTerms<DIM, TYPE>::NR;
    constexpr int NACC = nacc(DIM);
    const int b = blockIdx.y;
    if (active != nullptr && active[b] == 0) return;
    const int i = blockIdx.x * ACC_THREADS + threadIdx.x;
    double acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a) acc[a] = 0.0;

    if (i < N) {
        float T[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) T[q] = Tk[(size_t)b * 16 + q];
        // consume the NN key; arm the OTHER key buffer for the next iteration's atomic mins (two buffers: a row of a
        // skipped all-zero block reads the key of the pair's first zero row, which another workgroup owns)
        int isrc = i;
        if (allzero != nullptr && allzero[(size_t)b * nn_nsb + i / nn_pts] != 0 && zrep[b] < (i / nn_pts) * nn_pts) isrc = zrep[b];
        const unsigned long long key = packed[(size_t)b * N + isrc];
        if (packed_next != nullptr) packed_next[(size_t)b * N + i] = NN_KEY_INIT;
        // An index outside the target can only come from a key nobody armed (every NN engine writes each row's key: range 0
        // of the scan always does, also for non-finite points).  It must not become an address -- and it must not pass
        // silently as correspondence 0 either: the launch raises MMK_ICP_STATUS_UNARMED_KEY in the caller's status word
        // (mmk_icp_status; dICP/ICP.py turns it into an MmkError), then clamps so that the rest of the launch stays in bounds.
        const unsigned jraw = (unsigned)(key & 0xffffffffull);
        if ((key == NN_KEY_INIT || jraw >= (unsigned)M) && status != nullptr) atomicOr(status, (int32_t)MMK_ICP_STATUS_UNARMED_KEY);
        const int j = (int)min(jraw, (unsigned)(M - 1));
        idx_out[(size_t)b * N + i] = j;
        const float omega = weight ? weight[(size_t)b * N + i] : 1.f;
        PointTerms<DIM, TYPE> t;
        point_terms<DIM, TYPE>(t, T, src + ((size_t)b * N + i) * 3, tgt + ((size_t)b * M + j) * tgt_cols, omega,
                               loss, k, k2, trim2);
        int a = 0;
#pragma unroll
        for (int m = 0; m < P; ++m) {
#pragma unroll
            for (int n = m; n < P; ++n) {
                double v = 0.0;
#pragma unroll
                for (int r = 0; r < NR; ++r) v += (double)(t.w * t.J[r][m]) * (double)t.J[r][n];
                acc[a++] = v;
            }
        }
#pragma unroll
        for (int m = 0; m < P; ++m) {
            double v = 0.0;
#pragma unroll
            for (int r = 0; r < NR; ++r) v += (double)(t.w * t.J[r][m]) * (double)t.er[r];
            acc[a++] = v;
        }
    }

    __shared__ double red[ACC_THREADS / 64][NACC];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int a = 0; a < NACC; ++a) {
        double v = wave_sum(acc[a]);
        if (lane == 0) red[wv][a] = v;
    }
    __syncthreads();
    if (threadIdx.x < NACC) {
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < ACC_THREADS / 64; ++w) v += red[w][threadIdx.x];
        partials[((size_t)b * gridDim.x + blockIdx.x) * NACC + threadIdx.x] = v;
    }
}

// ------------------------------------------------------------------------------------------
// Small dense fp64 helpers (one lane).
// (the callers keep A, rhs, x and the work space `ws` (P * P + P doubles) in LDS: as per-lane arrays with run-time indices
// they would live in scratch memory, which this library does not use anywhere -- see nn_mfma_kernel)
template <int P>
__device__ bool chol_solve(const double *A /*P*P row-major symmetric*/, const double *rhs, double *x, double *ws)
{
    double (*L)[P] = reinterpret_cast<double (*)[P]>(ws);
    double *y = ws + P * P;
    bool ok = true;
    for (int i = 0; i < P; ++i)
        for (int j = 0; j < P; ++j) {
            L[i][j] = 0.0;
            if (!isfinite(A[i * P + j])) ok = false;
        }
    for (int j = 0; j < P && ok; ++j) {
        double d = A[j * P + j];
        for (int q = 0; q < j; ++q) d -= L[j][q] * L[j][q];
        if (!(d > 0.0)) {
            ok = false;
            break;
        }
        d = sqrt(d);
        L[j][j] = d;
        for (int i = j + 1; i < P; ++i) {
            double v = A[i * P + j];
            for (int q = 0; q < j; ++q) v -= L[i][q] * L[j][q];
            L[i][j] = v / d;
        }
    }
    if (!ok) {
        for (int i = 0; i < P; ++i) x[i] = 0.0;
        return false;
    }
    for (int i = 0; i < P; ++i) {
        double v = rhs[i];
        for (int q = 0; q < i; ++q) v -= L[i][q] * y[q];
        y[i] = v / L[i][i];
    }
    for (int i = P - 1; i >= 0; --i) {
        double v = y[i];
        for (int q = i + 1; q < P; ++q) v -= L[q][i] * x[q];
        x[i] = v / L[i][i];
    }
    return true;
}

// Closed-form Exp(delta) -> 4x4 (oracle: se_exp).
template <int DIM>
__device__ void se_exp(const double *dl, double E[16])
{
    for (int i = 0; i < 16; ++i) E[i] = 0.0;
    E[0] = E[5] = E[10] = E[15] = 1.0;
    if (DIM == 2) {
        const double x = dl[0], y = dl[1], th = dl[2];
        const double th2 = th * th;
        double a, bb;
        if (th2 < 1e-8) {
            a = 1.0 - th2 / 6.0 + th2 * th2 / 120.0;
            bb = th * (0.5 - th2 / 24.0 + th2 * th2 / 720.0);
        } else {
            a = sin(th) / th;
            bb = (1.0 - cos(th)) / th;
        }
        const double c = cos(th), s = sin(th);
        E[0] = c; E[1] = -s; E[3] = a * x - bb * y;
        E[4] = s; E[5] = c;  E[7] = bb * x + a * y;
    } else {
        const double *rho = dl, *phi = dl + 3;
        const double th2 = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
        double A_, B_, C_;
        if (th2 < 1e-8) {
            A_ = 1.0 - th2 / 6.0 + th2 * th2 / 120.0;
            B_ = 0.5 - th2 / 24.0 + th2 * th2 / 720.0;
            C_ = 1.0 / 6.0 - th2 / 120.0 + th2 * th2 / 5040.0;
        } else {
            const double th = sqrt(th2);
            A_ = sin(th) / th;
            B_ = (1.0 - cos(th)) / th2;
            C_ = (th - sin(th)) / (th2 * th);
        }
        const double K[9] = {0.0, -phi[2], phi[1], phi[2], 0.0, -phi[0], -phi[1], phi[0], 0.0};
        double K2[9];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                double v = 0.0;
                for (int q = 0; q < 3; ++q) v += K[i * 3 + q] * K[q * 3 + j];
                K2[i * 3 + j] = v;
            }
        for (int i = 0; i < 3; ++i) {
            double tv = 0.0;
            for (int j = 0; j < 3; ++j) {
                const double id = (i == j) ? 1.0 : 0.0;
                E[i * 4 + j] = id + A_ * K[i * 3 + j] + B_ * K2[i * 3 + j];
                tv += (id + B_ * K[i * 3 + j] + C_ * K2[i * 3 + j]) * rho[j];
            }
            E[i * 4 + 3] = tv;
        }
    }
}

// I5 + I6: sum the block partials in a fixed order, solve, update the pose.
template <int DIM>
__global__ __launch_bounds__(64) void icp_solve_kernel(const double *__restrict__ partials, int nblk,
                                                       const float *__restrict__ T_in, float *__restrict__ T_out,
                                                       double *__restrict__ delta_out, double *__restrict__ A_out,
                                                       const int32_t *__restrict__ active_in,
                                                       int32_t *__restrict__ active_out, float tol)
{
    constexpr int P = (DIM == 2) ? 3 : 6;
    constexpr int NACC = nacc(DIM);
    const int b = blockIdx.x;
    const int lane = threadIdx.x;
    __shared__ double acc[NACC];
    __shared__ double stage[PART_ROWS * NACC];
    const bool act = active_in[b] != 0;
    {
        // the block partials, summed in index order as before; the wave fetches PART_ROWS rows at a time
        // (coalesced, all in flight) instead of one dependent load per row and lane
        double v = 0.0;
        if (act) {
            const double *base = partials + (size_t)b * nblk * NACC;
            for (int r0 = 0; r0 < nblk; r0 += PART_ROWS) {
                const int nr = min(PART_ROWS, nblk - r0);
                for (int e = lane; e < nr * NACC; e += 64) stage[e] = base[(size_t)r0 * NACC + e];
                __syncthreads();
                if (lane < NACC)
                    for (int q = 0; q < nr; ++q) v += stage[q * NACC + lane];
                __syncthreads();
            }
        }
        if (lane < NACC) acc[lane] = v;
    }
    __syncthreads();
    if (lane != 0) return;
    __shared__ double A[P * P], rhs[P], dl[6], E[16], ws[P * P + P];
    for (int q = 0; q < 6; ++q) dl[q] = 0.0;
    int a = 0;
    for (int m = 0; m < P; ++m)
        for (int n = m; n < P; ++n) {
            A[m * P + n] = acc[a];
            A[n * P + m] = acc[a];
            ++a;
        }
    for (int m = 0; m < P; ++m) rhs[m] = acc[a++];
    for (int q = 0; q < 36; ++q) A_out[(size_t)b * 36 + q] = (q < P * P) ? A[q] : 0.0;
    if (act) chol_solve<P>(A, rhs, dl, ws);
    for (int q = 0; q < 6; ++q) delta_out[(size_t)b * 6 + q] = dl[q];
    if (!act) {
        for (int q = 0; q < 16; ++q) T_out[(size_t)b * 16 + q] = T_in[(size_t)b * 16 + q];
        active_out[b] = 0;
        return;
    }
    se_exp<DIM>(dl, E);
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double v = 0.0;
            for (int q = 0; q < 4; ++q) v += E[i * 4 + q] * (double)T_in[(size_t)b * 16 + q * 4 + j];
            T_out[(size_t)b * 16 + i * 4 + j] = (float)v;
        }
    double nrm = 0.0;
    for (int q = 0; q < P; ++q) nrm += dl[q] * dl[q];
    active_out[b] = (sqrt(nrm) < (double)tol) ? 0 : 1;
}

// The head of a forward call in ONE launch: nearest-neighbour keys armed (both buffers), status word cleared, every pair
// active, T_hist[0] = T_init, the zero-row representatives reset -- five copy / fill launches of 3-6 us each before.
__global__ __launch_bounds__(256) void icp_init_kernel(unsigned long long *__restrict__ keys, size_t nkeys, int32_t *__restrict__ status,
                                                       int32_t *__restrict__ active0, const float *__restrict__ T_init,
                                                       float *__restrict__ T0, int32_t *__restrict__ zrep, int B, int N)
{
    const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = i0; i < nkeys; i += stride) keys[i] = ~0ull;
    if (i0 < 16) status[i0] = 0;
    if (i0 < (size_t)B) {
        active0[i0] = 1;
        if (zrep != nullptr) zrep[i0] = N;
    }
    if (i0 < (size_t)B * 16) T0[i0] = T_init[i0];
}

__global__ void fill_i32_kernel(int32_t *p, int n, int32_t v)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// ------------------------------------------------------------------------------------------
// I7 backward, per-pair part.  One wave per pair; the 64 lanes are the 8x8 entries of the
// block matrix [[X^T, Ebar],[0, X^T]] whose exponential carries exp(X)^T (top-left) and the
// adjoint of the exponential map applied to Ebar (top-right).
template <int DIM>
__global__ __launch_bounds__(64) void icp_bwd_pair_kernel(
    const double *__restrict__ Gdir_in, const double *__restrict__ pparts, int nparts,
    const float *__restrict__ Tk, const double *__restrict__ delta, const double *__restrict__ Amat,
    const int32_t *__restrict__ active, double *__restrict__ Gdir_out, double *__restrict__ lam_out)
{
    constexpr int P = (DIM == 2) ? 3 : 6;
    constexpr int NP = npose(DIM);
    const int b = blockIdx.x;
    const int lane = threadIdx.x;
    __shared__ double G[16], Eb[16], X[16], M0[64], R0[64], R1[64];
    __shared__ double stage[PART_ROWS * NP];

    // full gradient w.r.t. T_{k+1}: direct part + point path of the later iteration (the parts are
    // added in index order; rows are fetched PART_ROWS at a time by the whole wave)
    {
        double v = (lane < 16) ? Gdir_in[(size_t)b * 16 + lane] : 0.0;
        const int r = lane >> 2, c = lane & 3;
        int slot = -1;
        if (lane < 16 && r < DIM && c < DIM) slot = r * DIM + c;
        if (lane < 16 && r < DIM && c == 3) slot = DIM * DIM + r;
        const double *base = pparts + (size_t)b * nparts * NP;
        for (int r0 = 0; r0 < nparts; r0 += PART_ROWS) {
            const int nr = min(PART_ROWS, nparts - r0);
            for (int e = lane; e < nr * NP; e += 64) stage[e] = base[(size_t)r0 * NP + e];
            __syncthreads();
            if (slot >= 0)
                for (int q = 0; q < nr; ++q) v += stage[q * NP + slot];
            __syncthreads();
        }
        if (lane < 16) G[lane] = v;
    }
    __syncthreads();
    const bool act = (active != nullptr) && active[b] != 0;
    if (!act) {
        if (lane < 16) Gdir_out[(size_t)b * 16 + lane] = G[lane];
        if (lane < 6 && lam_out) lam_out[(size_t)b * 6 + lane] = 0.0;
        return;
    }
    const double *dl = delta + (size_t)b * 6;
    if (lane < 16) {
        const int r = lane >> 2, c = lane & 3;
        // Ebar = G * T_k^T
        double v = 0.0;
        for (int q = 0; q < 4; ++q) v += G[r * 4 + q] * (double)Tk[(size_t)b * 16 + c * 4 + q];
        Eb[lane] = v;
        double x = 0.0;
        if (DIM == 2) {
            if (r == 0 && c == 3) x = dl[0];
            if (r == 1 && c == 3) x = dl[1];
            if (r == 1 && c == 0) x = dl[2];
            if (r == 0 && c == 1) x = -dl[2];
        } else {
            if (c == 3 && r < 3) x = dl[r];
            if (r == 2 && c == 1) x = dl[3];
            if (r == 1 && c == 2) x = -dl[3];
            if (r == 0 && c == 2) x = dl[4];
            if (r == 2 && c == 0) x = -dl[4];
            if (r == 1 && c == 0) x = dl[5];
            if (r == 0 && c == 1) x = -dl[5];
        }
        X[lane] = x;
    }
    __syncthreads();
    const int i8 = lane >> 3, j8 = lane & 7;
    {
        double m = 0.0;
        const double sc = 1.0 / 64.0;
        if (i8 < 4 && j8 < 4) m = X[j8 * 4 + i8] * sc;             // X^T
        else if (i8 >= 4 && j8 >= 4) m = X[(j8 - 4) * 4 + (i8 - 4)] * sc;
        else if (i8 < 4 && j8 >= 4) m = Eb[i8 * 4 + (j8 - 4)] * sc;
        M0[lane] = m;
        R0[lane] = ((i8 == j8) ? 1.0 : 0.0) + m / 13.0;
    }
    __syncthreads();
    double *cur = R0, *nxt = R1;
    // Horner: R = I + M R / n, n = 12 .. 1
    for (int n = 12; n >= 1; --n) {
        double v = 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q) v += M0[i8 * 8 + q] * cur[q * 8 + j8];
        nxt[lane] = ((i8 == j8) ? 1.0 : 0.0) + v / (double)n;
        __syncthreads();
        double *t = cur; cur = nxt; nxt = t;
    }
    for (int sq = 0; sq < 6; ++sq) {
        double v = 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q) v += cur[i8 * 8 + q] * cur[q * 8 + j8];
        nxt[lane] = v;
        __syncthreads();
        double *t = cur; cur = nxt; nxt = t;
    }
    // cur = [[E^T, Xbar],[0, E^T]]
    if (lane < 16) {
        const int r = lane >> 2, c = lane & 3;
        double v = 0.0;
        for (int q = 0; q < 4; ++q) v += cur[r * 8 + q] * G[q * 4 + c];  // E^T G
        Gdir_out[(size_t)b * 16 + lane] = v;
    }
    if (lane == 0) {
        auto xb = [&](int r, int c) { return cur[r * 8 + 4 + c]; };
        __shared__ double db[P], A[P * P], lam[P], ws[P * P + P];
        if (DIM == 2) {
            db[0] = xb(0, 3);
            db[1] = xb(1, 3);
            db[2] = xb(1, 0) - xb(0, 1);
        } else {
            db[0] = xb(0, 3);
            db[1] = xb(1, 3);
            db[2] = xb(2, 3);
            db[P - 3] = xb(2, 1) - xb(1, 2);
            db[P - 2] = xb(0, 2) - xb(2, 0);
            db[P - 1] = xb(1, 0) - xb(0, 1);
        }
        for (int q = 0; q < P * P; ++q) A[q] = Amat[(size_t)b * 36 + q];
        // forward took delta = 0 when A was not positive definite: no dependence then
        chol_solve<P>(A, db, lam, ws);
        for (int q = 0; q < 6; ++q) lam_out[(size_t)b * 6 + q] = (q < P) ? lam[q] : 0.0;
    }
}

// I7 backward, per-point part: dL/dweight and the point path of dL/dT_k.
template <int DIM, int TYPE>
__global__ __launch_bounds__(ACC_THREADS) void icp_bwd_point_kernel(
    const float *__restrict__ src, const float *__restrict__ tgt, int tgt_cols,
    const float *__restrict__ weight, const float *__restrict__ Tk, const int32_t *__restrict__ active,
    const int32_t *__restrict__ idx, const double *__restrict__ lam64, const double *__restrict__ delta64, int N,
    int M, int loss, float k, float k2, float trim2, float *__restrict__ gw, double *__restrict__ pparts)
{
    constexpr int P = PointTerms<DIM, TYPE>::P;
    constexpr int NR = PointTerms<DIM, TYPE>::NR;
    constexpr int NP = npose(DIM);
    const int b = blockIdx.y;
    const int i = blockIdx.x * ACC_THREADS + threadIdx.x;
    double acc[NP];
#pragma unroll
    for (int a = 0; a < NP; ++a) acc[a] = 0.0;
    const bool act = active[b] != 0;

    if (act && i < N) {
        float T[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) T[q] = Tk[(size_t)b * 16 + q];
        float lam[P], dl[P];
#pragma unroll
        for (int q = 0; q < P; ++q) {
            lam[q] = (float)lam64[(size_t)b * 6 + q];
            dl[q] = (float)delta64[(size_t)b * 6 + q];
        }
        const int j = idx[(size_t)b * N + i];
        const float omega = weight ? weight[(size_t)b * N + i] : 1.f;
        PointTerms<DIM, TYPE> t;
        point_terms<DIM, TYPE>(t, T, src + ((size_t)b * N + i) * 3, tgt + ((size_t)b * M + j) * tgt_cols, omega,
                               loss, k, k2, trim2);
        float wbar = 0.f;
        float ebar[NR];
        float Jbar[NR][P];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            float u = 0.f, v = 0.f;
#pragma unroll
            for (int c = 0; c < P; ++c) {
                u += t.J[r][c] * lam[c];
                v += t.J[r][c] * dl[c];
            }
            const float res = t.er[r] - v;
            wbar += u * res;
            ebar[r] = t.w * u;
#pragma unroll
            for (int c = 0; c < P; ++c) Jbar[r][c] = t.w * (res * lam[c] - u * dl[c]);
        }
        // w = (omega*keep)*rho
        const float gomega = wbar * t.keep * t.rho;
        const float rhobar = wbar * t.omega * t.keep;
        float r2bar = 0.f;
        if (loss == MMK_LOSS_CAUCHY) {
            r2bar = rhobar * (-(t.rho * t.rho) / k2);
        } else if (loss == MMK_LOSS_HUBER) {
            if (sqrtf(t.r2) > k) r2bar = rhobar * (-t.rho / (2.f * t.r2));
        }
#pragma unroll
        for (int r = 0; r < NR; ++r) ebar[r] += 2.f * t.er[r] * r2bar;
        gw[(size_t)b * N + i] += gomega;

        float pbar[DIM];
#pragma unroll
        for (int c = 0; c < DIM; ++c) pbar[c] = 0.f;
        if (TYPE == MMK_ICP_PT2PT) {
#pragma unroll
            for (int c = 0; c < DIM; ++c) pbar[c] -= ebar[c % NR];
            if (DIM == 2) {
                pbar[1] -= Jbar[0][2];
                pbar[0] += Jbar[NR - 1][2];
            } else {
                pbar[DIM - 1] += Jbar[0][P - 2];
                pbar[1] -= Jbar[0][P - 1];
                pbar[DIM - 1] -= Jbar[1 % NR][P - 3];
                pbar[0] += Jbar[1 % NR][P - 1];
                pbar[1] += Jbar[2 % NR][P - 3];
                pbar[0] -= Jbar[2 % NR][P - 2];
            }
        } else {
#pragma unroll
            for (int c = 0; c < DIM; ++c) pbar[c] -= t.n[c] * ebar[0];
            if (DIM == 2) {
                pbar[0] += t.n[1] * Jbar[0][2];
                pbar[1] -= t.n[0] * Jbar[0][2];
            } else {
                const float nx = t.n[0], ny = t.n[1], nz = t.n[DIM - 1];
                const float j3 = Jbar[0][P - 3], j4 = Jbar[0][P - 2], j5 = Jbar[0][P - 1];
                pbar[0] += -nz * j4 + ny * j5;
                pbar[1] += nz * j3 - nx * j5;
                pbar[DIM - 1] += -ny * j3 + nx * j4;
            }
        }
        // p = R s + t
#pragma unroll
        for (int r = 0; r < DIM; ++r) {
#pragma unroll
            for (int c = 0; c < DIM; ++c) acc[r * DIM + c] = (double)pbar[r] * (double)t.s[c];
            acc[DIM * DIM + r] = (double)pbar[r];
        }
    }

    __shared__ double red[ACC_THREADS / 64][NP];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int a = 0; a < NP; ++a) {
        double v = wave_sum(acc[a]);
        if (lane == 0) red[wv][a] = v;
    }
    __syncthreads();
    if (threadIdx.x < NP) {
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < ACC_THREADS / 64; ++w) v += red[w][threadIdx.x];
        pparts[((size_t)b * gridDim.x + blockIdx.x) * NP + threadIdx.x] = v;
    }
}

__global__ void f32_to_f64_kernel(const float *in, double *out, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (double)in[i];
}

template <int DIM>
__global__ void bwd_final_kernel(const double *__restrict__ Gdir, const double *__restrict__ pparts, int nparts,
                                 float *__restrict__ out)
{
    constexpr int NP = npose(DIM);
    __shared__ double stage[PART_ROWS * NP];
    const int b = blockIdx.x, lane = threadIdx.x;
    double v = (lane < 16) ? Gdir[(size_t)b * 16 + lane] : 0.0;
    const int r = lane >> 2, c = lane & 3;
    int slot = -1;
    if (lane < 16 && r < DIM && c < DIM) slot = r * DIM + c;
    if (lane < 16 && r < DIM && c == 3) slot = DIM * DIM + r;
    const double *base = pparts + (size_t)b * nparts * NP;
    for (int r0 = 0; r0 < nparts; r0 += PART_ROWS) {
        const int nr = min(PART_ROWS, nparts - r0);
        for (int e = lane; e < nr * NP; e += blockDim.x) stage[e] = base[(size_t)r0 * NP + e];
        __syncthreads();
        if (slot >= 0)
            for (int q = 0; q < nr; ++q) v += stage[q * NP + slot];
        __syncthreads();
    }
    if (lane < 16) out[(size_t)b * 16 + lane] = (float)v;
}

// ------------------------------------------------------------------------------------------
struct NNPlan {
    int Mpad, ntiles, nsb, ntu, tiles_per_unit, total_units, grid, P, ucap;
};


NNPlan nn_plan(int B, int N, int M, int dim)
{
    NNPlan pl;
    pl.P = 2;                                   // points per lane of the vector-pipe filter = 512-point source blocks for both kernels
    pl.Mpad = (int)mmk::align_up((size_t)M, NN_TILE);
    pl.ntiles = pl.Mpad / NN_TILE;
    pl.nsb = (N + NN_THREADS * pl.P - 1) / (NN_THREADS * pl.P);
    // Vector-pipe filter: single tiles -- short ranges balance the CUs better, and a range without candidates below the starting
    // bound costs no atomic (5 / 3 / 2 / 1 tiles per unit = 159 / 154 / 147 / 141 us per launch).  Matrix-core filter: a lane's
    // running bound restarts with every unit, and without a seed (the first ICP iteration) every restart flags a "record"
    // sequence of chunks that the exact part then re-scans: 324 / 170 us for the first / a later launch at 1 tile per unit,
    // 265 / 162 at 2, 226 / 157 at 4 (round-3 build before the last tuning; the final kernel: 314 / 148 at 1 tile, 257 / 143 at 2,
    // 195 / 133 at 4, 191 / 141 at 5, 210 / 163 at 10 -- each unit start costs two dependent gathers for the seed).
    pl.tiles_per_unit = std::min(4, pl.ntiles);
    pl.ntu = (pl.ntiles + pl.tiles_per_unit - 1) / pl.tiles_per_unit;
    const int Bpad = (B + 7) / 8 * 8;
    pl.total_units = Bpad * pl.nsb * pl.ntu;
    pl.ucap = (Bpad / 8) * pl.nsb;             // entries per class of the scanned-block lists (src_units_kernel)
    // persistent grid: 8 blocks of 256 threads per CU on the 256 CUs (LDS 8-12 KB, <= 64 VGPRs); 3 of the matrix-core
    // kernel's (40 KB of LDS each, <= 128 VGPRs)
    pl.grid = std::min(pl.total_units, dim == 2 ? 1024 : 768);
    return pl;
}

// Optional in-library timing of the NN kernel (bench.py's roofline leg): a pool of HIP
// event pairs recorded on the launch stream around each nn_search launch.
struct NNProf {
    bool on = false;
    int cap = 0, n = 0;
    hipEvent_t *ev = nullptr;  // 2 * cap
};
thread_local NNProf g_prof;      // per host thread: the library keeps no process-global mutable state

// `packed` (B,N) must hold NN_KEY_INIT on entry (memset 0xFF or re-armed by the accumulate kernel).
// `seed_buf` ((B,N) int32, may be null): where the coarse pass puts the seeds of a launch without previous correspondents.
int launch_nn(int dim, const float *src, const float *tgtp, const float *Tk, const int32_t *active,
              const int32_t *prev_idx, const int32_t *ulist, const int32_t *ucnt, int B, int N, const NNPlan &pl,
              unsigned long long *packed, int32_t *seed_buf, const int32_t *allzero, const int32_t *zrep, const float *dec,
              hipStream_t st)
{
    const bool rec = g_prof.on && g_prof.n < g_prof.cap;
    if (rec) MMK_CHECK_HIP(hipEventRecord(g_prof.ev[2 * g_prof.n], st));
    if (prev_idx == nullptr && seed_buf != nullptr && pl.Mpad >= 4 * NN_COARSE_STRIDE) {
        const dim3 grid((N + NN_COARSE_THREADS - 1) / NN_COARSE_THREADS, B);
        const int nn_pts = NN_THREADS * pl.P;
        if (dim == 2)
            hipLaunchKernelGGL(nn_coarse_seed_kernel<2>, grid, dim3(NN_COARSE_THREADS), 0, st, src, tgtp, Tk, active, allzero, zrep, nn_pts, pl.nsb, N,
                               pl.Mpad, dec, seed_buf);
        else
            hipLaunchKernelGGL(nn_coarse_seed_kernel<3>, grid, dim3(NN_COARSE_THREADS), 0, st, src, tgtp, Tk, active, allzero, zrep, nn_pts, pl.nsb, N,
                               pl.Mpad, dec, seed_buf);
        MMK_LAUNCH_CHECK();
        prev_idx = seed_buf;
    }
    constexpr float U = 5.9604645e-8f;       // 2^-24
    if (dim == 2) {
        hipLaunchKernelGGL((nn_mfma_kernel<2>), dim3(pl.grid), dim3(NN_THREADS), 0, st, src, tgtp, Tk, active, prev_idx, ulist, ucnt,
                           pl.ucap, B, N, pl.Mpad, pl.nsb, pl.ntu, pl.tiles_per_unit, pl.total_units, 360.0f * U, packed);
    } else {
        hipLaunchKernelGGL((nn_mfma_kernel<3>), dim3(pl.grid), dim3(NN_THREADS), 0, st, src, tgtp, Tk, active, prev_idx, ulist, ucnt,
                           pl.ucap, B, N, pl.Mpad, pl.nsb, pl.ntu, pl.tiles_per_unit, pl.total_units, 496.0f * U, packed);
    }
    MMK_LAUNCH_CHECK();
    if (rec) {
        MMK_CHECK_HIP(hipEventRecord(g_prof.ev[2 * g_prof.n + 1], st));
        g_prof.n++;
    }
    return MMK_OK;
}

int check_params(const mmk_icp_params *p)
{
    MMK_REQUIRE(p != nullptr, "mmk_icp: params is NULL");
    MMK_REQUIRE(p->B >= 1 && p->N >= 1 && p->M >= 1, "mmk_icp: B, N, M must be >= 1 (got %d, %d, %d)", p->B, p->N, p->M);
    MMK_REQUIRE(p->dim == 2 || p->dim == 3, "mmk_icp: dim must be 2 or 3 (got %d)", p->dim);
    MMK_REQUIRE(p->tgt_cols == 3 || p->tgt_cols == 6, "mmk_icp: target must have 3 or 6 columns (got %d)", p->tgt_cols);
    MMK_REQUIRE(p->icp_type == MMK_ICP_PT2PT || p->icp_type == MMK_ICP_PT2PL, "mmk_icp: bad icp_type %d", p->icp_type);
    MMK_REQUIRE(p->icp_type != MMK_ICP_PT2PL || p->tgt_cols == 6, "mmk_icp: pt2pl needs target normals (B,M,6)");
    MMK_REQUIRE(p->loss >= MMK_LOSS_NONE && p->loss <= MMK_LOSS_HUBER, "mmk_icp: bad loss %d", p->loss);
    MMK_REQUIRE(p->max_iter >= 1, "mmk_icp: max_iter must be >= 1");
    MMK_REQUIRE(p->loss == MMK_LOSS_NONE || p->loss_k > 0.f, "mmk_icp: loss metric must be > 0");
    MMK_REQUIRE(p->nn_method == MMK_NN_BRUTE || p->nn_method == MMK_NN_GRID, "mmk_icp: bad nn_method %d", p->nn_method);
    return MMK_OK;
}

struct IcpWs {
    float *tgtp;
    unsigned long long *packed;  // 2 x (B,N) NN keys: read by iteration k's accumulate, armed for iteration k+1
    int32_t *allzero, *zrep;     // (B,nsb), (B): zero-row bookkeeping of the source (src_zero_scan_kernel)
    int32_t *ulist, *ucnt;       // (8,ucap), (16): the source blocks the NN kernel scans (src_units_kernel)
    int32_t *seed0;              // (B,N): coarse seeds of the first iteration (nn_coarse_seed_kernel)
    float *tdec;                 // (B,dim,Mpad/NN_COARSE_STRIDE): every 64th target, densely (its samples)
    double *partials;   // forward: (B,nblk,NACC); backward: pose parts (B,nblk,NP)
    double *G0, *G1;    // backward (B,16)
    double *lam;        // backward (B,6)
    int32_t *status;             // (16): word 0 = MMK_ICP_STATUS_* bits raised by the kernels of the last forward call
    int32_t *g_counts, *g_cursor, *g_starts;  // grid NN: (B,NC), (B,NC), (B,NC+1)
    float4 *g_sorted;                          // grid NN: (B,M) x,y,z,index
    size_t bytes;
};

IcpWs carve(const mmk_icp_params *p, void *ws, size_t cap)
{
    const NNPlan pl = nn_plan(p->B, p->N, p->M, p->dim);
    const int nblk = (p->N + ACC_THREADS - 1) / ACC_THREADS;
    mmk::Arena ar(ws, cap);
    IcpWs w;
    w.tgtp = ar.take<float>((size_t)p->B * p->dim * pl.Mpad);
    w.packed = ar.take<unsigned long long>((size_t)2 * p->B * p->N);
    w.allzero = ar.take<int32_t>((size_t)p->B * pl.nsb);
    w.zrep = ar.take<int32_t>((size_t)p->B);
    w.ulist = ar.take<int32_t>((size_t)8 * pl.ucap);
    w.ucnt = ar.take<int32_t>(16);
    w.seed0 = ar.take<int32_t>((size_t)p->B * p->N);
    w.tdec = ar.take<float>((size_t)p->B * p->dim * (pl.Mpad / NN_COARSE_STRIDE));
    w.partials = ar.take<double>((size_t)p->B * nblk * 27);
    w.G0 = ar.take<double>((size_t)p->B * 16);
    w.G1 = ar.take<double>((size_t)p->B * 16);
    w.lam = ar.take<double>((size_t)p->B * 6);
    w.status = ar.take<int32_t>(16);
    w.g_counts = w.g_cursor = w.g_starts = nullptr;
    w.g_sorted = nullptr;
    if (p->nn_method == MMK_NN_GRID) {
        w.g_counts = ar.take<int32_t>((size_t)p->B * GRID_NC * 2);
        w.g_cursor = w.g_counts + (size_t)p->B * GRID_NC;
        w.g_starts = ar.take<int32_t>((size_t)p->B * (GRID_NC + 1));
        w.g_sorted = ar.take<float4>((size_t)p->B * p->M);
    }
    w.bytes = mmk::align_up(ar.off, 256);
    return w;
}

template <int DIM, int TYPE>
int run_forward(const mmk_icp_params *p, const float *src, const float *tgt, const float *weight, float *T_hist,
                int32_t *idx_hist, double *delta_hist, double *A_hist, int32_t *active_hist, const IcpWs &w,
                int *iters_run, hipStream_t st)
{
    const int B = p->B, N = p->N, M = p->M;
    const NNPlan pl = nn_plan(B, N, M, DIM);
    const int nblk = (N + ACC_THREADS - 1) / ACC_THREADS;
    const float k = p->loss_k, k2 = p->loss_k * p->loss_k, trim2 = p->trim_dist * p->trim_dist;
    int k_done = 0;
    // (keys, status word, zero-row representatives: icp_init_kernel in mmk_icp_forward)
    const bool use_grid = p->nn_method == MMK_NN_GRID;
    const int nn_pts = NN_THREADS * pl.P;
    const bool dedup = !use_grid && pl.nsb <= 0xfff && B < (1 << 19);
    if (dedup) {
        hipLaunchKernelGGL(src_zero_scan_kernel, dim3(pl.nsb, B), dim3(NN_THREADS), 0, st, src, N, nn_pts, pl.nsb, w.allzero, w.zrep);
        MMK_LAUNCH_CHECK();
        hipLaunchKernelGGL(src_units_kernel, dim3(1), dim3(64), 0, st, w.allzero, w.zrep, B, pl.nsb, nn_pts, pl.ucap, w.ulist, w.ucnt);
        MMK_LAUNCH_CHECK();
    }
    const int32_t *az = dedup ? w.allzero : nullptr, *zr = dedup ? w.zrep : nullptr;
    if (use_grid) {   // the target is fixed over the iterations: bin it once
        MMK_CHECK_HIP(hipMemsetAsync(w.g_counts, 0, sizeof(int32_t) * (size_t)B * GRID_NC * 2, st));
        hipLaunchKernelGGL(grid_count_kernel, dim3((M + 255) / 256, B), dim3(256), 0, st, tgt, M, p->tgt_cols, w.g_counts);
        MMK_LAUNCH_CHECK();
        hipLaunchKernelGGL(grid_scan_kernel, dim3(B), dim3(256), 0, st, w.g_counts, w.g_starts);
        MMK_LAUNCH_CHECK();
        hipLaunchKernelGGL(grid_fill_kernel, dim3((M + 255) / 256, B), dim3(256), 0, st, tgt, M, p->tgt_cols, DIM, w.g_starts,
                           w.g_cursor, w.g_sorted);
        MMK_LAUNCH_CHECK();
    }
    for (int it = 0; it < p->max_iter; ++it) {
        const float *Tk = T_hist + (size_t)it * B * 16;
        const int32_t *act = active_hist + (size_t)it * B;
        int32_t *idx = idx_hist + (p->save_state ? (size_t)it * B * N : 0);
        unsigned long long *keys = w.packed + (size_t)(it & 1) * B * N, *keys_next = w.packed + (size_t)((it + 1) & 1) * B * N;
        if (use_grid) {
            const bool rec = g_prof.on && g_prof.n < g_prof.cap;
            if (rec) MMK_CHECK_HIP(hipEventRecord(g_prof.ev[2 * g_prof.n], st));
            // the correspondences of the previous iteration (still in the index buffer) bound the search
            const int32_t *prev = (it > 0) ? idx_hist + (p->save_state ? (size_t)(it - 1) * B * N : 0) : nullptr;
            hipLaunchKernelGGL(grid_nn_kernel<DIM>, dim3((N + 255) / 256, B), dim3(256), 0, st, src, Tk, act, w.g_starts,
                               w.g_sorted, N, M, tgt, p->tgt_cols, prev, keys);
            MMK_LAUNCH_CHECK();
            if (rec) {
                MMK_CHECK_HIP(hipEventRecord(g_prof.ev[2 * g_prof.n + 1], st));
                g_prof.n++;
            }
        } else {
            // the correspondences of the previous iteration (still in the index buffer) start the filter's bound
            const int32_t *prev = (it > 0) ? idx_hist + (p->save_state ? (size_t)(it - 1) * B * N : 0) : nullptr;
            int rc = launch_nn(DIM, src, w.tgtp, Tk, act, prev, dedup ? w.ulist : nullptr, w.ucnt, B, N, pl, keys, w.seed0, az, zr, w.tdec, st);
            if (rc != MMK_OK) return rc;
        }
        hipLaunchKernelGGL((icp_accumulate_kernel<DIM, TYPE>), dim3(nblk, B), dim3(ACC_THREADS), 0, st, src, tgt,
                           p->tgt_cols, weight, Tk, act, keys, keys_next, az, zr, nn_pts, pl.nsb, idx, N, M, p->loss, k, k2, trim2,
                           w.partials, w.status);
        MMK_LAUNCH_CHECK();
        hipLaunchKernelGGL(icp_solve_kernel<DIM>, dim3(B), dim3(64), 0, st, w.partials, nblk, Tk,
                           T_hist + (size_t)(it + 1) * B * 16, delta_hist + (size_t)it * B * 6,
                           A_hist + (size_t)it * B * 36, act, active_hist + (size_t)(it + 1) * B, p->tolerance);
        MMK_LAUNCH_CHECK();
        k_done = it + 1;
        if (p->check_every > 0 && (k_done % p->check_every) == 0 && k_done < p->max_iter) {
            static thread_local int32_t host_act[1 << 16];
            MMK_REQUIRE(B <= (1 << 16), "mmk_icp_forward: check_every needs B <= 65536");
            MMK_CHECK_HIP(hipMemcpyAsync(host_act, active_hist + (size_t)k_done * B, sizeof(int32_t) * B,
                                         hipMemcpyDeviceToHost, st));
            MMK_CHECK_HIP(hipStreamSynchronize(st));
            bool any = false;
            for (int b = 0; b < B; ++b) any = any || host_act[b] != 0;
            if (!any) break;
        }
    }
    // iterations that were skipped after an early stop: carry the pose forward
    for (int it = k_done; it < p->max_iter; ++it) {
        MMK_CHECK_HIP(hipMemcpyAsync(T_hist + (size_t)(it + 1) * B * 16, T_hist + (size_t)it * B * 16,
                                     sizeof(float) * B * 16, hipMemcpyDeviceToDevice, st));
        MMK_CHECK_HIP(hipMemsetAsync(active_hist + (size_t)(it + 1) * B, 0, sizeof(int32_t) * B, st));
        MMK_CHECK_HIP(hipMemsetAsync(delta_hist + (size_t)it * B * 6, 0, sizeof(double) * B * 6, st));
    }
    if (iters_run) *iters_run = k_done;
    return MMK_OK;
}

template <int DIM, int TYPE>
int run_backward(const mmk_icp_params *p, const float *src, const float *tgt, const float *weight,
                 const int32_t *idx_hist, const float *T_hist, const double *delta_hist, const double *A_hist,
                 const int32_t *active_hist, const float *grad_T, float *gw, float *grad_T_init, const IcpWs &w,
                 hipStream_t st)
{
    const int B = p->B, N = p->N, M = p->M;
    const int nblk = (N + ACC_THREADS - 1) / ACC_THREADS;
    const float k = p->loss_k, k2 = p->loss_k * p->loss_k, trim2 = p->trim_dist * p->trim_dist;
    MMK_CHECK_HIP(hipMemsetAsync(gw, 0, sizeof(float) * (size_t)B * N, st));
    hipLaunchKernelGGL(f32_to_f64_kernel, dim3((B * 16 + 255) / 256), dim3(256), 0, st, grad_T, w.G0, B * 16);
    MMK_LAUNCH_CHECK();
    double *Gin = w.G0, *Gout = w.G1;
    int nparts = 0;
    for (int it = p->max_iter - 1; it >= 0; --it) {
        const float *Tk = T_hist + (size_t)it * B * 16;
        const int32_t *act = active_hist + (size_t)it * B;
        hipLaunchKernelGGL(icp_bwd_pair_kernel<DIM>, dim3(B), dim3(64), 0, st, Gin, w.partials, nparts, Tk,
                           delta_hist + (size_t)it * B * 6, A_hist + (size_t)it * B * 36, act, Gout, w.lam);
        MMK_LAUNCH_CHECK();
        hipLaunchKernelGGL((icp_bwd_point_kernel<DIM, TYPE>), dim3(nblk, B), dim3(ACC_THREADS), 0, st, src, tgt,
                           p->tgt_cols, weight, Tk, act, idx_hist + (size_t)it * B * N, w.lam,
                           delta_hist + (size_t)it * B * 6, N, M, p->loss, k, k2, trim2, gw, w.partials);
        MMK_LAUNCH_CHECK();
        nparts = nblk;
        double *t = Gin; Gin = Gout; Gout = t;
    }
    if (grad_T_init) {
        hipLaunchKernelGGL(bwd_final_kernel<DIM>, dim3(B), dim3(64), 0, st, Gin, w.partials, nparts, grad_T_init);
        MMK_LAUNCH_CHECK();
    }
    return MMK_OK;
}

}  // namespace

// ================================================================================== C ABI
extern "C" int32_t mmk_nn_padded_m(int32_t M) { return (int32_t)mmk::align_up((size_t)std::max(M, 1), NN_TILE); }

extern "C" int mmk_pack_target(const float *target, int32_t B, int32_t M, int32_t tgt_cols, int32_t dim,
                               float *target_planar, void *stream)
{
    MMK_REQUIRE(target && target_planar, "mmk_pack_target: NULL pointer");
    MMK_REQUIRE(B >= 1 && M >= 1, "mmk_pack_target: B, M must be >= 1");
    MMK_REQUIRE((dim == 2 || dim == 3) && tgt_cols >= dim, "mmk_pack_target: dim must be 2|3 and <= tgt_cols");
    const int Mpad = mmk_nn_padded_m(M);
    hipLaunchKernelGGL(pack_target_kernel, dim3(Mpad / 256, B), dim3(256), 0, (hipStream_t)stream, target, M, tgt_cols,
                       dim, Mpad, target_planar, static_cast<float *>(nullptr), 1);
    MMK_LAUNCH_CHECK();
    return MMK_OK;
}

extern "C" size_t mmk_nn_workspace_bytes(int32_t B, int32_t N, int32_t M, int32_t dim)
{
    (void)dim;
    if (B < 1 || N < 1 || M < 1) return 0;
    return mmk::align_up((size_t)B * N * sizeof(unsigned long long), 256) + mmk::align_up((size_t)B * N * sizeof(int32_t), 256) + 512;
}

extern "C" int mmk_nn_search(const float *source, const float *target_planar, const float *T, int32_t B, int32_t N,
                             int32_t M, int32_t dim, int32_t *idx, float *d2, void *workspace, size_t workspace_bytes,
                             void *stream)
{
    MMK_REQUIRE(source && target_planar && T && idx && d2, "mmk_nn_search: NULL pointer");
    MMK_REQUIRE(B >= 1 && N >= 1 && M >= 1, "mmk_nn_search: B, N, M must be >= 1");
    MMK_REQUIRE(dim == 2 || dim == 3, "mmk_nn_search: dim must be 2 or 3");
    const NNPlan pl = nn_plan(B, N, M, dim);
    mmk::Arena ar(workspace, workspace_bytes);
    unsigned long long *packed = ar.take<unsigned long long>((size_t)B * N);
    int32_t *seed0 = ar.take<int32_t>((size_t)B * N);
    if (!ar.ok() || workspace == nullptr) {
        mmk::set_error("mmk_nn_search: workspace too small (%zu < %zu)", workspace_bytes, ar.off);
        return MMK_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    MMK_CHECK_HIP(hipMemsetAsync(packed, 0xFF, sizeof(unsigned long long) * (size_t)B * N, st));
    int rc = launch_nn(dim, source, target_planar, T, nullptr, nullptr, nullptr, nullptr, B, N, pl, packed, seed0, nullptr, nullptr, nullptr, st);
    if (rc != MMK_OK) return rc;
    hipLaunchKernelGGL(nn_unpack_kernel, dim3((B * N + 255) / 256), dim3(256), 0, st, packed, B * N, idx, d2);
    MMK_LAUNCH_CHECK();
    return MMK_OK;
}

extern "C" int mmk_nn_profile_begin(int32_t capacity)
{
    MMK_REQUIRE(capacity >= 1 && capacity <= (1 << 20), "mmk_nn_profile_begin: bad capacity %d", capacity);
    if (g_prof.cap < capacity) {
        for (int i = 0; i < 2 * g_prof.cap; ++i) (void)hipEventDestroy(g_prof.ev[i]);
        delete[] g_prof.ev;
        g_prof.ev = new hipEvent_t[2 * (size_t)capacity];
        g_prof.cap = 0;
        for (int i = 0; i < 2 * capacity; ++i) MMK_CHECK_HIP(hipEventCreate(&g_prof.ev[i]));
        g_prof.cap = capacity;
    }
    g_prof.n = 0;
    g_prof.on = true;
    return MMK_OK;
}

extern "C" int mmk_nn_profile_end(float *ms_out, int32_t max_out, int32_t *n_out)
{
    MMK_REQUIRE(n_out != nullptr, "mmk_nn_profile_end: NULL n_out");
    g_prof.on = false;
    int n = g_prof.n;
    if (n > max_out) n = max_out;
    for (int i = 0; i < n; ++i) {
        MMK_CHECK_HIP(hipEventSynchronize(g_prof.ev[2 * i + 1]));
        MMK_CHECK_HIP(hipEventElapsedTime(&ms_out[i], g_prof.ev[2 * i], g_prof.ev[2 * i + 1]));
    }
    *n_out = g_prof.n;
    g_prof.n = 0;
    return MMK_OK;
}

extern "C" size_t mmk_icp_workspace_bytes(const mmk_icp_params *p)
{
    if (check_params(p) != MMK_OK) return 0;
    return carve(p, nullptr, 0).bytes;
}

extern "C" int mmk_icp_forward(const mmk_icp_params *p, const float *source, const float *target,
                               const float *weight, const float *T_init, float *T_out, int32_t *idx_hist,
                               float *T_hist, double *delta_hist, double *A_hist, int32_t *active_hist,
                               void *workspace, size_t workspace_bytes, int *iters_run, void *stream)
{
    int rc = check_params(p);
    if (rc != MMK_OK) return rc;
    MMK_REQUIRE(source && target && T_init && T_out && idx_hist && T_hist && delta_hist && A_hist && active_hist,
                "mmk_icp_forward: NULL pointer");
    const IcpWs w = carve(p, workspace, workspace_bytes);
    if (workspace == nullptr || w.bytes > workspace_bytes) {
        mmk::set_error("mmk_icp_forward: workspace too small (%zu < %zu)", workspace_bytes, w.bytes);
        return MMK_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const int B = p->B;
    {
        const int Mpad = mmk_nn_padded_m(p->M);
        hipLaunchKernelGGL(pack_target_kernel, dim3(Mpad / 256, B), dim3(256), 0, st, target, p->M, p->tgt_cols, p->dim, Mpad, w.tgtp,
                           w.tdec, NN_COARSE_STRIDE);
        MMK_LAUNCH_CHECK();
    }
    {
        const size_t nkeys = (size_t)2 * B * p->N;
        const unsigned blocks = (unsigned)std::min<size_t>(std::max<size_t>((nkeys + 255) / 256, (size_t)(B * 16 + 255) / 256), 2048);
        hipLaunchKernelGGL(icp_init_kernel, dim3(blocks), dim3(256), 0, st, w.packed, nkeys, w.status, active_hist, T_init, T_hist, w.zrep, B,
                           p->N);
        MMK_LAUNCH_CHECK();
    }
    if (p->dim == 2 && p->icp_type == MMK_ICP_PT2PT)
        rc = run_forward<2, MMK_ICP_PT2PT>(p, source, target, weight, T_hist, idx_hist, delta_hist, A_hist, active_hist, w, iters_run, st);
    else if (p->dim == 2)
        rc = run_forward<2, MMK_ICP_PT2PL>(p, source, target, weight, T_hist, idx_hist, delta_hist, A_hist, active_hist, w, iters_run, st);
    else if (p->icp_type == MMK_ICP_PT2PT)
        rc = run_forward<3, MMK_ICP_PT2PT>(p, source, target, weight, T_hist, idx_hist, delta_hist, A_hist, active_hist, w, iters_run, st);
    else
        rc = run_forward<3, MMK_ICP_PT2PL>(p, source, target, weight, T_hist, idx_hist, delta_hist, A_hist, active_hist, w, iters_run, st);
    if (rc != MMK_OK) return rc;
    MMK_CHECK_HIP(hipMemcpyAsync(T_out, T_hist + (size_t)p->max_iter * B * 16, sizeof(float) * B * 16,
                                 hipMemcpyDeviceToDevice, st));
    return MMK_OK;
}

extern "C" int mmk_icp_status(const mmk_icp_params *p, const void *workspace, size_t workspace_bytes, int32_t *status_out,
                              void *stream)
{
    int rc = check_params(p);
    if (rc != MMK_OK) return rc;
    MMK_REQUIRE(status_out != nullptr, "mmk_icp_status: NULL status_out");
    const IcpWs w = carve(p, const_cast<void *>(workspace), workspace_bytes);
    if (workspace == nullptr || w.bytes > workspace_bytes) {
        mmk::set_error("mmk_icp_status: workspace too small (%zu < %zu)", workspace_bytes, w.bytes);
        return MMK_ERR_WORKSPACE;
    }
    MMK_CHECK_HIP(hipMemcpyAsync(status_out, w.status, sizeof(int32_t), hipMemcpyDefault, (hipStream_t)stream));
    return MMK_OK;
}

extern "C" size_t mmk_icp_partials_count(const mmk_icp_params *p)
{
    if (check_params(p) != MMK_OK) return 0;
    return (size_t)p->B * ((p->N + ACC_THREADS - 1) / ACC_THREADS) * nacc(p->dim);
}

namespace {
template <int DIM, int TYPE>
int run_accumulate(const mmk_icp_params *p, const float *src, const float *tgt, const float *weight, const float *T,
                   const unsigned long long *keys, int32_t *idx, double *partials, int32_t *status, hipStream_t st)
{
    const int nblk = (p->N + ACC_THREADS - 1) / ACC_THREADS;
    hipLaunchKernelGGL((icp_accumulate_kernel<DIM, TYPE>), dim3(nblk, p->B), dim3(ACC_THREADS), 0, st, src, tgt, p->tgt_cols, weight,
                       T, static_cast<const int32_t *>(nullptr), keys, static_cast<unsigned long long *>(nullptr),
                       static_cast<const int32_t *>(nullptr), static_cast<const int32_t *>(nullptr), 1, 1, idx, p->N, p->M, p->loss,
                       p->loss_k, p->loss_k * p->loss_k, p->trim_dist * p->trim_dist, partials, status);
    MMK_LAUNCH_CHECK();
    return MMK_OK;
}

template <int DIM>
int run_solve_update(const mmk_icp_params *p, const double *partials, const float *T_in, float *T_out, double *delta, double *A,
                     const int32_t *active_in, int32_t *active_out, hipStream_t st)
{
    const int nblk = (p->N + ACC_THREADS - 1) / ACC_THREADS;
    hipLaunchKernelGGL(icp_solve_kernel<DIM>, dim3(p->B), dim3(64), 0, st, partials, nblk, T_in, T_out, delta, A, active_in, active_out,
                       p->tolerance);
    MMK_LAUNCH_CHECK();
    return MMK_OK;
}
}  // namespace

extern "C" int mmk_icp_accumulate(const mmk_icp_params *p, const float *source, const float *target, const float *weight,
                                  const float *T, const uint64_t *nn_keys, int32_t *idx_out, double *partials, int32_t *status,
                                  void *stream)
{
    int rc = check_params(p);
    if (rc != MMK_OK) return rc;
    MMK_REQUIRE(source && target && T && nn_keys && idx_out && partials && status, "mmk_icp_accumulate: NULL pointer");
    hipStream_t st = (hipStream_t)stream;
    const unsigned long long *keys = reinterpret_cast<const unsigned long long *>(nn_keys);
    if (p->dim == 2 && p->icp_type == MMK_ICP_PT2PT) return run_accumulate<2, MMK_ICP_PT2PT>(p, source, target, weight, T, keys, idx_out, partials, status, st);
    if (p->dim == 2) return run_accumulate<2, MMK_ICP_PT2PL>(p, source, target, weight, T, keys, idx_out, partials, status, st);
    if (p->icp_type == MMK_ICP_PT2PT) return run_accumulate<3, MMK_ICP_PT2PT>(p, source, target, weight, T, keys, idx_out, partials, status, st);
    return run_accumulate<3, MMK_ICP_PT2PL>(p, source, target, weight, T, keys, idx_out, partials, status, st);
}

extern "C" int mmk_icp_solve_update(const mmk_icp_params *p, const double *partials, const float *T_in, float *T_out,
                                    double *delta_out, double *A_out, const int32_t *active_in, int32_t *active_out, void *stream)
{
    int rc = check_params(p);
    if (rc != MMK_OK) return rc;
    MMK_REQUIRE(partials && T_in && T_out && delta_out && A_out && active_in && active_out, "mmk_icp_solve_update: NULL pointer");
    if (p->dim == 2) return run_solve_update<2>(p, partials, T_in, T_out, delta_out, A_out, active_in, active_out, (hipStream_t)stream);
    return run_solve_update<3>(p, partials, T_in, T_out, delta_out, A_out, active_in, active_out, (hipStream_t)stream);
}

extern "C" int mmk_icp_backward(const mmk_icp_params *p, const float *source, const float *target,
                                const float *weight, const int32_t *idx_hist, const float *T_hist,
                                const double *delta_hist, const double *A_hist, const int32_t *active_hist,
                                const float *grad_T, float *grad_weight, float *grad_T_init, void *workspace,
                                size_t workspace_bytes, void *stream)
{
    int rc = check_params(p);
    if (rc != MMK_OK) return rc;
    MMK_REQUIRE(p->save_state, "mmk_icp_backward: forward must have run with save_state = 1");
    MMK_REQUIRE(source && target && idx_hist && T_hist && delta_hist && A_hist && active_hist && grad_T && grad_weight,
                "mmk_icp_backward: NULL pointer");
    const IcpWs w = carve(p, workspace, workspace_bytes);
    if (workspace == nullptr || w.bytes > workspace_bytes) {
        mmk::set_error("mmk_icp_backward: workspace too small (%zu < %zu)", workspace_bytes, w.bytes);
        return MMK_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    if (p->dim == 2 && p->icp_type == MMK_ICP_PT2PT)
        return run_backward<2, MMK_ICP_PT2PT>(p, source, target, weight, idx_hist, T_hist, delta_hist, A_hist, active_hist, grad_T, grad_weight, grad_T_init, w, st);
    if (p->dim == 2)
        return run_backward<2, MMK_ICP_PT2PL>(p, source, target, weight, idx_hist, T_hist, delta_hist, A_hist, active_hist, grad_T, grad_weight, grad_T_init, w, st);
    if (p->icp_type == MMK_ICP_PT2PT)
        return run_backward<3, MMK_ICP_PT2PT>(p, source, target, weight, idx_hist, T_hist, delta_hist, A_hist, active_hist, grad_T, grad_weight, grad_T_init, w, st);
    return run_backward<3, MMK_ICP_PT2PL>(p, source, target, weight, idx_hist, T_hist, delta_hist, A_hist, active_hist, grad_T, grad_weight, grad_T_init, w, st);
}
