// Radar image operators of the hot path on gfx950 (HBM-bound byte/float streaming):
// GO-CFAR, blob-centre extraction with stable compaction, polar -> Cartesian bilinear
// resampling, bilinear weight gather / scatter-add, BEV rasterisation.
// Semantics: /root/reference/mm_masking/radar_utils.py (line numbers per kernel);
// CPU restatement: oracle/radar_ref.py; golden vectors: tests/golden/radar_*.npz.
#include <math.h>

#include "mmk_common.h"

namespace {

constexpr int RT = 256;  // threads per block for the row kernels

// Block-wide exclusive scan of one double per thread (RT threads); returns the exclusive
// prefix of `v`, *total gets the block sum.  `sm` holds RT/64 doubles.
template <int NTHR = RT>
__device__ __forceinline__ double block_excl_scan(double v, double *sm, double *total)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        double o = __shfl_up(inc, off, 64);
        if (lane >= off) inc += o;
    }
    if (lane == 63) sm[wv] = inc;
    __syncthreads();
    double base = 0.0, tot = 0.0;
#pragma unroll
    for (int w = 0; w < NTHR / 64; ++w) {
        if (w < wv) base += sm[w];
        tot += sm[w];
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

__device__ __forceinline__ int block_excl_scan_i(int v, int *sm, int *total)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int o = __shfl_up(inc, off, 64);
        if (lane >= off) inc += o;
    }
    if (lane == 63) sm[wv] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < RT / 64; ++w) {
        if (w < wv) base += sm[w];
        tot += sm[w];
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

// ------------------------------------------------------------------------------------------
// R2 cfar_mask (radar_utils.py:29-69).  One block per azimuth row: the row is staged in
// LDS, an fp64 prefix sum gives every 50-cell window sum exactly rounded to fp32, and the
// mask row is written back coalesced: one read + one write of the image.
// (CFAR_T = 512 threads per row: the 40 KB row buffer admits four blocks per CU, and with 256 threads each that was 16 waves per
// CU in a launch that is load -> scan -> compute -> store latency from end to end; 32 waves hide twice as much of it)
constexpr int CFAR_T = 512;
__global__ __launch_bounds__(CFAR_T) void cfar_mask_kernel(const float *__restrict__ raw, int R, int w2, int guard,
                                                       int mincol, int maxcol, float a_th, float b_th, int diff,
                                                       float steep, float *__restrict__ mask)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *cs = reinterpret_cast<double *>(smem);                 // R + 1
    float *row = reinterpret_cast<float *>(cs + (R + 1));          // R
    __shared__ double wsum[CFAR_T / 64];
    const size_t base = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * R;
    // (16-byte loads / stores were tried in round 4: no change, 107 -> 112 us -- the kernel is bound by the latency of one row per
    // block at four blocks per CU (40 KB of LDS each), not by the width of its accesses)
    for (int c = threadIdx.x; c < R; c += CFAR_T) row[c] = raw[base + c];
    __syncthreads();
    const int L = (R + CFAR_T - 1) / CFAR_T;
    const int c0 = min(R, (int)threadIdx.x * L), c1 = min(R, c0 + L);
    double s = 0.0;
    for (int c = c0; c < c1; ++c) s += (double)row[c];
    double tot;
    double run = block_excl_scan<CFAR_T>(s, wsum, &tot);
    for (int c = c0; c < c1; ++c) {
        cs[c] = run;
        run += (double)row[c];
    }
    if (threadIdx.x == CFAR_T - 1) cs[R] = tot;
    __syncthreads();
    auto cell = [&](int c) -> float {
        float th = 1000.0f;
        if (c >= mincol && c < maxcol) {
            const float left = (float)(cs[c - guard] - cs[c - w2 - guard]);
            const float right = (float)(cs[min(R, c + w2 + guard + 1)] - cs[min(R, c + guard + 1)]);
            const float stat = fmaxf(left, right) / (float)w2;
            th = a_th * stat + b_th;
        }
        const float x = row[c];
        float m;
        if (diff) {
            m = 0.5f * tanhf(steep * (x - th) + 2.5f) + 0.5f;
            m = (fabsf(m) > 0.99f) ? m : 0.0f;
        } else {
            m = (x > th) ? 1.0f : 0.0f;
        }
        return m;
    };
    for (int c = threadIdx.x; c < R; c += CFAR_T) mask[base + c] = cell(c);
}

// The same, persistent (round 5): a block walks rows blockIdx.x, + gridDim.x, ... and holds the NEXT row in registers while it
// scans / thresholds / stores the current one from LDS, so that no row's HBM latency is exposed (the one-row-per-block form is a
// load -> scan -> compute -> store latency chain end to end, at four rows in flight per CU).  Same thread -> cell partition and
// the same fp64 sums as cfar_mask_kernel: bit-identical masks.  NPRE >= ceil(R / CFAR_T).
template <int NPRE>
__global__ __launch_bounds__(CFAR_T) void cfar_mask_rows_kernel(const float *__restrict__ raw, int rows, int R, int w2, int guard,
                                                                int mincol, int maxcol, float a_th, float b_th, int diff,
                                                                float steep, float *__restrict__ mask)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *cs = reinterpret_cast<double *>(smem);                 // R + 1
    float *row = reinterpret_cast<float *>(cs + (R + 1));          // R
    __shared__ double wsum[CFAR_T / 64];
    float nx[NPRE];
    int r = blockIdx.x;
    if (r >= rows) return;
    auto fetch = [&](int rr) {
        const float *src = raw + (size_t)rr * R;
#pragma unroll
        for (int i = 0; i < NPRE; ++i) {
            const int c = threadIdx.x + i * CFAR_T;
            nx[i] = src[c < R ? c : R - 1];
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int i = 0; i < NPRE; ++i) {
            const int c = threadIdx.x + i * CFAR_T;
            if (c < R) row[c] = nx[i];
        }
    };
    fetch(r);
    stage();
    __syncthreads();
    const int L = (R + CFAR_T - 1) / CFAR_T;
    const int c0 = min(R, (int)threadIdx.x * L), c1 = min(R, c0 + L);
    while (true) {
        const int rn = r + gridDim.x;
        fetch(rn < rows ? rn : r);                                  // (unconditional: past the last row the current one again)
        double s = 0.0;
        for (int c = c0; c < c1; ++c) s += (double)row[c];
        double tot;
        double run = block_excl_scan<CFAR_T>(s, wsum, &tot);
        for (int c = c0; c < c1; ++c) {
            cs[c] = run;
            run += (double)row[c];
        }
        if (threadIdx.x == CFAR_T - 1) cs[R] = tot;
        __syncthreads();
        float *out = mask + (size_t)r * R;
        for (int c = threadIdx.x; c < R; c += CFAR_T) {
            float th = 1000.0f;
            if (c >= mincol && c < maxcol) {
                const float left = (float)(cs[c - guard] - cs[c - w2 - guard]);
                const float right = (float)(cs[min(R, c + w2 + guard + 1)] - cs[min(R, c + guard + 1)]);
                const float stat = fmaxf(left, right) / (float)w2;
                th = a_th * stat + b_th;
            }
            const float x = row[c];
            float m;
            if (diff) {
                m = 0.5f * tanhf(steep * (x - th) + 2.5f) + 0.5f;
                m = (fabsf(m) > 0.99f) ? m : 0.0f;
            } else {
                m = (x > th) ? 1.0f : 0.0f;
            }
            out[c] = m;
        }
        if (rn >= rows) break;
        __syncthreads();                                            // everyone is done with this row's LDS image
        stage();
        __syncthreads();
        r = rn;
    }
}

// ------------------------------------------------------------------------------------------
// R3 + R4: mean_peaks_parallel_fast (radar_utils.py:167-185) + extract_pc (:71-106).
__device__ __forceinline__ float peak_value(const float *__restrict__ mrow, int j, int R, float res, int diff,
                                            float steep)
{
    // marker stored at column j (< R-1): arr[j]*z[j+1] + arr[j+1]*z[j]
    const float a0 = (res * (float)j) * mrow[j];
    const float a1 = (res * (float)(j + 1)) * mrow[j + 1];
    float z0, z1;
    if (diff) {
        z0 = 1.0f - tanhf(steep * a0);
        z1 = 1.0f - tanhf(steep * a1);
    } else {
        z0 = (a0 == 0.0f) ? 1.0f : 0.0f;
        z1 = (a1 == 0.0f) ? 1.0f : 0.0f;
    }
    return a0 * z1 + a1 * z0;
}

// (both row kernels stage the mask row in LDS with 16-byte loads -- every cell is needed twice, as mrow[j] and mrow[j + 1] --
// instead of two dword loads per cell)
__device__ __forceinline__ const float *stage_row(const float *__restrict__ grow, int R, float *lrow)
{
    if ((R & 3) == 0 && ((uintptr_t)grow & 15) == 0) {
        for (int c = threadIdx.x * 4; c < R; c += RT * 4) *reinterpret_cast<float4 *>(lrow + c) = *reinterpret_cast<const float4 *>(grow + c);
    } else {
        for (int c = threadIdx.x; c < R; c += RT) lrow[c] = grow[c];
    }
    __syncthreads();
    return lrow;
}

__global__ __launch_bounds__(RT) void peaks_count_kernel(const float *__restrict__ mask, int R, float res, int diff,
                                                         float steep, int32_t *__restrict__ row_count)
{
    extern __shared__ __attribute__((aligned(16))) float lrow_dyn[];
    __shared__ int sm[RT / 64];
    const int rowid = blockIdx.y * gridDim.x + blockIdx.x;
    const float *mrow = stage_row(mask + (size_t)rowid * R, R, lrow_dyn);
    int cnt = 0;
    for (int j = threadIdx.x; j < R - 1; j += RT) cnt += (peak_value(mrow, j, R, res, diff, steep) != 0.0f) ? 1 : 0;
    int tot;
    block_excl_scan_i(cnt, sm, &tot);
    if (threadIdx.x == 0) row_count[rowid] = tot;
}

__global__ __launch_bounds__(RT) void peaks_scan_kernel(const int32_t *__restrict__ row_count, int A,
                                                        int32_t *__restrict__ row_off, int32_t *__restrict__ total)
{
    __shared__ int sm[RT / 64];
    const int b = blockIdx.x;
    const int L = (A + RT - 1) / RT;
    const int a0 = min(A, (int)threadIdx.x * L), a1 = min(A, a0 + L);
    int s = 0;
    for (int a = a0; a < a1; ++a) s += row_count[b * A + a];
    int tot;
    int run = block_excl_scan_i(s, sm, &tot);
    for (int a = a0; a < a1; ++a) {
        row_off[b * A + a] = run;
        run += row_count[b * A + a];
    }
    if (threadIdx.x == 0) total[b] = tot;
}

__global__ __launch_bounds__(RT) void peaks_emit_kernel(const float *__restrict__ mask, int R, float res, int diff,
                                                        float steep, const int32_t *__restrict__ row_off, int cap,
                                                        float *__restrict__ mval, int32_t *__restrict__ mrow_out)
{
    extern __shared__ __attribute__((aligned(16))) float lrow_dyn[];
    __shared__ int sm[RT / 64];
    const int a = blockIdx.x, b = blockIdx.y, A = gridDim.x;
    const int rowid = b * A + a;
    const float *mrow = stage_row(mask + (size_t)rowid * R, R, lrow_dyn);
    // A thread owns a run of consecutive cells: ONE block scan of the runs' marker counts places every marker (the markers
    // of a row keep their column order: run t precedes run t + 1) -- instead of a block scan (two barriers) per 256 cells,
    // thirteen per row of 3 360, in a launch that is latency from end to end.
    const int L = (R - 1 + RT - 1) / RT;
    const int j0 = min(R - 1, (int)threadIdx.x * L), j1 = min(R - 1, j0 + L);
    int cnt = 0;
    for (int j = j0; j < j1; ++j) cnt += (peak_value(mrow, j, R, res, diff, steep) != 0.0f) ? 1 : 0;
    int tot;
    int g = row_off[rowid] + block_excl_scan_i(cnt, sm, &tot);
    if (cnt != 0) {
        for (int j = j0; j < j1; ++j) {
            const float v = peak_value(mrow, j, R, res, diff, steep);
            if (v != 0.0f) {
                if (g < cap) {
                    mval[(size_t)b * cap + g] = v;
                    mrow_out[(size_t)b * cap + g] = a;
                }
                ++g;
            }
        }
    }
}

__global__ void peaks_pair_kernel(const float *__restrict__ mval, const int32_t *__restrict__ mrow, int cap,
                                  const int32_t *__restrict__ total, const float *__restrict__ az,
                                  const float *__restrict__ T_ab, int A, int max_pts, float *__restrict__ out_pc,
                                  int32_t *__restrict__ out_count)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    const int npts = total[b] / 2;
    if (k == 0) out_count[b] = npts;
    if (k >= max_pts) return;
    float x = 0.f, y = 0.f, z = 0.f;
    if (k < npts && 2 * k + 1 < cap) {
        const size_t m = (size_t)b * cap + 2 * k;
        const float rho = (mval[m + 1] + mval[m]) / 2.0f;
        const float phi = (az[b * A + mrow[m + 1]] + az[b * A + mrow[m]]) / 2.0f;
        x = rho * cosf(phi);
        y = rho * sinf(phi);
        if (T_ab) {
            const float *T = T_ab + (size_t)b * 16;
            const float xx = ((T[0] * x + T[1] * y) + T[2] * z) + T[3];
            const float yy = ((T[4] * x + T[5] * y) + T[6] * z) + T[7];
            const float zz = ((T[8] * x + T[9] * y) + T[10] * z) + T[11];
            x = xx; y = yy; z = zz;
        }
    }
    float *o = out_pc + ((size_t)b * max_pts + k) * 3;
    o[0] = x; o[1] = y; o[2] = z;
}

// ------------------------------------------------------------------------------------------
// Bilinear tap fetch with zero padding; `rows` > H means the image is wrap-padded by one
// row at both ends (interpolate_crossover, radar_utils.py:317-319) without materialising it.
__device__ __forceinline__ float tap_polar(const float *__restrict__ img, int A, int R, int yi, int xi, int wrap)
{
    const int rows = wrap ? A + 2 : A;
    if (xi < 0 || xi >= R || yi < 0 || yi >= rows) return 0.0f;
    int r = yi;
    if (wrap) r = (yi == 0) ? (A - 1) : ((yi == A + 1) ? 0 : yi - 1);
    return img[(size_t)r * R + xi];
}

// 8f.3 radar_cartesian_to_polar (radar_utils.py:338-372).  One thread per polar cell; fp64 throughout, as
// the reference (its sampling grid is cast to double at :370, so it only accepts an fp64 image).  sin / cos
// of the azimuths and the range coordinates come from the host (torch's CPU sin / cos / linspace, the
// reference's own library calls: device libm results differ in the last bit); every later operation is an
// IEEE fp64 operation in the reference's order, the four-tap blend the FMA chain of PyTorch's CPU
// grid_sample kernel (oracle/nn_search.c: mmk_oracle_blend4_f64) -> bit-identical output.
__global__ __launch_bounds__(256) void cart_to_polar_kernel(const double *__restrict__ cart, const double *__restrict__ sin_az,
                                                            const double *__restrict__ cos_az, const double *__restrict__ range_coords,
                                                            int A, int R, int H, int W, double cart_resolution,
                                                            double *__restrict__ polar)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const int a = blockIdx.y, b = blockIdx.z;
    const double rc = range_coords[r];
    const double sx = sin_az[(size_t)b * A + a] * rc, sy = cos_az[(size_t)b * A + a] * rc;
    double u = sx / cart_resolution, v = -sy / cart_resolution;
    u = u / (double)(W - 1) * 2.0;
    v = v / (double)(H - 1) * 2.0;
    const double ix = ((u + 1.0) / 2.0) * (double)(W - 1), iy = ((v + 1.0) / 2.0) * (double)(H - 1);
    const double x0 = floor(ix), y0 = floor(iy);
    const double wx = ix - x0, wy = iy - y0, ex = 1.0 - wx, sy1 = 1.0 - wy;
    const long xi = (long)x0, yi = (long)y0;
    const double *img = cart + (size_t)b * H * W;
    auto tap = [&](long y, long x) -> double {
        return (x >= 0 && x < W && y >= 0 && y < H) ? img[(size_t)y * W + x] : 0.0;
    };
    const double t0 = tap(yi, xi), t1 = tap(yi, xi + 1), t2 = tap(yi + 1, xi), t3 = tap(yi + 1, xi + 1);
    polar[((size_t)b * A + a) * R + r] = fma(t3, wy * wx, fma(t2, wy * ex, fma(t1, sy1 * wx, t0 * (sy1 * ex))));
}

// R5 radar_polar_to_cartesian_diff (radar_utils.py:258-336).  One thread per Cartesian
// pixel; the batch item's azimuth table sits in LDS for the binary search (wobble fix).
__global__ __launch_bounds__(256) void polar_to_cart_kernel(const float *__restrict__ polar,
                                                            const float *__restrict__ az,
                                                            const float *__restrict__ rgrid,
                                                            const float *__restrict__ agrid, int A, int R, int W,
                                                            float res, float half_res, int wrap, int fix_wobble,
                                                            float *__restrict__ cart, const float *__restrict__ polar2,
                                                            float *__restrict__ cart2)
{
    extern __shared__ float laz[];
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < A; i += blockDim.x) laz[i] = az[(size_t)b * A + i];
    __syncthreads();
    // a block is a 32 x 8 patch of the Cartesian image (compact footprint in the polar one: the taps of
    // neighbouring lanes share cache lines), not a 256-pixel strip of one row
    const int tiles_x = (W + 31) >> 5;
    const int px = (blockIdx.x % tiles_x) * 32 + (threadIdx.x & 31), py = (blockIdx.x / tiles_x) * 8 + (threadIdx.x >> 5);
    if (px >= W || py >= W) return;
    const int pix = py * W + px;
    const float rng = rgrid[pix], ang = agrid[pix];
    float u = (rng - half_res) / res;
    float v;
    if (fix_wobble) {
        // lower_bound: first i with laz[i] >= ang.  The table is ascending and nearly uniform (the wobble is a fraction of a
        // step), so the search starts from the uniform table's answer and walks: two or three dependent LDS reads instead
        // of the nine of a bisection over 400 entries, the same index for any ascending table (both loops end at the first
        // entry that is not below ang); a table that is far from uniform only costs more steps.
        const float a0 = laz[0];
        const float inv_step = (float)(A - 1) / ((laz[A - 1] - a0) + 1e-30f);
        int lo = (int)fminf(fmaxf((ang - a0) * inv_step, 0.f), (float)(A - 1));
        while (lo > 0 && laz[lo - 1] >= ang) --lo;
        while (lo < A && laz[lo] < ang) ++lo;
        int c3 = lo;
        if (c3 == A) c3 -= 1;
        int c2 = c3 - 1;
        if (c2 < 0) c2 += 1;
        const float a3 = laz[c3], a2 = laz[c2];
        const float df = ang - a3;
        const float delta = ((df * ((df < 0.f) ? 1.f : 0.f)) * ((c3 > 0) ? 1.f : 0.f)) / ((a3 - a2) + 1e-14f);
        v = (float)c3 + delta;
    } else {
        const float step = (laz[A - 1] - laz[0]) / (float)(A - 1);
        v = (ang - laz[0]) / step;
    }
    if (u < 0.f) u = 0.f;
    const int rows = wrap ? A + 2 : A;
    if (wrap) v = v + 1.0f;
    // normalise to [-1,1] and back exactly as F.grid_sample(align_corners=True) does
    const float gx = u / (float)(R - 1) * 2.0f - 1.0f;
    const float gy = v / (float)(rows - 1) * 2.0f - 1.0f;
    const float ix = ((gx + 1.0f) / 2.0f) * (float)(R - 1);
    const float iy = ((gy + 1.0f) / 2.0f) * (float)(rows - 1);
    const float x0 = floorf(ix), y0 = floorf(iy);
    const float wx = ix - x0, wy = iy - y0;
    const float ex = 1.0f - wx, sy = 1.0f - wy;
    const int xi = (int)x0, yi = (int)y0;
    // The gather is bound by the number of load instructions (64 scattered addresses each), not by bytes: the two taps of a
    // row are neighbours in memory, so each row of each image is ONE 8-byte load (4-byte aligned) instead of two 4-byte ones, all
    // four issued back to back; the zero padding is applied as selects afterwards.  Same products, same order of additions as
    // tap_polar() per tap.  (xi >= 0 because u >= 0; xi = R - 1 shifts the pair one cell left and takes its second element.)
    typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
    const int xa = min(max(xi, 0), R - 2);
    const bool shifted = xi > xa;                   // xi == R - 1
    int roff[2];
    bool okr[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int yk = yi + k;
        okr[k] = !(yk < 0 || yk >= rows);
        int r = yk;
        if (wrap) r = (yk == 0) ? (A - 1) : ((yk == A + 1) ? 0 : yk - 1);
        roff[k] = min(max(r, 0), A - 1) * R + xa;
    }
    const bool okx0 = xi >= 0 && xi < R, okx1 = xi + 1 >= 0 && xi + 1 < R;
    const float *img = polar + (size_t)b * A * R;
    const float *img2 = polar2 != nullptr ? polar2 + (size_t)b * A * R : img;
    f32x2u p1[2], p2[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) p1[k] = *reinterpret_cast<const f32x2u *>(img + roff[k]);
#pragma unroll
    for (int k = 0; k < 2; ++k) p2[k] = *reinterpret_cast<const f32x2u *>(img2 + roff[k]);
    float t1[4], t2[4];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        t1[2 * k] = (okr[k] && okx0) ? (shifted ? p1[k].y : p1[k].x) : 0.0f;
        t1[2 * k + 1] = (okr[k] && okx1) ? p1[k].y : 0.0f;
        t2[2 * k] = (okr[k] && okx0) ? (shifted ? p2[k].y : p2[k].x) : 0.0f;
        t2[2 * k + 1] = (okr[k] && okx1) ? p2[k].y : 0.0f;
    }
    cart[(size_t)b * W * W + pix] = t1[0] * (sy * ex) + t1[1] * (sy * wx) + t1[2] * (wy * ex) + t1[3] * (wy * wx);
    // a second image on the same grid (the dataset resamples the FFT and the CFAR image with the same
    // azimuths, icp_weight_dataset.py:350-352): the coordinates and tap weights are shared
    if (polar2 != nullptr) cart2[(size_t)b * W * W + pix] = t2[0] * (sy * ex) + t2[1] * (sy * wx) + t2[2] * (wy * ex) + t2[3] * (wy * wx);
}

// ------------------------------------------------------------------------------------------
// R8 + R9: point_to_cart_idx(min_to_plus_1=True) (radar_utils.py:374-391) feeding the
// bilinear sampler of extract_weights (:108-128).
struct Taps {
    int xi, yi;
    float w00, w01, w10, w11;
};

// cw = cart_pixel_width of point_to_cart_idx: the normalisation is by the Cartesian grid's width
// whatever the mask's own shape is (grid_sample then maps [-1,1] onto the mask's H and W).
__device__ __forceinline__ Taps weight_taps(const float *__restrict__ p, int H, int W, int cw, float cres)
{
    const float x = p[0], y = p[1];
    const bool fake = (x == 0.0f) && (y == 0.0f);
    const float gu = -x / cres;
    const float gv = y / cres;
    float gx = gv / (float)(cw - 1) * 2.0f;
    float gy = gu / (float)(cw - 1) * 2.0f;
    if (fake) {
        gx = -100.0f;
        gy = -100.0f;
    }
    const float ix = ((gx + 1.0f) / 2.0f) * (float)(W - 1);
    const float iy = ((gy + 1.0f) / 2.0f) * (float)(H - 1);
    const float x0 = floorf(ix), y0 = floorf(iy);
    const float wx = ix - x0, wy = iy - y0;
    Taps t;
    t.xi = (int)x0;
    t.yi = (int)y0;
    t.w00 = (1.0f - wy) * (1.0f - wx);
    t.w01 = (1.0f - wy) * wx;
    t.w10 = wy * (1.0f - wx);
    t.w11 = wy * wx;
    return t;
}

__global__ void sample_weights_fwd_kernel(const float *__restrict__ mask, const float *__restrict__ pc, int N,
                                          int cols, int H, int W, int cw, float cres, float *__restrict__ out)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (n >= N) return;
    const Taps t = weight_taps(pc + ((size_t)b * N + n) * cols, H, W, cw, cres);
    const float *m = mask + (size_t)b * H * W;
    auto tap = [&](int yy, int xx) -> float {
        return (xx >= 0 && xx < W && yy >= 0 && yy < H) ? m[(size_t)yy * W + xx] : 0.0f;
    };
    out[(size_t)b * N + n] = tap(t.yi, t.xi) * t.w00 + tap(t.yi, t.xi + 1) * t.w01 + tap(t.yi + 1, t.xi) * t.w10 +
                             tap(t.yi + 1, t.xi + 1) * t.w11;
}

// Backward of the gather = scatter-add of g * w into the four taps of every point, WITHOUT float atomics (their order of
// arrival would make the mask gradient differ from run to run where several taps fall on one pixel: neighbouring
// azimuths near the sensor, neighbouring peaks of one azimuth).  Three passes over the 4 N entries e = 4 n + q of an image:
//   link   every in-image entry pushes itself on its pixel's chain with an INTEGER exchange on the (zero-filled)
//          gradient word itself: head = e + 1, next[e] = previous head - 1 (0 = empty -> -1).  The chain's ORDER
//          depends on arrival; its SET of entries does not.
//   sum    every entry walks its pixel's chain; the entry with the lowest index owns the pixel and adds the chain's
//          values in ascending entry order (repeated selection of the next-larger index: chains are short) -- the order
//          of a sequential loop over points and taps, as PyTorch's CPU grid_sample backward runs it.
//   store  the owners write the sums over the chain heads.
__device__ __forceinline__ int tap_pixel(const Taps &t, int q, int H, int W, float &w)
{
    const int yy = t.yi + (q >> 1), xx = t.xi + (q & 1);
    w = q == 0 ? t.w00 : (q == 1 ? t.w01 : (q == 2 ? t.w10 : t.w11));
    return (xx >= 0 && xx < W && yy >= 0 && yy < H) ? yy * W + xx : -1;
}

__global__ void sample_weights_bwd_link_kernel(const float *__restrict__ gw, const float *__restrict__ pc, int N, int cols,
                                               int H, int W, int cw, float cres, float *__restrict__ gmask,
                                               int *__restrict__ next, float *__restrict__ val, int *__restrict__ pix)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (e >= 4 * N) return;
    const int n = e >> 2, q = e & 3;
    const Taps t = weight_taps(pc + ((size_t)b * N + n) * cols, H, W, cw, cres);
    float w;
    const int p = tap_pixel(t, q, H, W, w);
    const size_t o = (size_t)b * 4 * N + e;
    pix[o] = p;
    if (p < 0) return;
    val[o] = gw[(size_t)b * N + n] * w;
    int *head = reinterpret_cast<int *>(gmask + (size_t)b * H * W + p);
    next[o] = atomicExch(head, e + 1) - 1;
}

__global__ void sample_weights_bwd_sum_kernel(int N, int H, int W, const float *__restrict__ gmask, const int *__restrict__ next,
                                              const float *__restrict__ val, int *__restrict__ pix, float *__restrict__ res)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (e >= 4 * N) return;
    const size_t base = (size_t)b * 4 * N;
    const int p = pix[base + e];
    if (p < 0) return;
    const int head = reinterpret_cast<const int *>(gmask + (size_t)b * H * W)[p] - 1;
    // the owner is the chain's lowest entry index
    int lo = head, len = 0;
    for (int c = head; c >= 0 && len < 4 * N; c = next[base + c], ++len) lo = c < lo ? c : lo;
    if (lo != e) {
        pix[base + e] = -1;                    // not the owner: nothing to store
        return;
    }
    float sum = 0.f;
    int last = -1;
    for (int k = 0; k < len; ++k) {            // the next-larger index, len times: ascending order of the entries
        int pick = 0x7fffffff;
        for (int c = head, j = 0; c >= 0 && j < len; c = next[base + c], ++j)
            if (c > last && c < pick) pick = c;
        sum += val[base + pick];
        last = pick;
    }
    res[base + e] = sum;
}

__global__ void sample_weights_bwd_store_kernel(int N, int H, int W, const int *__restrict__ pix, const float *__restrict__ res,
                                                float *__restrict__ gmask)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (e >= 4 * N) return;
    const int p = pix[(size_t)b * 4 * N + e];
    if (p >= 0) gmask[(size_t)b * H * W + p] = res[(size_t)b * 4 * N + e];
}

// ------------------------------------------------------------------------------------------
// R10 extract_bev_from_pts (radar_utils.py:142-165): idempotent stores of 1.0.
__global__ void bev_raster_kernel(const float *__restrict__ pc, int M, int cols, int W, float cres,
                                  float *__restrict__ bev)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (j >= M) return;
    const float *p = pc + ((size_t)b * M + j) * cols;
    float iu = -p[0] / cres + (float)W / 2.0f;
    float iv = p[1] / cres + (float)W / 2.0f;
    const float mid = (float)(W / 2);
    if (iu < 0.f || iu > (float)(W - 1)) iu = mid;
    if (iv < 0.f || iv > (float)(W - 1)) iv = mid;
    const int uf = (int)floorf(iu), uc = (int)ceilf(iu), vf = (int)floorf(iv), vc = (int)ceilf(iv);
    float *o = bev + (size_t)b * W * W;
    o[(size_t)uc * W + vf] = 1.0f;
    o[(size_t)uc * W + vc] = 1.0f;
    o[(size_t)uf * W + vf] = 1.0f;
    o[(size_t)uf * W + vc] = 1.0f;
}

__global__ void bev_centre_kernel(float *__restrict__ bev, int B, int W)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) bev[(size_t)b * W * W + (size_t)(W / 2) * W + (W / 2)] = 0.0f;
}

// ------------------------------------------------------------------------------------------
// Statistics of extract_weights (radar_utils.py:130-138) and the point count the policy logs
// (icp_weight_policy.py:209-212) in two launches instead of ~25 small reductions: one block per scan
// forms its partial sums in a fixed order, one wave combines the scans in index order (deterministic).
// out[0] = sum_real(0.5 tanh(5 w) + 0.5) / B     (diff_mean_num_non0)
// out[1] = count(w > 0.05 & real) / B            (mean_num_non0)
// out[2] = sum_real(w) / n_real                  (mean_w; NaN when no real point, as torch.mean of nothing)
// out[3] = max_real(w), out[4] = min_real(w)     (-inf / +inf when no real point)
// out[5] = count(x != 0 & y != 0) / B            (mean_all_pts)
// out[6] = n_real
constexpr int WS_NPART = 8;

__global__ __launch_bounds__(256) void weight_stats_partial_kernel(const float *__restrict__ w, const float *__restrict__ pc, int N,
                                                                   int cols, float *__restrict__ part)
{
    __shared__ float red[4][WS_NPART];
    const int b = blockIdx.x;
    float soft = 0.f, cnt = 0.f, sum = 0.f, mx = -INFINITY, mn = INFINITY, nz = 0.f, nr = 0.f;
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        const float *p = pc + ((size_t)b * N + n) * cols;
        const float x = p[0], y = p[1];
        const float v = w[(size_t)b * N + n];
        const bool real = !(x == 0.0f && y == 0.0f);
        if (real) {
            soft += 0.5f * tanhf(5.0f * v) + 0.5f;
            cnt += (v > 0.05f) ? 1.f : 0.f;
            sum += v;
            mx = fmaxf(mx, v);
            mn = fminf(mn, v);
            nr += 1.f;
        }
        nz += (x != 0.0f && y != 0.0f) ? 1.f : 0.f;
    }
    float vals[WS_NPART] = {soft, cnt, sum, mx, mn, nz, nr, 0.f};
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int i = 0; i < WS_NPART; ++i) {
            const float o = __shfl_down(vals[i], off, 64);
            vals[i] = (i == 3) ? fmaxf(vals[i], o) : (i == 4) ? fminf(vals[i], o) : vals[i] + o;
        }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0)
#pragma unroll
        for (int i = 0; i < WS_NPART; ++i) red[wv][i] = vals[i];
    __syncthreads();
    if (threadIdx.x < WS_NPART) {
        const int i = threadIdx.x;
        float t = red[0][i];
        for (int k = 1; k < 4; ++k) t = (i == 3) ? fmaxf(t, red[k][i]) : (i == 4) ? fminf(t, red[k][i]) : t + red[k][i];
        part[(size_t)b * WS_NPART + i] = t;
    }
}

__global__ void weight_stats_final_kernel(const float *__restrict__ part, int B, float *__restrict__ out)
{
    const int i = threadIdx.x;
    if (i >= WS_NPART) return;
    float t = part[i], nr = part[6];
    for (int b = 1; b < B; ++b) {
        const float o = part[(size_t)b * WS_NPART + i];
        t = (i == 3) ? fmaxf(t, o) : (i == 4) ? fminf(t, o) : t + o;
        nr += part[(size_t)b * WS_NPART + 6];
    }
    if (i == 0 || i == 1 || i == 5) t = t / (float)B;
    if (i == 2) t = t / nr;
    out[i] = t;
}

struct PeakWs {
    int32_t *row_count, *row_off, *total, *mrow;
    float *mval;
    int cap;
    size_t bytes;
};

PeakWs carve_peaks(int B, int A, int max_pts, void *ws, size_t cap_bytes)
{
    mmk::Arena ar(ws, cap_bytes);
    PeakWs w;
    w.cap = 2 * max_pts;
    w.row_count = ar.take<int32_t>((size_t)B * A);
    w.row_off = ar.take<int32_t>((size_t)B * A);
    w.total = ar.take<int32_t>((size_t)B);
    w.mrow = ar.take<int32_t>((size_t)B * w.cap);
    w.mval = ar.take<float>((size_t)B * w.cap);
    w.bytes = mmk::align_up(ar.off, 256);
    return w;
}

}  // namespace

// ================================================================================== C ABI
extern "C" int mmk_cfar_mask(const float *raw, int32_t B, int32_t A, int32_t R, int32_t w2, int32_t guard,
                             int32_t mincol, int32_t maxcol, float a_thresh, float b_thresh, int32_t diff,
                             float steep_fact, float *mask, void *stream)
{
    MMK_REQUIRE(raw && mask, "mmk_cfar_mask: NULL pointer");
    MMK_REQUIRE(B >= 1 && A >= 1 && R >= 1, "mmk_cfar_mask: raw_scans must be 3D with non-empty dims");
    MMK_REQUIRE(w2 >= 1 && guard >= 0, "mmk_cfar_mask: bad window (w2=%d guard=%d)", w2, guard);
    MMK_REQUIRE(mincol >= w2 + guard && maxcol <= R, "mmk_cfar_mask: column range [%d,%d) outside the row", mincol, maxcol);
    const size_t smem = (size_t)(R + 1) * sizeof(double) + (size_t)R * sizeof(float);
    MMK_REQUIRE(smem <= 160 * 1024 - 64, "mmk_cfar_mask: R=%d does not fit the 160 KB LDS row buffer", R);
    if (smem > 64 * 1024)
        MMK_CHECK_HIP(hipFuncSetAttribute((const void *)cfar_mask_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    const int rows = A * B;
    if (R <= 8 * CFAR_T && smem <= 40 * 1024) {
        // persistent form: four blocks per CU (40 KB of LDS each), each with its next row in registers
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) {
            int v = 0;
            if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
        }
        const int grid = std::min(rows, 4 * cus);
        hipLaunchKernelGGL(cfar_mask_rows_kernel<8>, dim3(grid), dim3(CFAR_T), smem, (hipStream_t)stream, raw, rows, R, w2, guard, mincol,
                           maxcol, a_thresh, b_thresh, diff, steep_fact, mask);
    } else {
        hipLaunchKernelGGL(cfar_mask_kernel, dim3(A, B), dim3(CFAR_T), smem, (hipStream_t)stream, raw, R, w2, guard, mincol,
                           maxcol, a_thresh, b_thresh, diff, steep_fact, mask);
    }
    MMK_LAUNCH_CHECK();
    return MMK_OK;
}

extern "C" size_t mmk_extract_peaks_workspace_bytes(int32_t B, int32_t A, int32_t R, int32_t max_pts)
{
    (void)R;
    if (B < 1 || A < 1 || max_pts < 1) return 0;
    return carve_peaks(B, A, max_pts, nullptr, 0).bytes;
}

extern "C" int mmk_extract_peaks(const float *mask, int32_t B, int32_t A, int32_t R, float res, const float *azimuths,
                                 const float *times, const float *T_ab, int32_t diff, float steep_fact,
                                 int32_t max_pts, float *out_pc, int32_t *out_count, void *workspace,
                                 size_t workspace_bytes, void *stream)
{
    (void)times;  // the reference averages the azimuth times too but drops them in pol_2_cart (radar_utils.py:99,187-195)
    MMK_REQUIRE(mask && azimuths && out_pc && out_count, "mmk_extract_peaks: NULL pointer");
    MMK_REQUIRE(B >= 1 && A >= 1 && R >= 2 && max_pts >= 1, "mmk_extract_peaks: bad shape");
    const PeakWs w = carve_peaks(B, A, max_pts, workspace, workspace_bytes);
    if (workspace == nullptr || w.bytes > workspace_bytes) {
        mmk::set_error("mmk_extract_peaks: workspace too small (%zu < %zu)", workspace_bytes, w.bytes);
        return MMK_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const size_t row_lds = (size_t)R * sizeof(float);
    MMK_REQUIRE(row_lds <= 64 * 1024 - 64, "mmk_extract_peaks: R=%d does not fit the LDS row buffer", R);
    hipLaunchKernelGGL(peaks_count_kernel, dim3(A, B), dim3(RT), row_lds, st, mask, R, res, diff, steep_fact, w.row_count);
    MMK_LAUNCH_CHECK();
    hipLaunchKernelGGL(peaks_scan_kernel, dim3(B), dim3(RT), 0, st, w.row_count, A, w.row_off, w.total);
    MMK_LAUNCH_CHECK();
    hipLaunchKernelGGL(peaks_emit_kernel, dim3(A, B), dim3(RT), row_lds, st, mask, R, res, diff, steep_fact, w.row_off, w.cap,
                       w.mval, w.mrow);
    MMK_LAUNCH_CHECK();
    hipLaunchKernelGGL(peaks_pair_kernel, dim3((max_pts + 255) / 256, B), dim3(256), 0, st, w.mval, w.mrow, w.cap,
                       w.total, azimuths, T_ab, A, max_pts, out_pc, out_count);
    MMK_LAUNCH_CHECK();
    return MMK_OK;
}

extern "C" int mmk_polar_to_cart(const float *polar, const float *azimuths, const float *range_grid,
                                 const float *angle_grid, int32_t B, int32_t A, int32_t R, int32_t W,
                                 float radar_resolution, int32_t interpolate_crossover, int32_t fix_wobble, float *cart,
                                 void *stream)
{
    MMK_REQUIRE(polar && azimuths && range_grid && angle_grid && cart, "mmk_polar_to_cart: NULL pointer");
    MMK_REQUIRE(B >= 1 && A >= 2 && R >= 2 && W >= 1, "mmk_polar_to_cart: bad shape");
    MMK_REQUIRE((size_t)A * 4 <= 64 * 1024, "mmk_polar_to_cart: too many azimuths (%d)", A);
    const float half_res = (float)((double)radar_resolution / 2.0);
    hipLaunchKernelGGL(polar_to_cart_kernel, dim3(((W + 31) / 32) * ((W + 7) / 8), B), dim3(256), (size_t)A * 4, (hipStream_t)stream,
                       polar, azimuths, range_grid, angle_grid, A, R, W, radar_resolution, half_res,
                       interpolate_crossover ? 1 : 0, fix_wobble ? 1 : 0, cart, (const float *)nullptr, (float *)nullptr);
    MMK_LAUNCH_CHECK();
    return MMK_OK;
}

extern "C" int mmk_cart_to_polar(const double *cart, const double *sin_az, const double *cos_az, const double *range_coords,
                                 int32_t B, int32_t A, int32_t R, int32_t H, int32_t W, double cart_resolution, double *polar,
                                 void *stream)
{
    MMK_REQUIRE(cart && sin_az && cos_az && range_coords && polar, "mmk_cart_to_polar: NULL pointer");
    MMK_REQUIRE(B >= 1 && B <= 65535 && A >= 1 && A <= 65535 && R >= 1 && H >= 2 && W >= 2, "mmk_cart_to_polar: bad shape");
    MMK_REQUIRE(cart_resolution > 0.0, "mmk_cart_to_polar: cart_resolution must be positive");
    hipLaunchKernelGGL(cart_to_polar_kernel, dim3((R + 255) / 256, A, B), dim3(256), 0, (hipStream_t)stream, cart, sin_az, cos_az,
                       range_coords, A, R, H, W, cart_resolution, polar);
    MMK_LAUNCH_CHECK();
    return MMK_OK;
}

extern "C" int mmk_polar_to_cart_pair(const float *polar, const float *polar2, const float *azimuths, const float *range_grid,
                                      const float *angle_grid, int32_t B, int32_t A, int32_t R, int32_t W,
                                      float radar_resolution, int32_t interpolate_crossover, int32_t fix_wobble, float *cart,
                                      float *cart2, void *stream)
{
    MMK_REQUIRE(polar && polar2 && azimuths && range_grid && angle_grid && cart && cart2, "mmk_polar_to_cart_pair: NULL pointer");
    MMK_REQUIRE(B >= 1 && A >= 2 && R >= 2 && W >= 1, "mmk_polar_to_cart_pair: bad shape");
    MMK_REQUIRE((size_t)A * 4 <= 64 * 1024, "mmk_polar_to_cart_pair: too many azimuths (%d)", A);
    const float half_res = (float)((double)radar_resolution / 2.0);
    hipLaunchKernelGGL(polar_to_cart_kernel, dim3(((W + 31) / 32) * ((W + 7) / 8), B), dim3(256), (size_t)A * 4, (hipStream_t)stream,
                       polar, azimuths, range_grid, angle_grid, A, R, W, radar_resolution, half_res,
                       interpolate_crossover ? 1 : 0, fix_wobble ? 1 : 0, cart, polar2, cart2);
    MMK_LAUNCH_CHECK();
    return MMK_OK;
}

extern "C" int mmk_sample_weights_fwd(const float *mask, const float *pc, int32_t B, int32_t N, int32_t pc_cols,
                                      int32_t H, int32_t W, int32_t cart_pixel_width, float cart_resolution,
                                      float *weights, void *stream)
{
    MMK_REQUIRE(mask && pc && weights, "mmk_sample_weights_fwd: NULL pointer");
    MMK_REQUIRE(B >= 1 && N >= 1 && pc_cols >= 2 && H >= 2 && W >= 2 && cart_pixel_width >= 2,
                "mmk_sample_weights_fwd: bad shape");
    hipLaunchKernelGGL(sample_weights_fwd_kernel, dim3((N + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, mask, pc, N,
                       pc_cols, H, W, cart_pixel_width, cart_resolution, weights);
    MMK_LAUNCH_CHECK();
    return MMK_OK;
}

extern "C" size_t mmk_sample_weights_bwd_ws_bytes(int32_t B, int32_t N)
{
    return B < 1 || N < 1 ? 0 : (size_t)B * N * 4 * 4 * sizeof(float);       // next | val | pix | res, one word per (point, tap) each
}

extern "C" int mmk_sample_weights_bwd(const float *grad_weights, const float *pc, int32_t B, int32_t N, int32_t pc_cols,
                                      int32_t H, int32_t W, int32_t cart_pixel_width, float cart_resolution,
                                      float *grad_mask, void *ws, size_t ws_bytes, void *stream)
{
    MMK_REQUIRE(grad_weights && pc && grad_mask && ws, "mmk_sample_weights_bwd: NULL pointer");
    MMK_REQUIRE(B >= 1 && N >= 1 && pc_cols >= 2 && H >= 2 && W >= 2 && cart_pixel_width >= 2,
                "mmk_sample_weights_bwd: bad shape");
    MMK_REQUIRE((size_t)N * 4 < ((size_t)1 << 30) && (size_t)H * W < ((size_t)1 << 31), "mmk_sample_weights_bwd: shape too large");
    MMK_REQUIRE(ws_bytes >= mmk_sample_weights_bwd_ws_bytes(B, N), "mmk_sample_weights_bwd: workspace too small (%zu < %zu bytes)",
                ws_bytes, mmk_sample_weights_bwd_ws_bytes(B, N));
    hipStream_t st = (hipStream_t)stream;
    const size_t ne = (size_t)B * N * 4;
    int *next = static_cast<int *>(ws);
    float *val = reinterpret_cast<float *>(next + ne);
    int *pix = reinterpret_cast<int *>(val + ne);
    float *res = reinterpret_cast<float *>(pix + ne);
    MMK_CHECK_HIP(hipMemsetAsync(grad_mask, 0, sizeof(float) * (size_t)B * H * W, st));
    const dim3 grid((4 * N + 255) / 256, B);
    hipLaunchKernelGGL(sample_weights_bwd_link_kernel, grid, dim3(256), 0, st, grad_weights, pc, N, pc_cols, H, W, cart_pixel_width,
                       cart_resolution, grad_mask, next, val, pix);
    MMK_LAUNCH_CHECK();
    hipLaunchKernelGGL(sample_weights_bwd_sum_kernel, grid, dim3(256), 0, st, N, H, W, grad_mask, next, val, pix, res);
    MMK_LAUNCH_CHECK();
    hipLaunchKernelGGL(sample_weights_bwd_store_kernel, grid, dim3(256), 0, st, N, H, W, pix, res, grad_mask);
    MMK_LAUNCH_CHECK();
    return MMK_OK;
}

extern "C" int mmk_weight_stats(const float *weights, const float *pc, int32_t B, int32_t N, int32_t pc_cols, float *partial,
                                float *out, void *stream)
{
    MMK_REQUIRE(weights && pc && partial && out, "mmk_weight_stats: NULL pointer");
    MMK_REQUIRE(B >= 1 && N >= 1 && pc_cols >= 2, "mmk_weight_stats: bad shape");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(weight_stats_partial_kernel, dim3(B), dim3(256), 0, st, weights, pc, N, pc_cols, partial);
    MMK_LAUNCH_CHECK();
    hipLaunchKernelGGL(weight_stats_final_kernel, dim3(1), dim3(64), 0, st, partial, B, out);
    MMK_LAUNCH_CHECK();
    return MMK_OK;
}

extern "C" int mmk_bev_raster(const float *pc, int32_t B, int32_t M, int32_t pc_cols, int32_t W, float cart_resolution,
                              float *bev, void *stream)
{
    MMK_REQUIRE(pc && bev, "mmk_bev_raster: NULL pointer");
    MMK_REQUIRE(B >= 1 && M >= 1 && pc_cols >= 2 && W >= 2, "mmk_bev_raster: bad shape");
    hipStream_t st = (hipStream_t)stream;
    MMK_CHECK_HIP(hipMemsetAsync(bev, 0, sizeof(float) * (size_t)B * W * W, st));
    hipLaunchKernelGGL(bev_raster_kernel, dim3((M + 255) / 256, B), dim3(256), 0, st, pc, M, pc_cols, W, cart_resolution,
                       bev);
    MMK_LAUNCH_CHECK();
    hipLaunchKernelGGL(bev_centre_kernel, dim3((B + 255) / 256), dim3(256), 0, st, bev, B, W);
    MMK_LAUNCH_CHECK();
    return MMK_OK;
}
