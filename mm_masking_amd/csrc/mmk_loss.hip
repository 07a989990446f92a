// Loss terms of the training step as single launches (round 4): the pose terms and the BCE mask terms of
// eval_training_loss (mm_masking/train_icp_weights.py:179-253) and their gradients.  Through PyTorch these were ~45 launches
// of 2-36 us per step (slices, subtractions, norms, means, BCELoss forward / mean / backward, scalings); each is one kernel
// (plus an ordered final sum) here.  Deterministic: block partials in a fixed order, no float atomics.
//
//   pose terms (:192-200, gt_eye):  xi = T_pred - I;  rot = mean_b |xi[b,1,0]|  (torch.norm over a 1-vector),
//                                   trans = mean_b sqrt(xi[b,0,3]^2 + xi[b,1,3]^2)
//   mask terms (:204-226):          torch.nn.BCELoss()(mask, target) = mean(-(t max(log x, -100) + (1 - t) max(log(1 - x), -100)))
//                                   gradient (x - t) / max((1 - x) x, 1e-12) / n, as torch's binary_cross_entropy_backward
#include <math.h>

#include <algorithm>

#include "mmk_common.h"

namespace {

constexpr int BCE_BLOCKS = 2048, BCE_THREADS = 256;

__global__ __launch_bounds__(64) void pose_loss_fwd_kernel(const float *__restrict__ T, int B, float *__restrict__ out)
{
    // one wave; lane l sums pairs l, l + 64, ... in index order, then a fixed shuffle tree
    double rot = 0.0, trans = 0.0;
    for (int b = threadIdx.x; b < B; b += 64) {
        const float *t = T + (size_t)b * 16;
        const float th = t[4], x = t[3], y = t[7];               // xi[1,0]; xi[0,3], xi[1,3] (the identity has zeros there)
        rot += (double)fabsf(th);
        trans += (double)sqrtf(x * x + y * y);
    }
    rot = wave_sum(rot);
    trans = wave_sum(trans);
    if (threadIdx.x == 0) {
        out[0] = (float)(rot / B);
        out[1] = (float)(trans / B);
    }
}

__global__ void pose_loss_bwd_kernel(const float *__restrict__ T, int B, const float *__restrict__ g_rot, const float *__restrict__ g_trans,
                                     float *__restrict__ gT)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float gr = g_rot ? g_rot[0] : 0.f, gt = g_trans ? g_trans[0] : 0.f;
    const float *t = T + (size_t)b * 16;
    float *g = gT + (size_t)b * 16;
#pragma unroll
    for (int i = 0; i < 16; ++i) g[i] = 0.f;
    const float th = t[4], x = t[3], y = t[7];
    // d|th| = sign(th) (0 at 0, as torch.norm's backward masks the zero norm); d sqrt(x^2 + y^2) = (x, y) / norm (0 at 0)
    const float inv_b = 1.0f / (float)B;
    g[4] = (th > 0.f ? 1.f : (th < 0.f ? -1.f : 0.f)) * gr * inv_b;
    const float n = sqrtf(x * x + y * y);
    if (n > 0.f) {
        g[3] = x / n * gt * inv_b;
        g[7] = y / n * gt * inv_b;
    }
}

__device__ __forceinline__ float bce_term(float x, float t)
{
    const float lx = fmaxf(logf(x), -100.f), l1 = fmaxf(logf(1.f - x), -100.f);
    return -(t * lx + (1.f - t) * l1);
}

__global__ __launch_bounds__(BCE_THREADS) void bce_partial_kernel(const float *__restrict__ x, const float *__restrict__ t, size_t n,
                                                                  double *__restrict__ part)
{
    __shared__ double red[BCE_THREADS / 64];
    double s = 0.0;
    const size_t n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * BCE_THREADS + threadIdx.x; i < n4; i += (size_t)gridDim.x * BCE_THREADS) {
        const float4 xv = reinterpret_cast<const float4 *>(x)[i], tv = reinterpret_cast<const float4 *>(t)[i];
        s += (double)((bce_term(xv.x, tv.x) + bce_term(xv.y, tv.y)) + (bce_term(xv.z, tv.z) + bce_term(xv.w, tv.w)));
    }
    if (blockIdx.x == 0)
        for (size_t i = n4 * 4 + threadIdx.x; i < n; i += BCE_THREADS) s += (double)bce_term(x[i], t[i]);
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void bce_final_kernel(const double *__restrict__ part, int nblk, double inv_n, float *__restrict__ out)
{
    __shared__ double red[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) s += part[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (float)(((red[0] + red[1]) + (red[2] + red[3])) * inv_n);
}

__global__ __launch_bounds__(BCE_THREADS) void bce_bwd_kernel(const float *__restrict__ x, const float *__restrict__ t, size_t n,
                                                              const float *__restrict__ gout, float inv_n, float *__restrict__ g)
{
    const float s = gout[0] * inv_n;
    auto grad = [s](float xv, float tv) { return (xv - tv) / fmaxf((1.f - xv) * xv, 1e-12f) * s; };
    const size_t n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * BCE_THREADS + threadIdx.x; i < n4; i += (size_t)gridDim.x * BCE_THREADS) {
        const float4 xv = reinterpret_cast<const float4 *>(x)[i], tv = reinterpret_cast<const float4 *>(t)[i];
        reinterpret_cast<float4 *>(g)[i] = make_float4(grad(xv.x, tv.x), grad(xv.y, tv.y), grad(xv.z, tv.z), grad(xv.w, tv.w));
    }
    if (blockIdx.x == 0)
        for (size_t i = n4 * 4 + threadIdx.x; i < n; i += BCE_THREADS) g[i] = grad(x[i], t[i]);
}

}  // namespace

extern "C" int mmk_pose_loss_fwd(const float *T_pred, int32_t B, float *out2, void *stream)
{
    MMK_REQUIRE(T_pred && out2 && B >= 1, "mmk_pose_loss_fwd: bad argument");
    hipLaunchKernelGGL(pose_loss_fwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, T_pred, B, out2);
    MMK_LAUNCH_CHECK();
    return MMK_OK;
}

extern "C" int mmk_pose_loss_bwd(const float *T_pred, int32_t B, const float *g_rot, const float *g_trans, float *grad_T, void *stream)
{
    MMK_REQUIRE(T_pred && grad_T && B >= 1, "mmk_pose_loss_bwd: bad argument");
    hipLaunchKernelGGL(pose_loss_bwd_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, T_pred, B, g_rot, g_trans, grad_T);
    MMK_LAUNCH_CHECK();
    return MMK_OK;
}

extern "C" size_t mmk_bce_ws_bytes(void) { return (size_t)BCE_BLOCKS * sizeof(double); }

extern "C" int mmk_bce_mean_fwd(const float *x, const float *target, int64_t n, void *ws, size_t ws_bytes, float *out, void *stream)
{
    MMK_REQUIRE(x && target && out && ws && n >= 1, "mmk_bce_mean_fwd: bad argument");
    MMK_REQUIRE(ws_bytes >= mmk_bce_ws_bytes(), "mmk_bce_mean_fwd: workspace too small");
    MMK_REQUIRE((((uintptr_t)x | (uintptr_t)target) & 15) == 0, "mmk_bce_mean_fwd: inputs must be 16-byte aligned");
    const int nblk = (int)std::min<size_t>(BCE_BLOCKS, ((size_t)n / 4 + BCE_THREADS - 1) / BCE_THREADS + 1);
    hipLaunchKernelGGL(bce_partial_kernel, dim3(nblk), dim3(BCE_THREADS), 0, (hipStream_t)stream, x, target, (size_t)n, (double *)ws);
    MMK_LAUNCH_CHECK();
    hipLaunchKernelGGL(bce_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const double *)ws, nblk, 1.0 / (double)n, out);
    MMK_LAUNCH_CHECK();
    return MMK_OK;
}

extern "C" int mmk_bce_mean_bwd(const float *x, const float *target, int64_t n, const float *grad_out, float *grad_x, void *stream)
{
    MMK_REQUIRE(x && target && grad_out && grad_x && n >= 1, "mmk_bce_mean_bwd: bad argument");
    MMK_REQUIRE((((uintptr_t)x | (uintptr_t)target | (uintptr_t)grad_x) & 15) == 0, "mmk_bce_mean_bwd: buffers must be 16-byte aligned");
    const int nblk = (int)std::min<size_t>(4096, ((size_t)n / 4 + BCE_THREADS - 1) / BCE_THREADS + 1);
    hipLaunchKernelGGL(bce_bwd_kernel, dim3(nblk), dim3(BCE_THREADS), 0, (hipStream_t)stream, x, target, (size_t)n, grad_out,
                       (float)(1.0 / (double)n), grad_x);
    MMK_LAUNCH_CHECK();
    return MMK_OK;
}
