// Micro-benchmark (development tool): how fast can 256-thread blocks read a (B,H,W,8) bf16 tensor in 10 x 34-pixel halo tiles
// (16 bytes per pixel, the access pattern of the thin-layer kernels), with DEPTH tiles in flight per block, against a plain
// linear sweep of the same bytes?   hipcc --offload-arch=gfx950 -O3 -o tile_read tile_read.hip ; ./tile_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int TH = 8, TW = 32, HT = 10, WT = 34, NIN = HT * WT, THREADS = 256, RIN = (NIN + THREADS - 1) / THREADS;

template <int DEPTH, int NT>       // NT tensors read per tile
__global__ __launch_bounds__(THREADS) void tile_read(const u32x4 *__restrict__ x, const u32x4 *__restrict__ x2, int B, int H, int W, u32x4 *out)
{
    const int tiles_x = W / TW, tiles_y = H / TH, tpi = tiles_x * tiles_y, total = tpi * B;
    const int per_xcd = (total + 7) / 8, xcd = blockIdx.x & 7;
    int t = xcd * per_xcd + (blockIdx.x >> 3);
    const int step = gridDim.x >> 3, t_end = min(total, (xcd + 1) * per_xcd);
    u32x4 r[DEPTH][NT][RIN];
    u32x4 acc = {0, 0, 0, 0};
    auto load = [&](u32x4 (&rr)[NT][RIN], int tt) {
        tt = tt < t_end ? tt : t_end - 1;
        const int b = tt / tpi, tr = tt % tpi, ty0 = (tr / tiles_x) * TH, tx0 = (tr % tiles_x) * TW;
#pragma unroll
        for (int i = 0; i < RIN; ++i) {
            int gi = threadIdx.x + i * THREADS;
            gi = gi < NIN ? gi : NIN - 1;
            int yy = ty0 + gi / WT - 1, xx = tx0 + gi % WT - 1;
            yy = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);
            xx = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
            const size_t p = ((size_t)b * H + yy) * W + xx;
            rr[0][i] = x[p];
            if (NT > 1) rr[NT - 1][i] = x2[p];
        }
    };
    if (t >= t_end) return;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) load(r[d], t + d * step);
    for (; t < t_end; t += DEPTH * step) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            if (t + d * step < t_end) {
#pragma unroll
                for (int k = 0; k < NT; ++k)
#pragma unroll
                    for (int i = 0; i < RIN; ++i) acc ^= r[d][k][i];
                load(r[d], t + (d + DEPTH) * step);
            }
        }
    }
    if (acc[0] == 0x12345678u) out[threadIdx.x] = acc;
}

__global__ void linear_read(const u32x4 *__restrict__ x, size_t n, u32x4 *out)
{
    u32x4 acc = {0, 0, 0, 0};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= x[i];
    if (acc[0] == 0x12345678u) out[threadIdx.x] = acc;
}

template <class F>
float timeit(F f)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) f();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) f();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 20 * 1e3f;
}

int main()
{
    const int B = 32, H = 640, W = 640;
    const size_t n = (size_t)B * H * W;             // 16-byte pixels
    u32x4 *x, *x2, *out;
    hipMalloc(&x, n * 16); hipMalloc(&x2, n * 16); hipMalloc(&out, 4096);
    hipMemset(x, 1, n * 16); hipMemset(x2, 2, n * 16);
    const double mb = n * 16 / 1e6;
    for (int grid : {1024, 1536, 2048, 4096}) {
        float a = timeit([&] { hipLaunchKernelGGL((tile_read<1, 1>), dim3(grid), dim3(THREADS), 0, 0, x, x2, B, H, W, out); });
        float b = timeit([&] { hipLaunchKernelGGL((tile_read<2, 1>), dim3(grid), dim3(THREADS), 0, 0, x, x2, B, H, W, out); });
        float c = timeit([&] { hipLaunchKernelGGL((tile_read<4, 1>), dim3(grid), dim3(THREADS), 0, 0, x, x2, B, H, W, out); });
        float d = timeit([&] { hipLaunchKernelGGL((tile_read<1, 2>), dim3(grid), dim3(THREADS), 0, 0, x, x2, B, H, W, out); });
        float e = timeit([&] { hipLaunchKernelGGL((tile_read<2, 2>), dim3(grid), dim3(THREADS), 0, 0, x, x2, B, H, W, out); });
        printf("grid %4d: 1 tensor depth 1/2/4: %.1f / %.1f / %.1f us (%.2f / %.2f / %.2f TB/s of %.0f MB);  2 tensors depth 1/2: %.1f / %.1f us (%.2f / %.2f TB/s)\n",
               grid, a, b, c, mb / a, mb / b, mb / c, mb, d, e, 2 * mb / d, 2 * mb / e);
    }
    for (int grid : {2048, 8192}) {
        float l = timeit([&] { hipLaunchKernelGGL(linear_read, dim3(grid), dim3(256), 0, 0, x, n, out); });
        printf("linear sweep, grid %d: %.1f us (%.2f TB/s)\n", grid, l, mb / l);
    }
    return 0;
}
