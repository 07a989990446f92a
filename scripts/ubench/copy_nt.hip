// Micro-benchmark: what a device copy of 210 MB + 210 MB reaches with plain, non-temporal and "streaming" (sc1 / nt) loads and
// stores on gfx950 -- are the write-once output tensors of the thin layers better written around the L2?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/copy_nt scripts/ubench/copy_nt.hip && /tmp/copy_nt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void copy_kernel(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        u32x4 v;
        if (MODE & 2) v = __builtin_nontemporal_load(in + i);
        else v = in[i];
        if (MODE & 1) __builtin_nontemporal_store(v, out + i);
        else out[i] = v;
    }
}

template <int MODE>
float run(const u32x4 *in, u32x4 *out, size_t n, int grid)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(copy_kernel<MODE>, dim3(grid), dim3(256), 0, 0, in, out, n);
    hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(copy_kernel<MODE>, dim3(grid), dim3(256), 0, 0, in, out, n);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps * 1e3f;
}

int main()
{
    const size_t bytes = (size_t)32 * 640 * 640 * 8 * 2;      // one 8-channel bf16 tensor at B = 32, 640 x 640
    u32x4 *in, *out;
    hipMalloc(&in, bytes); hipMalloc(&out, bytes);
    hipMemset(in, 1, bytes); hipMemset(out, 0, bytes);
    const size_t n = bytes / 16;
    for (int grid : {256 * 4, 256 * 8, 256 * 16, 256 * 32}) {
        const float a = run<0>(in, out, n, grid), b = run<1>(in, out, n, grid), c = run<2>(in, out, n, grid), d = run<3>(in, out, n, grid);
        printf("grid %5d  plain %.1f us %.2f TB/s | nt store %.1f us %.2f | nt load %.1f us %.2f | both %.1f us %.2f\n", grid, a, 2 * bytes / a * 1e-6,
               b, 2 * bytes / b * 1e-6, c, 2 * bytes / c * 1e-6, d, 2 * bytes / d * 1e-6);
    }
    return 0;
}
