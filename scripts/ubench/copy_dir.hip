// Micro-benchmark: does a consumer that walks its input in the OPPOSITE direction to the producer's walk find the producer's
// tail in the memory-side cache?  K1 copies X -> Y front to back; K2 copies Y -> Z front to back or back to front.  Sizes: one
// 8- / 16- / 32-channel bf16 tensor at B = 32, 640 x 640 (210 / 419 / 839 MB) and 52 / 105 MB.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/copy_dir scripts/ubench/copy_dir.hip && scripts/ubench/copy_dir
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void copy_kernel(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, size_t n, int rev)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const size_t j = rev ? n - 1 - i : i;
        out[j] = in[j];
    }
}

int main()
{
    const int grid = 256 * 8;
    for (size_t mb : {52, 105, 210, 419, 839}) {
        const size_t bytes = mb * 1000000 / 16 * 16;
        u32x4 *x, *y, *z;
        (void)hipMalloc(&x, bytes); (void)hipMalloc(&y, bytes); (void)hipMalloc(&z, bytes);
        (void)hipMemset(x, 1, bytes);
        const size_t n = bytes / 16;
        hipEvent_t e0, e1, e2;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventCreate(&e2);
        for (int rev = 0; rev < 2; ++rev) {
            float t1 = 0.f, t2 = 0.f;
            const int reps = 10;
            for (int r = 0; r < reps + 2; ++r) {
                (void)hipEventRecord(e0);
                hipLaunchKernelGGL(copy_kernel, dim3(grid), dim3(256), 0, 0, x, y, n, 0);
                (void)hipEventRecord(e1);
                hipLaunchKernelGGL(copy_kernel, dim3(grid), dim3(256), 0, 0, y, z, n, rev);
                (void)hipEventRecord(e2);
                (void)hipEventSynchronize(e2);
                float a, b;
                (void)hipEventElapsedTime(&a, e0, e1); (void)hipEventElapsedTime(&b, e1, e2);
                if (r >= 2) { t1 += a; t2 += b; }
            }
            printf("%4zu MB  consumer %s: producer %.1f us, consumer %.1f us (%.2f TB/s)\n", mb, rev ? "back to front" : "front to back",
                   t1 / reps * 1e3, t2 / reps * 1e3, 2.0 * bytes / (t2 / reps * 1e-3) * 1e-12);
        }
        (void)hipFree(x); (void)hipFree(y); (void)hipFree(z);
    }
    return 0;
}
