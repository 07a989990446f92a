// Does a small per-lane scratch array survive a store -> (work) -> load-at-run-time-offset round trip in a persistent kernel?
// Mimics what the first cut of nn_mfma_kernel did with its point coordinates (HISTORY.md 9.1): per unit, 8 floats per lane are
// written through a pointer (so that they live in scratch), a few thousand cycles of LDS + MFMA work follow, then 4 of them are
// read back at an offset that depends on the lane's half.  Counts read-backs that differ from the value the lane computed.
//   hipcc --offload-arch=gfx950 -O3 -o scratch_roundtrip scratch_roundtrip.hip && ./scratch_roundtrip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __noinline__ void fill(float *p, unsigned seed)      // (noinline: the array's address escapes -> scratch)
{
#pragma unroll
    for (int k = 0; k < 8; ++k) p[k] = __uint_as_float(0x3f800000u | ((seed * 2654435761u + k * 40503u) & 0x7fffffu));
}

__global__ __launch_bounds__(256, 4) void roundtrip(int units, int work, unsigned long long *bad, unsigned long long *total, float *sink)
{
    __shared__ __attribute__((aligned(16))) float lds[10240];       // 40 KB, as the NN kernel
    const int lane = threadIdx.x & 63, half = lane >> 5;
    unsigned long long nbad = 0, ntot = 0;
    float acc_sink = 0.f;
    for (int u = blockIdx.x; u < units; u += gridDim.x) {
        float p[8];
        const unsigned seed = (unsigned)u * 1315423911u + threadIdx.x;
        fill(p, seed);
        // ---- work: LDS traffic + MFMAs between the store and the load
        __syncthreads();
        for (int i = threadIdx.x; i < 10240; i += 256) lds[i] = (float)(i ^ u);
        __syncthreads();
        f32x16 acc = {};
        bf16x8 a, b;
#pragma unroll
        for (int k = 0; k < 8; ++k) { a[k] = (__bf16)(float)((lane + k) & 7); b[k] = (__bf16)1.0f; }
        for (int w = 0; w < work; ++w) {
            const float4 v = *reinterpret_cast<const float4 *>(&lds[((w * 64 + lane) * 4) % 10240]);
            a[0] = (__bf16)v.x;
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        }
        acc_sink += acc[0] + acc[15];
        // ---- read back at a run-time offset
        const float *q = p + (half ? 4 : 0);
        float exp4[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            exp4[k] = __uint_as_float(0x3f800000u | ((seed * 2654435761u + (unsigned)(k + (half ? 4 : 0)) * 40503u) & 0x7fffffu));
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            nbad += (q[k] != exp4[k]);
            ntot += 1;
        }
    }
    atomicAdd(bad, nbad);
    atomicAdd(total, ntot);
    if (acc_sink == 12345.f) sink[0] = acc_sink;
}

int main()
{
    unsigned long long *bad, *total;
    float *sink;
    CK(hipMalloc(&bad, 8)); CK(hipMalloc(&total, 8)); CK(hipMalloc(&sink, 4));
    hipFuncAttributes at;
    CK(hipFuncGetAttributes(&at, (const void *)roundtrip));
    printf("kernel: %d VGPRs, %zu bytes of scratch per lane\n", at.numRegs, (size_t)at.localSizeBytes);
    for (int work = 16; work <= 1024; work *= 4)
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(bad, 0, 8)); CK(hipMemset(total, 0, 8));
            hipLaunchKernelGGL(roundtrip, dim3(1024), dim3(256), 0, 0, 200000, work, bad, total, sink);
            CK(hipDeviceSynchronize());
            unsigned long long hb, ht;
            CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&ht, total, 8, hipMemcpyDeviceToHost));
            printf("work %4d rep %d: %llu wrong read-backs of %llu\n", work, rep, hb, ht);
        }
    return 0;
}
