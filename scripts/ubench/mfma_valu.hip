// Micro-benchmark: how v_mfma_f32_32x32x16_bf16 (result in VGPRs) shares a SIMD with the vector instructions that consume
// its result (the chunk-minimum reduction of nn_mfma_kernel).  NV = v_min3 per MFMA (0, 4, 8), EXTRA = the 5 flag / bound
// instructions, waves per SIMD = blocks per CU (256-thread blocks).  Prints shader cycles per MFMA and SIMD (s_memtime).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu mfma_valu.hip && ./mfma_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ f32x16 mfma0(bf16x8 a, bf16x8 b)
{
    f32x16 d;
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b));
    return d;
}

template <int NV, int EXTRA, int USE_LDS>
__global__ __launch_bounds__(256) void k(const uint4 *__restrict__ in, float *__restrict__ out, int iters, long long *__restrict__ cyc)
{
    __shared__ uint4 frag[32][64];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 32 * 64; i += 256) frag[i / 64][i % 64] = in[i];
    __syncthreads();
    bf16x8 b[4];
    for (int g = 0; g < 4; ++g) b[g] = __builtin_bit_cast(bf16x8, in[2048 + g * 64 + lane]);
    float thr[4] = {1e30f, 1e30f, 1e30f, 1e30f}, brun[4] = {1e30f, 1e30f, 1e30f, 1e30f};
    unsigned w[4] = {0, 0, 0, 0};
    bf16x8 a = __builtin_bit_cast(bf16x8, frag[0][lane]);
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll 2
        for (int c = 0; c < 32; ++c) {
            if (USE_LDS) a = __builtin_bit_cast(bf16x8, frag[c][lane]);
            f32x16 acc[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = mfma0(a, b[g]);
            asm volatile("s_nop 3" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]));
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (g == 3) asm volatile("" : "+v"(acc[3]), "+v"(thr[0]), "+v"(thr[1]), "+v"(thr[2]));
                auto m3 = [](float x, float y, float z) { return __builtin_fminf(__builtin_fminf(x, y), z); };
                float t = thr[g];
                if (NV >= 8) {
                    const float m0 = m3(thr[g], acc[g][0], acc[g][1]), m1 = m3(acc[g][2], acc[g][3], acc[g][4]), m2 = m3(acc[g][5], acc[g][6], acc[g][7]);
                    const float m4 = m3(acc[g][8], acc[g][9], acc[g][10]), m5 = m3(acc[g][11], acc[g][12], acc[g][13]);
                    t = m3(m3(m0, m1, m2), m3(m4, m5, acc[g][14]), acc[g][15]);
                } else if (NV >= 4) {
                    const float m0 = m3(thr[g], acc[g][0], acc[g][1]), m1 = m3(acc[g][2], acc[g][3], acc[g][4]);
                    t = m3(m0, m1, m3(acc[g][5], acc[g][6], acc[g][7]));
                    t = m3(t, acc[g][8], acc[g][15]);
                } else {
                    t = __builtin_fminf(t, acc[g][0]);
                }
                if (EXTRA == 1) {
                    w[g] = (w[g] << 1) | ((t < thr[g]) ? 1u : 0u);
                    brun[g] = __builtin_fminf(brun[g], t);
                    thr[g] = __builtin_fmaf(brun[g], 1.00001f, 1e-3f);
                } else if (EXTRA == 2) {          // sign bit of t - thr shifted in: no compare, no lane mask
                    w[g] = __builtin_amdgcn_alignbit(w[g], __float_as_uint(t - thr[g]), 31);
                    brun[g] = __builtin_fminf(brun[g], t);
                    thr[g] = __builtin_fmaf(brun[g], 1.00001f, 1e-3f);
                } else if (EXTRA == 3) {          // ... and the bound refreshed every other chunk
                    w[g] = __builtin_amdgcn_alignbit(w[g], __float_as_uint(t - thr[g]), 31);
                    brun[g] = __builtin_fminf(brun[g], t);
                    if (c & 1) thr[g] = __builtin_fmaf(brun[g], 1.00001f, 1e-3f);
                } else if (EXTRA == 4) {          // flags only
                    w[g] = __builtin_amdgcn_alignbit(w[g], __float_as_uint(t - thr[g]), 31);
                    thr[g] = t + 1.0f;
                } else if (EXTRA == 5) {          // bound only
                    brun[g] = __builtin_fminf(brun[g], t);
                    thr[g] = __builtin_fmaf(brun[g], 1.00001f, 1e-3f);
                } else {
                    thr[g] = t;
                }
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int g = 0; g < 4; ++g) s += thr[g] + brun[g] + (float)w[g];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NV, int EXTRA, int USE_LDS>
void run(const char *name, const uint4 *in, float *out, long long *cyc, int per_cu)
{
    const int iters = 200, blocks = 256 * per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NV, EXTRA, USE_LDS>), dim3(blocks), dim3(256), 0, 0, in, out, 10, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NV, EXTRA, USE_LDS>), dim3(blocks), dim3(256), 0, 0, in, out, iters, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= blocks;
    const double mfma_per_wave = (double)iters * 32 * 4;             // per wave
    const double per_simd = mfma_per_wave * per_cu;                  // waves per SIMD = per_cu (4 waves per block, 4 SIMDs)
    printf("%-28s waves/SIMD %d: %.1f us, %.1f cycles (s_memtime) per MFMA of a wave, %.1f per MFMA and SIMD, %.2f ns per MFMA and SIMD\n", name, per_cu,
           ms * 1e3, avg / mfma_per_wave, avg / per_simd, ms * 1e6 / per_simd);
}

int main()
{
    uint4 *in; float *out; long long *cyc;
    hipMalloc(&in, (2048 + 256) * 16); hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&cyc, 4096 * 8);
    std::vector<unsigned> h((2048 + 256) * 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3f803f80u ^ (unsigned)(i * 2654435761u & 0x007f007fu);     // bf16 values near 1
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int per_cu = 3; per_cu <= 4; ++per_cu) {
        run<0, 0, 0>("mfma only", in, out, cyc, per_cu);
        run<8, 0, 0>("mfma + 8 min3", in, out, cyc, per_cu);
        run<8, 1, 0>("+ cmp/cndmask/or, min, fma", in, out, cyc, per_cu);
        run<8, 2, 0>("+ sub/alignbit, min, fma", in, out, cyc, per_cu);
        run<8, 3, 0>("+ sub/alignbit, min, fma/2", in, out, cyc, per_cu);
        run<8, 4, 0>("+ sub/alignbit, add", in, out, cyc, per_cu);
        run<8, 5, 0>("+ min, fma", in, out, cyc, per_cu);
        run<8, 2, 1>("sub/alignbit,min,fma + LDS", in, out, cyc, per_cu);
    }
    return 0;
}
