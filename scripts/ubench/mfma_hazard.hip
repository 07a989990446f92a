// How many wait states does a vector read of a v_mfma_f32_32x32x16_bf16 result (VGPR destination, inline asm: hipcc pads
// nothing) need?  Two MFMAs back to back into acc0 / acc1, alternating operands so that a stale read returns the previous
// iteration's value; acc0 is read behind "s_nop N", acc1 behind 12 further vector instructions.  Counts wrong reads for
// N = 0..15 and 1..4 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 -o mfma_hazard mfma_hazard.hip
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

template <int NOPS>
__global__ __launch_bounds__(256) void k(int iters, unsigned *__restrict__ bad0, unsigned *__restrict__ bad1, float *__restrict__ dbg)
{
    // A = all ones; B = all (1 or 2): every element of D = 16 or 32
    const __bf16 one = (__bf16)1.0f, two = (__bf16)2.0f;
    const bf16x8 a = {one, one, one, one, one, one, one, one};
    const bf16x8 b1 = {one, one, one, one, one, one, one, one}, b2 = {two, two, two, two, two, two, two, two};
    unsigned e0 = 0, e1 = 0;
    float sink = 0.f;
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            const float want = par ? 32.f : 16.f;
            asm volatile("s_nop 7");                       // (operands written long ago; nothing of the previous round in flight)
            f32x16 acc0 = par ? mfma0(a, b2) : mfma0(a, b1);
            f32x16 acc1 = par ? mfma0(a, b2) : mfma0(a, b1);
            if (NOPS == 0) asm volatile("s_nop 0" : "+v"(acc0));
            if (NOPS == 1) asm volatile("s_nop 1" : "+v"(acc0));
            if (NOPS == 3) asm volatile("s_nop 3" : "+v"(acc0));
            if (NOPS == 5) asm volatile("s_nop 5" : "+v"(acc0));
            if (NOPS == 7) asm volatile("s_nop 7" : "+v"(acc0));
            if (NOPS == 9) asm volatile("s_nop 9" : "+v"(acc0));
            if (NOPS == 11) asm volatile("s_nop 11" : "+v"(acc0));
            if (NOPS == 15) asm volatile("s_nop 15" : "+v"(acc0));
            if (NOPS == 23) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" : "+v"(acc0));
            const float r0 = acc0[0], r15 = acc0[15], r7 = acc0[7];
            e0 += (r0 != want) + (r15 != want) + (r7 != want);
            float t = r0;
#pragma unroll
            for (int q = 0; q < 12; ++q) { t = __builtin_fmaf(t, 1.0000001f, 0.5f); asm volatile("" : "+v"(t)); }
            sink += t;
            asm volatile("" : "+v"(acc1), "+v"(t));
            e1 += (acc1[0] != want) + (acc1[15] != want) + (acc1[7] != want);
        }
    }
    if (sink == 12345.f) e0 += 1;
    atomicAdd(bad0, e0);
    atomicAdd(bad1, e1);
}

template <int NOPS>
void run(unsigned *d0, unsigned *d1, int per_cu)
{
    hipMemset(d0, 0, 4); hipMemset(d1, 0, 4);
    float *dbg; hipMalloc(&dbg, 256);
    hipLaunchKernelGGL((k<NOPS>), dim3(256 * per_cu), dim3(256), 0, 0, 20000, d0, d1, dbg);
    hipDeviceSynchronize();
    unsigned h0 = 0, h1 = 0;
    hipMemcpy(&h0, d0, 4, hipMemcpyDeviceToHost); hipMemcpy(&h1, d1, 4, hipMemcpyDeviceToHost);
    float hd[32]; hipMemcpy(hd, dbg, 128, hipMemcpyDeviceToHost);
    printf("s_nop %2d, %d waves/SIMD: wrong reads of the first tile %10u, of the second %10u (of %.3g)\n", NOPS, per_cu, h0, h1,
           3.0 * 20000 * 64 * 4 * 256 * per_cu);
}

int main()
{
    unsigned *d0, *d1;
    hipMalloc(&d0, 4); hipMalloc(&d1, 4);
    for (int per_cu = 1; per_cu <= 4; per_cu += 1) {
        run<0>(d0, d1, per_cu); run<1>(d0, d1, per_cu); run<3>(d0, d1, per_cu); run<5>(d0, d1, per_cu); run<7>(d0, d1, per_cu);
        run<9>(d0, d1, per_cu); run<11>(d0, d1, per_cu); run<15>(d0, d1, per_cu); run<23>(d0, d1, per_cu);
    }
    return 0;
}
