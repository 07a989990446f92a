// Are agent-scope RMW atomics (no sc1 bit) on hipMalloc'ed memory coherent across the 8 XCD L2s?  Every block hammers the same
// small array with non-returning 64-bit atomicAdd / atomicMin; lost updates show up as a wrong total / a key that is not the minimum.
// hipcc --offload-arch=gfx950 -O3 -o xcd_atomics xcd_atomics.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void hammer(unsigned long long *cnt, unsigned long long *keys, int n, int rounds, int sys)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    for (int r = 0; r < rounds; ++r) {
        const int k = (g * 7 + r * 131) % n;
        const unsigned long long key = ((unsigned long long)(unsigned)((g * 2654435761u + r * 40503u) | 1u) << 32) | (unsigned)g;
        if (sys) {
            __hip_atomic_fetch_add(&cnt[k], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_fetch_min(&keys[k], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        } else {
            atomicAdd(&cnt[k], 1ull);
            atomicMin(&keys[k], key);
        }
    }
}
int main()
{
    const int n = 4096, blocks = 2048, threads = 256, rounds = 64;
    unsigned long long *cnt, *keys;
    CK(hipMalloc(&cnt, n * 8)); CK(hipMalloc(&keys, n * 8));
    std::vector<unsigned long long> hc(n), hk(n), want(n);
    for (int sys = 0; sys < 2; ++sys)
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipMemset(cnt, 0, n * 8)); CK(hipMemset(keys, 0xFF, n * 8));
            hipLaunchKernelGGL(hammer, dim3(blocks), dim3(threads), 0, 0, cnt, keys, n, rounds, sys);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(hc.data(), cnt, n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hk.data(), keys, n * 8, hipMemcpyDeviceToHost));
            for (int k = 0; k < n; ++k) want[k] = ~0ull;
            for (int g = 0; g < blocks * threads; ++g)
                for (int r = 0; r < rounds; ++r) {
                    const int k = (int)(((long long)g * 7 + (long long)r * 131) % n);
                    const unsigned long long key = ((unsigned long long)(unsigned)(((unsigned)g * 2654435761u + (unsigned)r * 40503u) | 1u) << 32) | (unsigned)g;
                    if (key < want[k]) want[k] = key;
                }
            unsigned long long tot = 0; int badk = 0;
            for (int k = 0; k < n; ++k) { tot += hc[k]; badk += hk[k] != want[k]; }
            printf("%s scope rep %d: adds %llu of %llu, wrong minima %d of %d\n", sys ? "system" : "agent ", rep, tot,
                   (unsigned long long)blocks * threads * rounds, badk, n);
        }
    return 0;
}
