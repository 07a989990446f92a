// Loader primitives of the real-data path (mm_masking/icp_weight_dataset.py:323-362 runs its per-item work in four
// DataLoader worker processes, train_icp_weights.py:454-455).  Here the byte-moving half of an item is ONE C call that
// Python worker THREADS make through ctypes (which drops the GIL): mmk_host_read_rows copies the rows of a decoded scan /
// a prepared cloud from the page cache straight into the batch's pinned host buffer, cutting the columns it needs and
// applying the augmentation's azimuth roll on the way -- one copy per byte, no interpreter in the loop.  The device half
// is mmk_u8_to_float (bytes -> fp32 through a 256-entry table the host computed with the reference's own division) and
// the batched polar -> Cartesian launch (mmk_polar_to_cart_pair).
#include <algorithm>
#include <errno.h>
#include <fcntl.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "mmk_common.h"

#include "mmk_loader_host.inc"

namespace {
__global__ __launch_bounds__(256) void u8_to_float_kernel(const uint8_t *__restrict__ in, const float *__restrict__ lut, size_t n16,
                                                          size_t n, float *__restrict__ out)
{
    __shared__ float tab[256];
    tab[threadIdx.x] = lut[threadIdx.x];
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = reinterpret_cast<const uint4 *>(in)[i];
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
        float4 *o = reinterpret_cast<float4 *>(out) + i * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            o[k] = make_float4(tab[w[k] & 255u], tab[(w[k] >> 8) & 255u], tab[(w[k] >> 16) & 255u], tab[w[k] >> 24]);
    }
    if (blockIdx.x == 0)                                                 // tail (n not a multiple of 16)
        for (size_t i = n16 * 16 + threadIdx.x; i < n; i += blockDim.x) out[i] = tab[in[i]];
}
}  // namespace

extern "C" int mmk_u8_to_float(const void *in, const float *lut256, int64_t n, float *out, void *stream)
{
    MMK_REQUIRE(in && lut256 && out && n >= 1, "mmk_u8_to_float: bad argument");
    MMK_REQUIRE(((uintptr_t)in & 15) == 0 && ((uintptr_t)out & 15) == 0, "mmk_u8_to_float: buffers must be 16-byte aligned");
    const size_t n16 = (size_t)n / 16;
    const unsigned blocks = (unsigned)std::max<size_t>(1, std::min<size_t>((n16 + 255) / 256, 256 * 16));
    hipLaunchKernelGGL(u8_to_float_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint8_t *)in, lut256, n16, (size_t)n,
                       out);
    MMK_LAUNCH_CHECK();
    return MMK_OK;
}
