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

extern "C" int mmk_host_read_rows(const char *path, int64_t header_bytes, int32_t rows, int32_t row_bytes, int32_t col0,
                                  int32_t ncols, int32_t roll, void *dst)
{
    MMK_REQUIRE(path && dst, "mmk_host_read_rows: NULL pointer");
    MMK_REQUIRE(header_bytes >= 0 && rows >= 1 && row_bytes >= 1 && col0 >= 0 && ncols >= 1 && col0 + (int64_t)ncols <= row_bytes,
                "mmk_host_read_rows: bad geometry (rows %d, row_bytes %d, columns [%d, %d))", rows, row_bytes, col0, col0 + ncols);
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) {
        mmk::set_error("mmk_host_read_rows: cannot open %s: %s", path, strerror(errno));
        return MMK_ERR_ARG;
    }
    struct stat st;
    const size_t need = (size_t)header_bytes + (size_t)rows * row_bytes;
    if (fstat(fd, &st) != 0 || (size_t)st.st_size < need) {
        close(fd);
        mmk::set_error("mmk_host_read_rows: %s is shorter than %zu bytes", path, need);
        return MMK_ERR_ARG;
    }
    char *out = static_cast<char *>(dst);
    int rc = MMK_OK;
    auto read_all = [&](char *to, size_t total, size_t file_off) -> bool {
        size_t done = 0;
        while (done < total) {
            const ssize_t n = pread(fd, to + done, total - done, (off_t)(file_off + done));
            if (n <= 0) {
                mmk::set_error("mmk_host_read_rows: read of %s failed: %s", path, n < 0 ? strerror(errno) : "short file");
                return false;
            }
            done += (size_t)n;
        }
        return true;
    };
    int sh = roll % rows;
    if (sh < 0) sh += rows;                      // torch.roll(x, roll, dims=0): dst[(r + roll) mod rows] = src[r]
    if (col0 == 0 && ncols == row_bytes) {
        // whole rows: the rotation is two contiguous pieces -- src rows [0, rows - sh) -> dst rows [sh, rows), the rest -> [0, sh)
        const size_t head = (size_t)(rows - sh) * row_bytes;
        if (!read_all(out + (size_t)sh * row_bytes, head, (size_t)header_bytes) ||
            (sh > 0 && !read_all(out, (size_t)sh * row_bytes, (size_t)header_bytes + head)))
            rc = MMK_ERR_ARG;
    } else {
        // column cut.  A narrow cut (the 2-byte encoder column of a Navtech row: 800 bytes out of 1.3 MB) reads just its bytes,
        // one pread per row; a wide one (the 3 360 power bytes behind the 11-byte row header) reads the rows in blocks of <= 256 KB
        // into a buffer on this call's stack frame and copies the cut out of it -- no mmap (mapping and faulting take the
        // process-wide mm lock, which serialises the loader's threads) and no per-thread heap buffer that the batch call's
        // short-lived threads would allocate and fault in again for every batch.
        if ((size_t)ncols * 8 <= (size_t)row_bytes) {
            for (int r = 0; r < rows && rc == MMK_OK; ++r) {
                int d = r + sh;
                if (d >= rows) d -= rows;
                if (!read_all(out + (size_t)d * ncols, (size_t)ncols, (size_t)header_bytes + (size_t)r * row_bytes + (size_t)col0)) rc = MMK_ERR_ARG;
            }
        } else {
            constexpr size_t BLOCK = 256 * 1024;
            char buf[BLOCK];
            const int rows_per = (int)std::max<size_t>(1, BLOCK / (size_t)row_bytes);
            if ((size_t)row_bytes > BLOCK) {
                mmk::set_error("mmk_host_read_rows: rows of %d bytes are longer than the %zu-byte staging block", row_bytes, BLOCK);
                rc = MMK_ERR_ARG;
            }
            for (int r0 = 0; r0 < rows && rc == MMK_OK; r0 += rows_per) {
                const int nr = std::min(rows_per, rows - r0);
                if (!read_all(buf, (size_t)nr * row_bytes, (size_t)header_bytes + (size_t)r0 * row_bytes)) {
                    rc = MMK_ERR_ARG;
                    break;
                }
                for (int r = 0; r < nr; ++r) {
                    int d = r0 + r + sh;
                    if (d >= rows) d -= rows;
                    memcpy(out + (size_t)d * ncols, buf + (size_t)r * row_bytes + col0, (size_t)ncols);
                }
            }
        }
    }
    close(fd);
    return rc;
}

// A whole batch in one call: the jobs (one per tensor and item) are drawn from a shared counter by `threads` host threads of
// this call's own (created here, joined before returning: no pool, no state left behind).  The Python side then needs ONE
// interpreter thread per loader instead of one per worker -- with eight Python worker threads the training thread's 2.5 ms of
// enqueue work per step kept waiting for the interpreter lock and a loader-fed step ran 25 % below the step's own rate.
extern "C" int mmk_host_read_rows_batch(const mmk_read_job *jobs, int32_t n_jobs, int32_t threads)
{
    MMK_REQUIRE(jobs && n_jobs >= 1, "mmk_host_read_rows_batch: no jobs");
    const int nt = std::max(1, std::min<int>(threads, n_jobs));
    std::atomic<int> next(0), failed(0);
    std::string first_error;
    std::atomic<bool> have_error(false);
    auto work = [&]() {
        for (int j = next.fetch_add(1); j < n_jobs; j = next.fetch_add(1)) {
            const mmk_read_job &q = jobs[j];
            const int rc = mmk_host_read_rows(q.path, q.header_bytes, q.rows, q.row_bytes, q.col0, q.ncols, q.roll, q.dst);
            if (rc != MMK_OK) {
                failed.fetch_add(1);
                bool expect = false;
                if (have_error.compare_exchange_strong(expect, true)) first_error = mmk_last_error();   // (the message is per thread)
            }
        }
    };
    std::vector<std::thread> pool;
    pool.reserve(nt - 1);
    for (int t = 1; t < nt; ++t) pool.emplace_back(work);
    work();
    for (auto &th : pool) th.join();
    if (failed.load() != 0) {
        mmk::set_error("mmk_host_read_rows_batch: %d of %d jobs failed; first: %s", failed.load(), n_jobs, first_error.c_str());
        return MMK_ERR_ARG;
    }
    return MMK_OK;
}

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
