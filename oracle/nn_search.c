/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported, linked or executed by the
 * product path (mm_masking_amd/).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library.
 *
 * CPU restatement of stage I2 (brute-force nearest neighbour) of the dICP hot
 * path.  The reference's own arithmetic lives in the third-party package
 * lisusdaniil/dICP (un-vendored, un-pinned: /root/reference/.gitmodules:4-6,
 * requirements.txt:11), which is ABSENT from /root/reference -> parity at this
 * boundary is UNPINNED; this file restates the published algorithm (exhaustive
 * argmin of squared euclidean distance) anchored on the reference call site
 * mm_masking/icp_weight_policy.py:281-287 (source (B,N,3), target (B,M,6|3),
 * dim=2|3).
 *
 * Normative arithmetic (the HIP kernel mmk_nn_search reproduces it bit for bit):
 *   dx = t_x - p_x ; dy = t_y - p_y ; (dz = t_z - p_z)
 *   dim 2:  d = fmaf(dy, dy, dx*dx)
 *   dim 3:  d = fmaf(dz, dz, fmaf(dy, dy, dx*dx))
 *   scan j ascending, replace on strict d < best  (ties -> lowest index)
 * Build with -ffp-contract=off so that only the explicit fmaf() calls fuse.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

/* p: (B,N,d) packed transformed source, t: (B,M,d) packed target. */
void mmk_oracle_nn_search(const float *p, const float *t, int B, int N, int M,
                          int d, int32_t *idx, float *d2)
{
    for (int b = 0; b < B; ++b) {
        const float *pb = p + (size_t)b * N * d;
        const float *tb = t + (size_t)b * M * d;
#pragma omp parallel for schedule(static)
        for (int i = 0; i < N; ++i) {
            float best = INFINITY;
            int32_t bi = 0;
            const float px = pb[(size_t)i * d + 0];
            const float py = pb[(size_t)i * d + 1];
            if (d == 2) {
                for (int j = 0; j < M; ++j) {
                    float dx = tb[(size_t)j * 2 + 0] - px;
                    float dy = tb[(size_t)j * 2 + 1] - py;
                    float dist = fmaf(dy, dy, dx * dx);
                    if (dist < best) { best = dist; bi = j; }
                }
            } else {
                const float pz = pb[(size_t)i * 3 + 2];
                for (int j = 0; j < M; ++j) {
                    float dx = tb[(size_t)j * 3 + 0] - px;
                    float dy = tb[(size_t)j * 3 + 1] - py;
                    float dz = tb[(size_t)j * 3 + 2] - pz;
                    float dist = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
                    if (dist < best) { best = dist; bi = j; }
                }
            }
            idx[(size_t)b * N + i] = bi;
            d2[(size_t)b * N + i] = best;
        }
    }
}

/* I1: s' = R s + t in fp32 with every product and sum individually rounded
 * (left to right), dim 2 uses the x,y block of the 4x4 pose only.
 * src: (B,N,3) as handed over by the reference (icp_weight_dataset.py:379-381),
 * T: (B,16) row-major 4x4, out: (B,N,d) packed. */
void mmk_oracle_transform(const float *src, const float *T, int B, int N, int d,
                          float *out)
{
    for (int b = 0; b < B; ++b) {
        const float *Tb = T + (size_t)b * 16;
        for (int i = 0; i < N; ++i) {
            const float *s = src + ((size_t)b * N + i) * 3;
            float *o = out + ((size_t)b * N + i) * d;
            if (d == 2) {
                o[0] = (Tb[0] * s[0] + Tb[1] * s[1]) + Tb[3];
                o[1] = (Tb[4] * s[0] + Tb[5] * s[1]) + Tb[7];
            } else {
                o[0] = ((Tb[0] * s[0] + Tb[1] * s[1]) + Tb[2] * s[2]) + Tb[3];
                o[1] = ((Tb[4] * s[0] + Tb[5] * s[1]) + Tb[6] * s[2]) + Tb[7];
                o[2] = ((Tb[8] * s[0] + Tb[9] * s[1]) + Tb[10] * s[2]) + Tb[11];
            }
        }
    }
}

/*
 * Four-tap blend of F.grid_sample(bilinear) in fp64 as PyTorch's CPU kernel evaluates it
 * (ATen/native/cpu/GridSamplerKernel.cpp, compiled with FMA contraction): the chain
 *   fma(se, w_se, fma(sw, w_sw, fma(ne, w_ne, nw * w_nw)))
 * — found by comparing candidate orders with the reference's radar_cartesian_to_polar
 * (mm_masking/radar_utils.py:338-372) on random input: only this one is bit-identical.
 */
void mmk_oracle_blend4_f64(const double *t0, const double *t1, const double *t2, const double *t3,
                           const double *w0, const double *w1, const double *w2, const double *w3,
                           long n, double *out)
{
    for (long i = 0; i < n; ++i)
        out[i] = fma(t3[i], w3[i], fma(t2[i], w2[i], fma(t1[i], w1[i], t0[i] * w0[i])));
}
