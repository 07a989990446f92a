// EXPERIMENT RECORD (round 5) -- not compiled into libmmk_hip.so.
//
// conv3x3_pp_kernel: the >= 64-output-channel convolutions with the block's 8 waves split into two groups of 4 that take turns on the
// matrix cores ("ping-pong": one group runs its MFMA loop while the other does its epilogue / prefetch delivery / LDS writes, one
// block-wide barrier per half-period).  It lived in mm_masking_amd/csrc/mmk_unet.hip between launch_conv_deep() and deep_weights_fit()
// (it uses that file's helpers: ConvArgs, dropout_words, pk_keep_mask, cvt_pk_bf16, g_zero16 / g_zero32 / g_sink32, MMK_CHECK_HIP ...)
// and was dispatched from dispatch_conv_deep() for the forward / data-gradient roles of the ReLU network (MMK_CONV_PP=1).
//
// Results were BIT-IDENTICAL to conv3x3_deep_kernel on all 14 layer shapes x 2 roles (scripts/ab_lib.py md5s), no scratch memory
// (143-254 VGPRs), and it was 20-30 % SLOWER everywhere: profiles/r05_conv_pp_ab.txt (layer times) and profiles/r05_conv_pp_stamps.txt
// (s_memtime per half-period).  Why, from the stamps (64 -> 64 at 160 x 160, cycles):
//   * a group's MFMA loop takes 5 500-6 000 cycles for 180 MFMAs (2 880 at full rate): ONE wave per SIMD cannot keep the matrix
//     core fed -- its own fragment reads, prefetch address arithmetic and waits sit between its MFMAs with nobody to fill the
//     gaps.  Two waves per SIMD in the same loop (conv3x3_deep_kernel) reach 6 400 cycles for 360.
//   * the other group's epilogue takes 7 800-8 700 cycles (4 800 when both waves of a SIMD run it together): a v_mfma holds the
//     SIMD's vector issue for 8 of its 16 cycles, so the vector-heavy wave gets half the issue slots while its neighbour is
//     in the MFMA loop.  The two phases do not overlap, they slow each other down.
//   * a 4-row tile per group also reads 1.5 x its pixels as halo (1.25 x for 8 rows).
// Per pair of stages: 29 100 cycles for 2 x 4 rows against 28 600 for 8 rows in the lock-step kernel.  What would overlap phases on
// this hardware is two blocks on DIFFERENT SIMDs' worth of resources, i.e. twice the LDS.
//
// Development notes kept for whoever tries again: (1) one copy of each phase in the loop body -- with a copy per group hipcc gave the
// 80 accumulators two register homes and moved them at every join (700 B of spills); (2) fragment double-buffering must be explicit with
// a full sched_barrier per tap, or the scheduler reads three taps ahead; (3) everything a lane derives from its thread index must go
// through an opaque asm, or ~30 loop-invariant addresses per lane live in registers for the whole kernel.

// ------------------------------------------------------------------------------------------
// Round 5: the >= 64-output-channel layers of the ReLU network on two wave groups that take turns on the matrix cores.
// conv3x3_deep_kernel runs one block of 8 waves per CU through load -> barrier -> LDS write -> barrier -> MFMA -> epilogue in
// lockstep: both waves of a SIMD are in the same phase, and while the block delivers, writes or stores, the matrix cores idle
// (stamps: 6 400 of a stage's 10 700-14 000 cycles in the MFMA loop).  Here the block's 8 waves are two GROUPS of 4 (one wave of
// each group per SIMD), each with its own 4-row x NT*16-pixel tile and its own halo buffer, half a stage apart:
//      half-period 2s     group 0: MFMA loop of its stage s (with the prefetch of stage s+1 inside)     group 1: the rest of ITS
//                                                                                                     stage s-1: epilogue,
//                                                                                                     delivery, LDS write
//      half-period 2s+1   group 0: epilogue / delivery / LDS write of stage s                         group 1: MFMA loop of stage s
// with one block-wide barrier between half-periods.  A group writes its halo buffer only in its own "rest" phase, behind its own
// MFMA loop, so the halo tiles need no double buffering; the packed weights of a stage (64 output channels x 32 input channels x
// 9 taps = 36 KB) are shared by the groups and double buffered -- each group stages half of them -- or resident for the whole
// launch when all chunks fit (CIN <= 64: WRES).  LDS: 2 x 39.4 KB halo tiles (pixel pitch 80 B: conflict-free for 16-byte
// reads like 96) + 2 x 36.9 KB weights.  Same arithmetic per output element as conv3x3_deep_kernel (k order: chunk, tap), same
// dropout draws: bit-identical results (tests/test_gpu_unet_kernels.py).
constexpr int PP_THREADS = 512;
template <int NT>
struct PPCfg {
    static constexpr int MT = 4, MTB = 4, BM = 64, TH = 4, TWD = NT * 16, HT = TH + 2, WT = TWD + 2;
    static constexpr int PK = 40, NS = 9, GPP = 4;
    static constexpr int NIN = HT * WT * GPP;                    // 16-byte granules of a halo tile
    static constexpr int RIN = (NIN + 255) / 256;                 // ... a thread of the group prefetches
    static constexpr int NW = NS * MTB * 64, NWH = NW / 2;        // granules of a stage's weights; each group stages half
    static constexpr int RWH = (NWH + 255) / 256;
    static constexpr size_t IN_BYTES = (size_t)HT * WT * PK * sizeof(bf16), W_BYTES = (size_t)NW * 8 * sizeof(bf16);
    static constexpr size_t smem(int wbufs) { return 2 * IN_BYTES + (size_t)wbufs * W_BYTES + BM * sizeof(float); }
};

template <int NT, int ROLE, bool WRES, bool SUBW>
__global__ __launch_bounds__(PP_THREADS) void conv3x3_pp_kernel(const ConvArgs a, int total_tiles, int tiles_per_xcd)
{
    using C = PPCfg<NT>;
    constexpr int MT = C::MT, MTB = C::MTB, PK = C::PK, WT = C::WT, GPP = C::GPP, BM = C::BM;
    constexpr int NIN = C::NIN, NW = C::NW, NWH = C::NWH, RIN = C::RIN, RWH = WRES ? 0 : C::RWH, NS = C::NS;
    constexpr bool R_FWD = ROLE == 1, R_BWD = ROLE == 2;
    static_assert(R_FWD || R_BWD, "forward or data-gradient role");
    static_assert(!(WRES && SUBW), "resident weights: the kernel's own packing only");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wv >> 2, wn = wv & 3, gtid = tid & 255;
    const int tiles_x = (a.W + C::TWD - 1) / C::TWD, tiles_y = (a.H + C::TH - 1) / C::TH;
    const int tpi = tiles_x * tiles_y;
    const int group = blockIdx.y;
    const int nchunk = a.CIN / 32;
    const int wbufs = WRES ? nchunk : 2;

    bf16 *in_tile = reinterpret_cast<bf16 *>(smem) + (size_t)grp * (C::HT * WT * PK);       // the group's own halo tile
    bf16 *w_lds = reinterpret_cast<bf16 *>(smem) + (size_t)2 * (C::HT * WT * PK);
    float *bias_lds = reinterpret_cast<float *>(w_lds + (size_t)wbufs * NW * 8);

    // tile walk: the two groups of the XCD's blocks are 2 nb "virtual blocks"; virtual block v takes tiles t_begin + v + k 2 nb
    const int xcd = blockIdx.x & 7, nb = gridDim.x >> 3;
    const int t_begin = xcd * tiles_per_xcd;
    const int t_end = (t_begin + tiles_per_xcd < total_tiles) ? t_begin + tiles_per_xcd : total_tiles;
    const int nbv = 2 * nb;
    const int first0 = t_begin + 2 * (int)(blockIdx.x >> 3);                 // group 0's first tile
    if (first0 >= t_end) return;
    const int first = first0 + grp;
    const int n_mine = first < t_end ? (t_end - first + nbv - 1) / nbv : 0;  // tiles of this group
    const int S = ((t_end - first0 + nbv - 1) / nbv) * nchunk;               // stages of group 0 (>= the other group's)
    const int S_mine = n_mine * nchunk;

    // Staging registers.  Everything a lane derives from its thread index for a load or an LDS write (granule -> pixel -> offset)
    // is worked out where it is used, from an index the compiler cannot see through: hoisted out of the loop these ~30 values
    // per lane are live for the whole kernel, beside 80 accumulators and 72 fragment registers, and get spilled.
    u32x4 rin[RIN], rwh[RWH > 0 ? RWH : 1];
    auto opaque_gtid = [&]() {
        int tv = gtid;
        asm volatile("" : "+v"(tv));
        return tv;
    };
    const bf16 *ld_base = nullptr;
    int ld_xc = 0, ld_ty0 = 0, ld_tx0 = 0;
    const u32x4 *ld_w = nullptr;
    int ld_moff = 0;
    auto set_in_stage = [&](int t, int chunk) {
        const int b = t / tpi, tr = t - b * tpi;
        const int tyi = tr / tiles_x;
        ld_tx0 = (tr - tyi * tiles_x) * C::TWD;
        ld_ty0 = tyi * C::TH;
        const int c0 = chunk * 32;
        const bool in1 = c0 < a.C1;                 // a 32-channel chunk lies entirely in one of the two concatenated inputs
        const bf16 *xb = in1 ? a.x1 : a.x2;
        ld_xc = in1 ? a.C1 : a.C2;
        const int cb = in1 ? c0 : c0 - a.C1;
        const long org = ((long)b * a.H + ld_ty0 - 1) * a.W + ld_tx0 - 1;      // halo origin pixel (may lie outside the image)
        ld_base = xb + org * ld_xc + cb;
    };
    auto set_w_stage = [&](int chunk) {
        if constexpr (SUBW) {
            const int pm = a.wpack_mtb, per = pm / MTB;           // tiles per packed group, kernel groups per packed group
            ld_w = reinterpret_cast<const u32x4 *>(a.wpack + ((size_t)((group / per) * nchunk + chunk)) * NS * pm * 512);
            ld_moff = (group % per) * MTB;
        } else {
            ld_w = reinterpret_cast<const u32x4 *>(a.wpack + ((size_t)(group * nchunk + chunk)) * NW * 8);
        }
    };
    auto issue_in = [&](int i) {               // (i: a compile-time constant at every call)
        int g = opaque_gtid() + i * 256;
        g = g < NIN ? g : NIN - 1;
        const int pix = g / GPP;
        const int dy1 = pix / WT, dx1 = pix % WT, cc = (g % GPP) * 8;
        const bool ok = (unsigned)(ld_ty0 + dy1 - 1) < (unsigned)a.H && (unsigned)(ld_tx0 + dx1 - 1) < (unsigned)a.W;
        const unsigned off = (unsigned)((dy1 * a.W + dx1) * ld_xc + cc);
        const u32x4 *sp = ok ? reinterpret_cast<const u32x4 *>(ld_base + off) : &g_zero16;
        rin[i] = *sp;
    };
    auto w_src = [&](int g) -> const u32x4 * {        // granule g of the stage's weights in global memory
        if constexpr (SUBW) return ld_w + ((g / (MTB * 64)) * a.wpack_mtb + ld_moff) * 64 + g % (MTB * 64);
        else return ld_w + g;
    };
    auto issue_w = [&](int i) {                // this group's half of the stage
        int g = grp * NWH + opaque_gtid() + i * 256;
        g = g < (grp + 1) * NWH ? g : (grp + 1) * NWH - 1;
        rwh[i] = *w_src(g);
    };
    auto write_in = [&]() {
        const int tv = opaque_gtid();
#pragma unroll
        for (int i = 0; i < RIN; ++i) {
            const int g = tv + i * 256;
            if (g < NIN) *reinterpret_cast<u32x4 *>(in_tile + (size_t)(g / GPP) * PK + (g % GPP) * 8) = rin[i];
        }
    };
    auto write_w = [&](int buf) {
        const int tv = opaque_gtid();
#pragma unroll
        for (int i = 0; i < RWH; ++i) {
            const int g = grp * NWH + tv + i * 256;
            if (g < (grp + 1) * NWH) reinterpret_cast<u32x4 *>(w_lds + (size_t)buf * NW * 8)[g] = rwh[i];
        }
    };

    const DropoutParams dp = dropout_params(a.drop_p);
    if (tid < BM) {
        const int c = group * BM + tid;
        const float bz = (a.bias && c < a.COUT) ? a.bias[c] : 0.f;
        bias_lds[tid] = R_FWD ? bz * dp.inv_keep : bz;
    }
    // ---- prologue: the weights of stages 0 and 1 (or of every chunk: WRES) by all 512 threads, each group's first halo tile
    for (int ck = 0; ck < wbufs; ++ck) {
        set_w_stage(ck % nchunk);
        for (int g = tid; g < NW; g += PP_THREADS) reinterpret_cast<u32x4 *>(w_lds + (size_t)ck * NW * 8)[g] = *w_src(g);
    }
    int tile = n_mine > 0 ? first : first0, chunk = 0;      // (a group without tiles walks group 0's first tile: loads stay in bounds)
    set_in_stage(tile, 0);
#pragma unroll
    for (int i = 0; i < RIN; ++i) issue_in(i);
    write_in();

    f32x4 acc[MT][NT];
    auto reset_acc = [&]() {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    reset_acc();

    constexpr int NCH = 4 * MT, LW = 8, NP = NCH / LW;
    typedef __attribute__((ext_vector_type(LW))) __bf16 bfl;

    // what the "rest" phase has to finish: the stage the group's last MFMA loop was for
    int ep_tile = tile;
    bool ep_last = false, ep_valid = false;

    // ---- the MFMA loop of stage sg (the group's own stage counter), with the prefetch of the next stage inside
    auto mfma_phase = [&](int sg) {
        int ntile = tile, nck = chunk + 1;
        if (nck == nchunk) {
            nck = 0;
            ntile = tile + nbv;
        }
        const bool has_next = sg + 1 < S_mine;
        set_in_stage(has_next ? ntile : tile, has_next ? nck : chunk);      // (clamped: the loads stay unconditional)
        int lq = lane;
        asm volatile("" : "+v"(lq));
        const bf16 *b_base = in_tile + ((size_t)(wn * WT + (lq & 15))) * PK + 8 * (lq >> 4);
        const bf16 *a_base = w_lds + (size_t)lq * 8 + (size_t)(WRES ? chunk : (sg & 1)) * NW * 8;
        // Only ONE wave of a SIMD is in its MFMA loop at a time, so nobody else hides this wave's LDS latency: the fragments of
        // tap s + 1 are read while tap s is on the matrix cores, in two explicit fragment sets, and a full scheduling fence per
        // tap keeps the compiler from hoisting more than that (left free to, it reads three taps ahead and spills the prefetch).
        bf16x8 bf[2][NT], af[2][MT];
        auto read_frags = [&](int s2, int set) {
            const int ty = s2 / 3, tx = s2 % 3;
#pragma unroll
            for (int n = 0; n < NT; ++n)
                bf[set][n] = *reinterpret_cast<const bf16x8 *>(b_base + ((size_t)(ty * WT + n * 16 + tx)) * PK);
#pragma unroll
            for (int m = 0; m < MT; ++m)
                af[set][m] = *reinterpret_cast<const bf16x8 *>(a_base + ((size_t)(s2 * MTB + m) * 64) * 8);
        };
        read_frags(0, 0);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (s + 1 < NS) read_frags(s + 1, (s + 1) & 1);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[s & 1][m], bf[s & 1][n], acc[m][n], 0, 0, 0);
            if (s < RIN) issue_in(s);
            __builtin_amdgcn_sched_barrier(0);
        }
        static_assert(RIN <= NS, "one halo prefetch load per tap");
        ep_tile = tile;
        ep_last = chunk == nchunk - 1;
        ep_valid = sg < S_mine;
        if (has_next) {
            tile = ntile;
            chunk = nck;
        }
    };

    // ---- the rest of a stage: epilogue of a finished tile, delivery of the prefetch, LDS writes
    auto rest_phase = [&](int sg) {
        // this group's half of the weights of stage sg + 1 (group 0) / sg + 2 (group 1): fetched here, written at the end of the
        // phase (they are live while no fragment registers are)
        if constexpr (!WRES) {
            set_w_stage((sg + 1 + grp) % nchunk);
#pragma unroll
            for (int i = 0; i < RWH; ++i) issue_w(i);
        }
        if (ep_last) {
            if (ep_valid) {
                const int b = ep_tile / tpi, tr = ep_tile - b * tpi;
                const int tyi = tr / tiles_x;
                const int tx0 = (tr - tyi * tiles_x) * C::TWD, yy = tyi * C::TH + wn;
                int lv = lane;
                asm volatile("" : "+v"(lv));        // (derived here, once per tile, instead of living in registers across the MFMA loop)
                const int g4 = lv >> 4;
                const int c0 = group * BM + NCH * g4, e_bias_at = NCH * g4;
                const bool e_on = c0 < a.COUT;
                const bool firstp = c0 < a.o1.C;
                bf16 *const o_y = firstp ? a.o1.y : a.o2.y;
                const bf16 *const o_src = firstp ? a.o1.relu_src : a.o2.relu_src;
                const int o_C = firstp ? a.o1.C : a.o2.C;
                const bool e_acc = (firstp ? a.o1.accumulate : a.o2.accumulate) != 0;
                const float e_scale = firstp ? a.o1.scale : a.o2.scale;
                const int cl = firstp ? c0 : c0 - a.o1.C;
                const int e_x0 = tx0 + (lv & 15);
                const long p0 = ((long)b * a.H + yy) * a.W + e_x0;                 // the lane's pixel of n-tile 0
                const unsigned e_e0 = (unsigned)p0 * (unsigned)a.COUT + (unsigned)c0 + a.hash_base;       // dropout element index
                bf16 *const e_y0 = o_y + p0 * o_C + cl;
                const int e_step = 16 * o_C;                                        // elements between the pixels of consecutive n-tiles
                const bool e_row_ok = yy < a.H;
                const bool e_src = o_src != nullptr;
                const bf16 *const e_pf0 = e_src ? o_src + p0 * o_C + cl : (e_acc ? (const bf16 *)e_y0 : nullptr);
                bfl opnd[R_BWD ? NT : 1][NP];
                if constexpr (R_BWD) {
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        const bool okp = e_on && e_pf0 != nullptr && e_row_ok && e_x0 + n * 16 < a.W;
                        const bf16 *pp = okp ? e_pf0 + n * e_step : reinterpret_cast<const bf16 *>(g_zero32);
#pragma unroll
                        for (int k = 0; k < NP; ++k) opnd[n][k] = *reinterpret_cast<const bfl *>(pp + k * LW);
                    }
                }
                if (e_on) {
                    if constexpr (R_FWD) {
                        float bk[NCH];
#pragma unroll
                        for (int k = 0; k < NCH; k += 4) {
                            const f32x4 t = *reinterpret_cast<const f32x4 *>(bias_lds + e_bias_at + k);
                            bk[k] = t[0]; bk[k + 1] = t[1]; bk[k + 2] = t[2]; bk[k + 3] = t[3];
                        }
                        const float kf = dp.inv_keep;
                        const int t1s = (int)dp.thr - 32768 - 1;
                        const unsigned t1 = (unsigned)(t1s < -32768 ? -32768 : t1s) & 0xffffu;
                        const unsigned thr1 = t1 | (t1 << 16);
                        const bool drop = a.drop_p > 0.f;
#pragma unroll
                        for (int n = 0; n < NT; ++n) {
                            const bool okp = e_row_ok && e_x0 + n * 16 < a.W;
                            unsigned hw[MT][2];
#pragma unroll
                            for (int m = 0; m < MT; ++m) hw[m][0] = hw[m][1] = 0x7fff7fffu;
                            if (drop) {
#pragma unroll
                                for (int m = 0; m < MT; ++m)
                                    dropout_words(a.seed, e_e0 + (unsigned)n * (16u * (unsigned)a.COUT) + 4u * m, hw[m][0], hw[m][1]);
                            }
                            unsigned pk[NCH / 2];
#pragma unroll
                            for (int m = 0; m < MT; ++m)
#pragma unroll
                                for (int hlf = 0; hlf < 2; ++hlf) {
                                    const float t0 = __builtin_fmaf(acc[m][n][2 * hlf], kf, bk[4 * m + 2 * hlf]);
                                    const float t1f = __builtin_fmaf(acc[m][n][2 * hlf + 1], kf, bk[4 * m + 2 * hlf + 1]);
                                    pk[2 * m + hlf] = pk_relu_bf16(cvt_pk_bf16(t0, t1f)) & pk_keep_mask(hw[m][hlf], thr1);
                                }
                            bf16 *const dst = okp ? e_y0 + n * e_step : reinterpret_cast<bf16 *>(g_sink32);
#pragma unroll
                            for (int k = 0; k < NCH / 2; k += 4)
                                *reinterpret_cast<u32x4 *>(dst + 2 * k) = (u32x4){pk[k], pk[k + 1], pk[k + 2], pk[k + 3]};
                            __builtin_amdgcn_sched_barrier(0);     // one n-tile at a time: bounds the live temporaries
                        }
                    } else {
                        const bool o_accf = !e_src && e_acc;            // the operand slots hold the accumulate target
#pragma unroll
                        for (int n = 0; n < NT; ++n) {
                            const bool okp = e_row_ok && e_x0 + n * 16 < a.W;
                            float v[NCH];
#pragma unroll
                            for (int m = 0; m < MT; ++m)
#pragma unroll
                                for (int r = 0; r < 4; ++r) v[4 * m + r] = acc[m][n][r];
                            if (e_pf0 != nullptr && okp) {
                                if (!o_accf) {
#pragma unroll
                                    for (int k = 0; k < NCH; k += LW) {
                                        const bfl sv = opnd[n][k / LW];
#pragma unroll
                                        for (int r = 0; r < LW; ++r) v[k + r] = ((float)sv[r] > 0.f) ? v[k + r] * e_scale : 0.f;
                                    }
                                } else {
#pragma unroll
                                    for (int k = 0; k < NCH; k += LW) {
                                        const bfl ov = opnd[n][k / LW];
#pragma unroll
                                        for (int r = 0; r < LW; ++r) v[k + r] += (float)ov[r];
                                    }
                                }
                            }
                            bf16 *const dst = okp ? e_y0 + n * e_step : reinterpret_cast<bf16 *>(g_sink32);
#pragma unroll
                            for (int k = 0; k < NCH; k += 8) {
                                bf16x8 o8;
#pragma unroll
                                for (int r = 0; r < 8; ++r) o8[r] = (bf16)v[k + r];
                                *reinterpret_cast<bf16x8 *>(dst + k) = o8;
                            }
                        }
                    }
                }
            }
            reset_acc();
        }
        // delivery of the prefetch, then the LDS writes: the group's own halo tile (its MFMA loop is over), its half of the
        // weights of stage sg + 1 (group 0) / sg + 2 (group 1) into the buffer nobody reads in this half-period
#pragma unroll
        for (int i = 0; i < RIN; ++i) asm volatile("" : "+v"(rin[i]));
#pragma unroll
        for (int i = 0; i < RWH; ++i) asm volatile("" : "+v"(rwh[i]));
        write_in();
        if constexpr (!WRES) write_w((sg + 1 + grp) & 1);
    };

    // half-periods h = 0 .. 2 S: group 0 runs the MFMA loop of stage h / 2 at even h and the rest of it at the next odd h; group 1
    // is one half-period behind (MFMA loop of stage (h - 1) / 2 at odd h, its rest at the next even h, the last one at h = 2 S).
    // (ONE copy of each phase in the loop body: with a copy per group the register allocator gave the accumulators two
    // homes and moved all 80 of them between the copies, spilling the prefetch.)
#ifdef MMK_DEEP_STAMPS
    // diagnostic build: per wave and half-period (the first 32), s_memtime behind the barrier and at the end of the phase, and
    // what the phase was (1 = MFMA loop, 2 = rest, 3 = rest with an epilogue)
    unsigned long long *stamp_lds = reinterpret_cast<unsigned long long *>(smem + (C::smem(wbufs) + 15) / 16 * 16);
    for (int i = tid; i < 8 * 32 * 4; i += PP_THREADS) stamp_lds[i] = 0ull;
#endif
    for (int h = 0; h <= 2 * S; ++h) {
        __syncthreads();
#ifdef MMK_DEEP_STAMPS
        const unsigned long long ts0 = __builtin_amdgcn_s_memtime();
        unsigned long long kind = 0;
#endif
        const int sg = (h - grp) >> 1;              // the group's stage of this half-period
        if (((h ^ grp) & 1) == 0) {
            if (sg < S) mfma_phase(sg);
#ifdef MMK_DEEP_STAMPS
            kind = 1;
#endif
        } else if (h > grp) {
#ifdef MMK_DEEP_STAMPS
            kind = ep_last ? 3 : 2;
#endif
            rest_phase(sg);
        }
#ifdef MMK_DEEP_STAMPS
        if (h < 32 && lane == 0) {
            unsigned long long *sl = stamp_lds + ((size_t)wv * 32 + h) * 4;
            sl[0] = ts0; sl[1] = __builtin_amdgcn_s_memtime(); sl[2] = kind;
        }
#endif
    }
#ifdef MMK_DEEP_STAMPS
    __syncthreads();
    if (g_deep_stamp_buf != nullptr && blockIdx.y == 0 && blockIdx.x < 64)
        for (int i = tid; i < 8 * 32 * 4; i += PP_THREADS) g_deep_stamp_buf[(size_t)blockIdx.x * (8 * 32 * 4) + i] = stamp_lds[i];
#endif
}

template <int NT, int ROLE, bool WRES, bool SUBW>
int launch_conv_pp(const ConvArgs &a, hipStream_t st)
{
    using C = PPCfg<NT>;
    static bool attr_set[64] = {};
    int dev = 0;
    MMK_CHECK_HIP(hipGetDevice(&dev));
    if (!attr_set[dev & 63]) {
        MMK_CHECK_HIP(hipFuncSetAttribute((const void *)conv3x3_pp_kernel<NT, ROLE, WRES, SUBW>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set[dev & 63] = true;
    }
#ifdef MMK_DEEP_STAMPS
    const size_t smem = (C::smem(WRES ? a.CIN / 32 : 2) + 15) / 16 * 16 + (size_t)8 * 32 * 4 * sizeof(unsigned long long);
#else
    const size_t smem = C::smem(WRES ? a.CIN / 32 : 2);
#endif
    MMK_REQUIRE(smem <= (size_t)160 * 1024, "mmk_conv3x3: %zu bytes of LDS for CIN=%d COUT=%d", smem, a.CIN, a.COUT);
    const int tiles = ((a.W + C::TWD - 1) / C::TWD) * ((a.H + C::TH - 1) / C::TH);
    const int groups = (a.COUT + 63) / 64;
    const int total = tiles * a.B;
    const int per_xcd = (total + 7) / 8;
    int nb = 32 / groups;                                    // one 8-wave block per CU, 32 CUs per XCD
    nb = nb < 1 ? 1 : nb;
    nb = std::min(nb, (per_xcd + 1) / 2);                    // (two tiles per block at least)
    hipLaunchKernelGGL((conv3x3_pp_kernel<NT, ROLE, WRES, SUBW>), dim3(8 * nb, groups), dim3(PP_THREADS), smem, st, a, total, per_xcd);
    MMK_LAUNCH_CHECK();
    return MMK_OK;
}

// which launches take the two-group kernel: forward / data-gradient roles of the ReLU network with >= 64 output channels
// (MMK_CONV_PP=0: conv3x3_deep_kernel for everything, the A/B switch)
bool use_conv_pp()
{
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("MMK_CONV_PP");
        v = (e && e[0] == '0') ? 0 : 1;
    }
    return v == 1;
}

int dispatch_conv_pp(const ConvArgs &a, int role, hipStream_t st)
{
    const bool narrow = a.W <= 48;
    const bool subw = conv_cm(a.CIN, a.COUT) == 128;          // packed for 128-channel blocks: this kernel's 64 are a slice
    ConvArgs c = a;
    if (subw) c.wpack_mtb = 8;
    const bool wres = !subw && a.CIN <= 64;
#define MMK_PP_CASE(R)                                                                                                   \
    if (role == R) {                                                                                                     \
        if (subw) return narrow ? launch_conv_pp<3, R, false, true>(c, st) : launch_conv_pp<5, R, false, true>(c, st);   \
        if (wres) return narrow ? launch_conv_pp<3, R, true, false>(c, st) : launch_conv_pp<5, R, true, false>(c, st);   \
        return narrow ? launch_conv_pp<3, R, false, false>(c, st) : launch_conv_pp<5, R, false, false>(c, st);           \
    }
    MMK_PP_CASE(1);
    MMK_PP_CASE(2);
#undef MMK_PP_CASE
    mmk::set_error("mmk_conv3x3: no two-group kernel for this role");
    return MMK_ERR_ARG;
}

