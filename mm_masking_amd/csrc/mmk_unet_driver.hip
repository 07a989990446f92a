// The mask U-Net as two C-ABI calls: mmk_unet_forward / mmk_unet_backward sequence every launch of the
// network (mm_masking/icp_weight_policy.py:161-184 forward, train_icp_weights.py:51 loss.backward()) on the
// host side of the ABI, from one descriptor.  The kernels are the building blocks of mmk_unet.hip, reached
// through their own C entry points (argument checks included); this file only owns the schedule and the
// layout of the activations in the caller's workspace.  A training step is ~290 launches: issued from C++
// they cost ~1 ms of host time instead of ~6 ms through one ctypes call each (bench.py: "host enqueue time").
//
// Layout (all offsets are functions of (B, H, W, cin) alone, so that the backward pass recomputes them):
//   workspace (kept from forward to backward):  packed bf16 weights of the 21 3x3 layers | every conv output,
//       pooled tensor and up-sampled tensor of the network (NHWC bf16) | raw mask, per-image maximum (fp32)
//   scratch (backward only): the transposed packed weights | every gradient tensor (NHWC bf16) | per-layer
//       partial-sum slices of the weight gradients (fp32) | small reduction buffers
// Nothing is re-used inside a pass: 288 GB of HBM make a bump allocation of ~10 GB at B = 32 the simplest
// correct choice (no aliasing hazards between the ~290 asynchronous launches).
#include <stdlib.h>
#include <string.h>

#include "mmk_common.h"

namespace {

constexpr int NCONV = 23;                       // conv k: 0..11 encoder (2 per block), 12..21 decoder, 22 final
const int ENC_CH[6] = {8, 16, 32, 64, 128, 256};

struct Tens {
    size_t off = 0;                             // byte offset
    int h = 0, w = 0, c = 0;
};

struct Plan {
    int B, H, W, cin;
    int rh[6], rw[6];                           // resolution of t[i]
    int cout[NCONV], cinn[NCONV];               // channel counts of conv k
    // forward tensors
    Tens a_enc[6], d_enc[6], t[6];
    size_t t_arg[6];                            // arg-max codes of the pooling that produced t[i] (i >= 1): B * rh[i] * rw[i] * ch / 2 bytes
    Tens u[5], a1[5], d1[5], a2[5], d2[5];
    size_t packs[NCONV];                        // packed weights (k = 1..21)
    size_t mask_raw, amax, norm_part;
    size_t ws_bytes;
    // backward tensors
    Tens gz_final, gz_a2[5], gsk[5], gz_d1[5], gz_a1[5], g_u[5], gz_up[5];
    Tens gz_d[6], gz_a[6];
    size_t packs_t[NCONV];
    size_t part[NCONV];                         // partial slices of layer k
    int slices[NCONV];
    size_t fin_ws, fin_red, first_ws;           // workspaces of the final / first layer's gradient reductions
    size_t scratch_bytes;
};

struct Bump {
    size_t off = 0;
    size_t take(size_t bytes)
    {
        off = mmk::align_up(off, 256);
        const size_t o = off;
        off += bytes;
        return o;
    }
    Tens tens(int B, int h, int w, int c)
    {
        Tens t;
        t.h = h; t.w = w; t.c = c;
        t.off = take((size_t)B * h * w * c * 2);
        return t;
    }
};

bool make_plan(Plan &p, int B, int H, int W, int cin, bool with_scratch)
{
    if (B < 1 || H < 32 || W < 32 || cin < 1 || cin > 4) return false;
    p.B = B; p.H = H; p.W = W; p.cin = cin;
    p.rh[0] = H; p.rw[0] = W;
    for (int i = 1; i < 6; ++i) { p.rh[i] = p.rh[i - 1] / 2; p.rw[i] = p.rw[i - 1] / 2; }
    // conv k channel counts: encoder block i = convs 2i, 2i+1; decoder block j = convs 12+2j (2cs -> cs), 13+2j (cs -> cs)
    for (int i = 0; i < 6; ++i) {
        p.cinn[2 * i] = i == 0 ? cin : ENC_CH[i - 1]; p.cout[2 * i] = ENC_CH[i];
        p.cinn[2 * i + 1] = ENC_CH[i]; p.cout[2 * i + 1] = ENC_CH[i];
    }
    for (int j = 0; j < 5; ++j) {
        const int cs = ENC_CH[4 - j];
        p.cinn[12 + 2 * j] = 2 * cs; p.cout[12 + 2 * j] = cs;
        p.cinn[13 + 2 * j] = cs; p.cout[13 + 2 * j] = cs;
    }
    p.cinn[22] = 8; p.cout[22] = 1;
    Bump b;
    for (int k = 1; k <= 21; ++k) {
        const size_t n = mmk_conv3x3_packed_elems(p.cout[k], p.cinn[k], 0);
        if (n == 0) return false;
        p.packs[k] = b.take(n * 2);
    }
    p.a_enc[0] = b.tens(B, H, W, 8);
    p.d_enc[0] = b.tens(B, H, W, 8);
    p.t[0] = p.d_enc[0];
    for (int i = 1; i < 6; ++i) {
        p.a_enc[i] = b.tens(B, p.rh[i - 1], p.rw[i - 1], ENC_CH[i]);
        p.d_enc[i] = b.tens(B, p.rh[i - 1], p.rw[i - 1], ENC_CH[i]);
        p.t[i] = b.tens(B, p.rh[i], p.rw[i], ENC_CH[i]);
        p.t_arg[i] = b.take((size_t)B * p.rh[i] * p.rw[i] * ENC_CH[i] / 2);
    }
    for (int j = 0; j < 5; ++j) {
        const int cs = ENC_CH[4 - j], h = p.rh[4 - j], w = p.rw[4 - j];
        p.u[j] = b.tens(B, h, w, 2 * cs);
        p.a1[j] = b.tens(B, h, w, cs);
        p.d1[j] = b.tens(B, h, w, cs);
        p.a2[j] = b.tens(B, h, w, cs);
        p.d2[j] = b.tens(B, h, w, cs);
    }
    p.mask_raw = b.take((size_t)B * H * W * 4);
    p.amax = b.take((size_t)B * 4);
    p.norm_part = b.take((size_t)B * 64 * 4);
    p.ws_bytes = mmk::align_up(b.off, 256);
    if (!with_scratch) return true;

    Bump s;
    for (int k = 1; k <= 21; ++k) p.packs_t[k] = s.take(mmk_conv3x3_packed_elems(p.cout[k], p.cinn[k], 1) * 2);
    p.gz_final = s.tens(B, H, W, 8);
    for (int j = 0; j < 5; ++j) {
        const int cs = ENC_CH[4 - j], h = p.rh[4 - j], w = p.rw[4 - j];
        p.gz_a2[j] = s.tens(B, h, w, cs);
        p.gsk[j] = s.tens(B, h, w, cs);
        p.gz_d1[j] = s.tens(B, h, w, cs);
        p.gz_a1[j] = s.tens(B, h, w, cs);
        p.g_u[j] = s.tens(B, h, w, 2 * cs);
        p.gz_up[j] = s.tens(B, p.rh[5 - j], p.rw[5 - j], 2 * cs);     // up-sampling adjoint: the lower level's size
    }
    for (int i = 1; i < 6; ++i) {
        p.gz_d[i] = s.tens(B, p.rh[i - 1], p.rw[i - 1], ENC_CH[i]);
        p.gz_a[i] = s.tens(B, p.rh[i - 1], p.rw[i - 1], ENC_CH[i]);
    }
    p.gz_a[0] = s.tens(B, H, W, 8);
    for (int k = 1; k <= 21; ++k) {
        // spatial extent / first-input split of layer k as the backward pass calls the weight-gradient kernels
        int h, w, c1;
        if (k < 12) {
            const int i = k / 2;
            h = i == 0 ? H : p.rh[i - 1]; w = i == 0 ? W : p.rw[i - 1];
            c1 = p.cinn[k];
        } else {
            const int j = (k - 12) / 2;
            h = p.rh[4 - j]; w = p.rw[4 - j];
            c1 = p.cinn[k];       // (the second application of a decoder block's first conv splits its input cs + cs:
                                  //  the slice count does not depend on the split for these shapes, checked at run time)
        }
        p.slices[k] = mmk_conv3x3_wgrad_slices(p.cout[k], p.cinn[k], c1, B, h, w);
    }
    for (int k = 1; k <= 21; ++k) {
        if (p.slices[k] <= 0) return false;      // every layer shape of the network has a partial-sum weight-gradient kernel
        // (a decoder convolution is applied twice per pass: each application writes slices of its own -- accumulating into
        // the first application's slices was a read-modify-write of up to 38 MB at the tail of the kernel, +36 us per launch)
        p.part[k] = s.take((size_t)(k >= 12 ? 2 : 1) * p.slices[k] * ((size_t)9 * p.cout[k] * p.cinn[k] + p.cout[k]) * 4);
    }
    p.fin_ws = s.take((size_t)B * 130 * 4);
    p.fin_red = s.take((size_t)MMK_FINAL_BWD_WS_FLOATS * 4);
    p.first_ws = s.take(mmk_conv_first_wgrad_ws_bytes(cin));
    p.scratch_bytes = mmk::align_up(s.off, 256);
    return true;
}

inline void *at(void *base, size_t off) { return static_cast<char *>(base) + off; }

#define MMK_TRY(expr)                \
    do {                             \
        const int rc_ = (expr);      \
        if (rc_ != MMK_OK) return rc_; \
    } while (0)

struct ConvCall {
    const void *x1 = nullptr, *x2 = nullptr;
    int C1 = 0, C2 = 0;
    const void *wpack = nullptr;
    const float *bias = nullptr;
    void *y1 = nullptr, *y2 = nullptr;
    const void *src1 = nullptr, *src2 = nullptr;
    int O1 = 0, O2 = 0, acc1 = 0, acc2 = 0;
    float scale1 = 1.f, scale2 = 1.f;
    int relu = 0;
    float drop_p = 0.f;
    unsigned seed = 0;
    void *pool_y = nullptr, *pool_arg = nullptr;
};

int conv(const Plan &p, int h, int w, float slope, const ConvCall &c, void *stream)
{
    mmk_conv_desc d;
    memset(&d, 0, sizeof(d));
    d.x1 = c.x1; d.x2 = c.x2; d.C1 = c.C1; d.C2 = c.C2; d.wpack = c.wpack; d.bias = c.bias;
    d.y1 = c.y1; d.relu_src1 = c.src1; d.O1 = c.O1; d.accumulate1 = c.acc1; d.scale1 = c.scale1;
    d.y2 = c.y2; d.relu_src2 = c.src2; d.O2 = c.O2; d.accumulate2 = c.acc2; d.scale2 = c.scale2;
    d.B = p.B; d.H = h; d.W = w; d.relu = c.relu; d.leaky_slope = slope; d.drop_p = c.drop_p; d.seed = c.seed;
    d.pool_y = c.pool_y; d.pool_arg = c.pool_arg;
    return mmk_conv3x3(&d, stream);
}

// Side stream of the backward pass: the 38 weight-gradient launches are leaves of the dependency graph (inputs:
// a stored activation and a gradient tensor of the main chain; output: partial sums nobody reads before the final
// reduction), so they run beside the data-gradient chain instead of inside it -- kernel tails, launch boundaries
// and the under-filled launches of the 20 x 20 / 40 x 40 levels overlap.  Forked and joined with events, so the
// caller still sees one stream-ordered call; one side stream per host thread and device, created on first use.
// MMK_UNET_SIDE_STREAM=0 keeps everything on the caller's stream (same results either way: every weight-gradient
// slice is written by exactly one launch sequence in program order).
struct SideStream {
    hipStream_t st = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
    int dev = -1;
};

bool use_side_stream()
{
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("MMK_UNET_SIDE_STREAM");
        v = (e && e[0] == '0') ? 0 : 1;
    }
    return v == 1;
}

int side_stream(SideStream **out)
{
    static thread_local SideStream pool[16];
    int dev = 0;
    MMK_CHECK_HIP(hipGetDevice(&dev));
    SideStream &s = pool[dev & 15];
    if (s.st == nullptr || s.dev != dev) {
        // Highest priority -- not for the priority's sake (it changes nothing in a process with few streams) but because ROCm maps
        // the streams of a priority class onto that class's own hardware queues: once a process group exists (RCCL's streams, the
        // communication stream of ddp.FlatGradSync) a default-priority side stream created after them lands on the SAME hardware
        // queue as the caller's stream (4 queues per process by default, handed out in creation order), its launches serialise
        // behind the data-gradient chain and the backward pass loses the whole overlap: 4.7 -> 5.2 ms at B = 32
        // (bench.py --gpus 1 --force-dist, profiles/r05_hw_queues_ab.txt).
        {
            int lo = 0, hi = 0;
            MMK_CHECK_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
            MMK_CHECK_HIP(hipStreamCreateWithPriority(&s.st, hipStreamNonBlocking, hi));
        }
        MMK_CHECK_HIP(hipEventCreateWithFlags(&s.fork, hipEventDisableTiming));
        MMK_CHECK_HIP(hipEventCreateWithFlags(&s.join, hipEventDisableTiming));
        s.dev = dev;
    }
    *out = &s;
    return MMK_OK;
}

// 1/keep as the kernels apply it (mmk_unet.hip: dropout_params)
float dropout_scale(float p)
{
    const unsigned thr = (unsigned)(p * 65536.0f + 0.5f);
    return thr ? 65536.0f / (float)(65536u - thr) : 1.0f;
}

}  // namespace

extern "C" size_t mmk_unet_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t cin)
{
    Plan p;
    if (!make_plan(p, B, H, W, cin, false)) {
        mmk::set_error("mmk_unet_workspace_bytes: unsupported network input (B=%d, %d x %d, %d channels)", B, H, W, cin);
        return 0;
    }
    return p.ws_bytes;
}

extern "C" size_t mmk_unet_scratch_bytes(int32_t B, int32_t H, int32_t W, int32_t cin)
{
    Plan p;
    if (!make_plan(p, B, H, W, cin, true)) {
        mmk::set_error("mmk_unet_scratch_bytes: unsupported network input (B=%d, %d x %d, %d channels)", B, H, W, cin);
        return 0;
    }
    return p.scratch_bytes;
}

extern "C" int mmk_unet_tensor(int32_t B, int32_t H, int32_t W, int32_t cin, int32_t id, size_t *offset, int32_t *h, int32_t *w,
                               int32_t *c)
{
    Plan p;
    MMK_REQUIRE(make_plan(p, B, H, W, cin, false), "mmk_unet_tensor: unsupported network input");
    MMK_REQUIRE(offset && h && w && c, "mmk_unet_tensor: NULL pointer");
    const Tens *t = nullptr;
    if (id >= 0 && id < 6) t = &p.a_enc[id];
    else if (id < 12) t = &p.d_enc[id - 6];
    else if (id < 18) t = &p.t[id - 12];
    else if (id < 18 + 25) {
        const int j = (id - 18) / 5, q = (id - 18) % 5;
        const Tens *arr[5] = {&p.u[j], &p.a1[j], &p.d1[j], &p.a2[j], &p.d2[j]};
        t = arr[q];
    }
    MMK_REQUIRE(t != nullptr, "mmk_unet_tensor: bad tensor id %d", id);
    *offset = t->off; *h = t->h; *w = t->w; *c = t->c;
    return MMK_OK;
}

extern "C" int mmk_unet_forward(const mmk_unet_desc *d, void *stream)
{
    MMK_REQUIRE(d != nullptr, "mmk_unet_forward: NULL descriptor");
    MMK_REQUIRE(d->x && d->params && d->workspace && d->mask, "mmk_unet_forward: NULL pointer");
    Plan p;
    MMK_REQUIRE(make_plan(p, d->B, d->H, d->W, d->cin, false), "mmk_unet_forward: unsupported network input (B=%d, %d x %d, %d channels)",
                d->B, d->H, d->W, d->cin);
    MMK_REQUIRE(d->workspace_bytes >= p.ws_bytes, "mmk_unet_forward: workspace too small (%zu < %zu bytes)", d->workspace_bytes,
                p.ws_bytes);
    MMK_REQUIRE(d->drop_p >= 0.f && d->drop_p < 1.f, "mmk_unet_forward: dropout probability out of range");
    for (int i = 0; i < 2 * NCONV; ++i) MMK_REQUIRE(d->params[i] != nullptr, "mmk_unet_forward: NULL parameter %d", i);
    void *ws = d->workspace;
    const float sl = d->leaky_slope;
    const int B = p.B;
    auto Wk = [&](int k) { return d->params[2 * k]; };
    auto Bk = [&](int k) { return d->params[2 * k + 1]; };
    unsigned ctr = d->seed * 64u;

    // weights of the 21 3x3 layers -> bf16 MFMA-fragment order, one launch
    {
        const float *Wp[21];
        void *Op[21];
        int32_t co[21], ci[21];
        for (int k = 1; k <= 21; ++k) {
            Wp[k - 1] = Wk(k); Op[k - 1] = at(ws, p.packs[k]); co[k - 1] = p.cout[k]; ci[k - 1] = p.cinn[k];
        }
        MMK_TRY(mmk_conv3x3_pack_weights_batch(21, Wp, co, ci, 0, Op, stream));
    }
    // ---- encoder
    MMK_TRY(mmk_conv_first(d->x, p.cin, Wk(0), Bk(0), d->pre, B, p.H, p.W, sl, at(ws, p.a_enc[0].off), stream));
    {
        ConvCall c;
        c.x1 = at(ws, p.a_enc[0].off); c.C1 = 8; c.wpack = at(ws, p.packs[1]); c.bias = Bk(1);
        c.y1 = at(ws, p.d_enc[0].off); c.O1 = 8; c.relu = 1; c.drop_p = d->drop_p; c.seed = ++ctr;
        MMK_TRY(conv(p, p.H, p.W, sl, c, stream));
    }
    for (int i = 1; i < 6; ++i) {
        const int h = p.rh[i - 1], w = p.rw[i - 1], ch = ENC_CH[i];
        ConvCall c;
        c.x1 = at(ws, p.t[i - 1].off); c.C1 = ENC_CH[i - 1]; c.wpack = at(ws, p.packs[2 * i]); c.bias = Bk(2 * i);
        c.y1 = at(ws, p.a_enc[i].off); c.O1 = ch; c.relu = 1;
        MMK_TRY(conv(p, h, w, sl, c, stream));
        ConvCall c2;
        c2.x1 = at(ws, p.a_enc[i].off); c2.C1 = ch; c2.wpack = at(ws, p.packs[2 * i + 1]); c2.bias = Bk(2 * i + 1);
        c2.y1 = at(ws, p.d_enc[i].off); c2.O1 = ch; c2.relu = 1; c2.drop_p = d->drop_p; c2.seed = ++ctr;
        // ReLU network: the pooling leaves arg-max codes for the backward pass (mmk_maxpool2_bwd_arg), which then never reads
        // the block's pre-pool output d_enc[i]; where the second conv pools in the same pass it does not even write it
        // (desc.keep_full_res: it does, and a pooling pass over it makes the codes -- diagnostics)
        const bool fuse = sl == 0.f && mmk_conv3x3_pool_fusable(ch, ch, B, h, w) != 0;
        const bool lean = fuse && !d->keep_full_res;
        if (fuse) c2.pool_y = at(ws, p.t[i].off);          // the second conv writes its 2x2 max-pool as well
        if (lean) { c2.pool_arg = at(ws, p.t_arg[i]); c2.y1 = nullptr; }
        MMK_TRY(conv(p, h, w, sl, c2, stream));
        if (sl == 0.f) {
            if (!lean) MMK_TRY(mmk_maxpool2_fwd_arg(at(ws, p.d_enc[i].off), B, h, w, ch, at(ws, p.t[i].off), at(ws, p.t_arg[i]), stream));
        } else {
            MMK_TRY(mmk_maxpool2_fwd(at(ws, p.d_enc[i].off), B, h, w, ch, at(ws, p.t[i].off), stream));
        }
    }
    // ---- decoder (each block applied twice with shared weights: icp_weight_policy.py:178,182)
    const Tens *cur = &p.t[5];
    for (int j = 0; j < 5; ++j) {
        const int cs = ENC_CH[4 - j], h = p.rh[4 - j], w = p.rw[4 - j];
        const int k0 = 12 + 2 * j, k1 = 13 + 2 * j;
        MMK_TRY(mmk_upsample_fwd(at(ws, cur->off), B, cur->h, cur->w, cur->c, h, w, at(ws, p.u[j].off), stream));
        ConvCall c;
        c.x1 = at(ws, p.u[j].off); c.C1 = 2 * cs; c.wpack = at(ws, p.packs[k0]); c.bias = Bk(k0);
        c.y1 = at(ws, p.a1[j].off); c.O1 = cs; c.relu = 1;
        MMK_TRY(conv(p, h, w, sl, c, stream));
        ConvCall c1;
        c1.x1 = at(ws, p.a1[j].off); c1.C1 = cs; c1.wpack = at(ws, p.packs[k1]); c1.bias = Bk(k1);
        c1.y1 = at(ws, p.d1[j].off); c1.O1 = cs; c1.relu = 1; c1.drop_p = d->drop_p; c1.seed = ++ctr;
        MMK_TRY(conv(p, h, w, sl, c1, stream));
        ConvCall c2;                                        // torch.cat([skip, d1]) without a copy
        c2.x1 = at(ws, p.t[4 - j].off); c2.C1 = cs; c2.x2 = at(ws, p.d1[j].off); c2.C2 = cs;
        c2.wpack = at(ws, p.packs[k0]); c2.bias = Bk(k0); c2.y1 = at(ws, p.a2[j].off); c2.O1 = cs; c2.relu = 1;
        MMK_TRY(conv(p, h, w, sl, c2, stream));
        ConvCall c3;
        c3.x1 = at(ws, p.a2[j].off); c3.C1 = cs; c3.wpack = at(ws, p.packs[k1]); c3.bias = Bk(k1);
        c3.y1 = at(ws, p.d2[j].off); c3.O1 = cs; c3.relu = 1; c3.drop_p = d->drop_p; c3.seed = ++ctr;
        MMK_TRY(conv(p, h, w, sl, c3, stream));
        cur = &p.d2[j];
    }
    // ---- final 1x1 + sigmoid (+ per-image amax normalisation)
    float *raw = d->norm ? static_cast<float *>(at(ws, p.mask_raw)) : d->mask;
    MMK_TRY(mmk_final_fwd(at(ws, cur->off), Wk(22), Bk(22), (int64_t)B * p.H * p.W, raw, stream));
    if (d->norm)
        MMK_TRY(mmk_mask_normalize(raw, B, (int64_t)p.H * p.W, static_cast<float *>(at(ws, p.norm_part)), d->mask,
                                   static_cast<float *>(at(ws, p.amax)), stream));
    return MMK_OK;
}

namespace {
int unet_backward_impl(const mmk_unet_desc *d, const float *gmask, float *const *grads, void *scratch, size_t scratch_bytes,
                       void *const *bucket_events, void *stream);
}

extern "C" int mmk_unet_backward(const mmk_unet_desc *d, const float *gmask, float *const *grads, void *scratch,
                                 size_t scratch_bytes, void *stream)
{
    return unet_backward_impl(d, gmask, grads, scratch, scratch_bytes, nullptr, stream);
}

// The gradients of a pass become final in three groups, in this order (the order of the weight-gradient reductions below):
// bucket 0 = decoder + final layer (parameters 24..45), bucket 1 = encoder blocks 3-5 (12..23), bucket 2 = encoder blocks 0-2
// (0..11).  A data-parallel caller hands over one event per bucket and all-reduces a bucket on another stream as soon as its
// event has fired, beside the rest of the backward pass (BASELINE.json configs[3]: "grad all-reduce overlapped ...").
extern "C" int32_t mmk_unet_grad_bucket(int32_t bucket, int32_t *first_param, int32_t *n_params)
{
    static const int32_t first[MMK_UNET_GRAD_BUCKETS] = {24, 12, 0}, count[MMK_UNET_GRAD_BUCKETS] = {22, 12, 12};
    MMK_REQUIRE(bucket >= 0 && bucket < MMK_UNET_GRAD_BUCKETS && first_param && n_params, "mmk_unet_grad_bucket: bad bucket %d", bucket);
    *first_param = first[bucket];
    *n_params = count[bucket];
    return MMK_OK;
}

extern "C" int mmk_unet_backward_buckets(const mmk_unet_desc *d, const float *gmask, float *const *grads, void *scratch,
                                         size_t scratch_bytes, void *const *bucket_events, void *stream)
{
    MMK_REQUIRE(bucket_events != nullptr, "mmk_unet_backward_buckets: NULL event array");
    for (int b = 0; b < MMK_UNET_GRAD_BUCKETS; ++b) MMK_REQUIRE(bucket_events[b] != nullptr, "mmk_unet_backward_buckets: NULL event %d", b);
    return unet_backward_impl(d, gmask, grads, scratch, scratch_bytes, bucket_events, stream);
}

namespace {
int unet_backward_impl(const mmk_unet_desc *d, const float *gmask, float *const *grads, void *scratch, size_t scratch_bytes,
                       void *const *bucket_events, void *stream)
{
    MMK_REQUIRE(d != nullptr, "mmk_unet_backward: NULL descriptor");
    MMK_REQUIRE(d->x && d->params && d->workspace && d->mask && gmask && grads && scratch, "mmk_unet_backward: NULL pointer");
    Plan p;
    MMK_REQUIRE(make_plan(p, d->B, d->H, d->W, d->cin, true), "mmk_unet_backward: unsupported network input");
    MMK_REQUIRE(d->workspace_bytes >= p.ws_bytes, "mmk_unet_backward: workspace too small");
    MMK_REQUIRE(scratch_bytes >= p.scratch_bytes, "mmk_unet_backward: scratch too small (%zu < %zu bytes)", scratch_bytes, p.scratch_bytes);
    for (int i = 0; i < 2 * NCONV; ++i) MMK_REQUIRE(d->params[i] && grads[i], "mmk_unet_backward: NULL parameter / gradient %d", i);
    void *ws = d->workspace, *sc = scratch;
    hipStream_t st = (hipStream_t)stream;
    const float sl = d->leaky_slope, s = dropout_scale(d->drop_p);
    const int B = p.B;
    auto Wk = [&](int k) { return d->params[2 * k]; };

    {
        const float *Wp[21];
        void *Op[21];
        int32_t co[21], ci[21];
        for (int k = 1; k <= 21; ++k) {
            Wp[k - 1] = Wk(k); Op[k - 1] = at(sc, p.packs_t[k]); co[k - 1] = p.cout[k]; ci[k - 1] = p.cinn[k];
        }
        MMK_TRY(mmk_conv3x3_pack_weights_batch(21, Wp, co, ci, 1, Op, stream));
    }
    SideStream *ss = nullptr;
    if (use_side_stream()) MMK_TRY(side_stream(&ss));
    void *wstream = ss ? (void *)ss->st : stream;          // where the weight-gradient launches go
    if (ss) {       // everything enqueued so far (weight packing, zero fills) precedes the side stream's work
        MMK_CHECK_HIP(hipEventRecord(ss->fork, st));
        MMK_CHECK_HIP(hipStreamWaitEvent(ss->st, ss->fork, 0));
    }
    int part_used[NCONV] = {};      // sets of partial slices layer k has written so far (decoder layers: one per application)
    auto part_ptr = [&](int k) {
        float *ptr = static_cast<float *>(at(sc, p.part[k])) +
                     (size_t)part_used[k] * p.slices[k] * ((size_t)9 * p.cout[k] * p.cinn[k] + p.cout[k]);
        part_used[k] += 1;
        return ptr;
    };
    // parameter gradients of layers k0..k1: sum the slices / transpose, one launch on the weight-gradient stream, behind the
    // weight-gradient launches enqueued so far
    auto unpack = [&](int k0, int k1) -> int {
        if (ss) {   // (some slices were written on the caller's stream: the fused 8 -> 8 backward launches)
            MMK_CHECK_HIP(hipEventRecord(ss->fork, st));
            MMK_CHECK_HIP(hipStreamWaitEvent(ss->st, ss->fork, 0));
        }
        const float *src[21];
        int32_t slices[21], co[21], ci[21];
        float *dW[21], *db[21];
        int n = 0;
        for (int k = k0; k <= k1; ++k, ++n) {
            src[n] = static_cast<const float *>(at(sc, p.part[k]));
            if (part_used[k] != (k >= 12 ? 2 : 1)) {
                mmk::set_error("mmk_unet_backward: layer %d wrote %d sets of partial slices", k, part_used[k]);
                return MMK_ERR_ARG;
            }
            slices[n] = p.slices[k] * part_used[k]; co[n] = p.cout[k]; ci[n] = p.cinn[k];
            dW[n] = grads[2 * k]; db[n] = grads[2 * k + 1];
        }
        return mmk_conv3x3_wgrad_unpack_batch(n, src, slices, co, ci, dW, db, wstream);
    };

    // 8 -> 8 and 16 -> 16 second convolutions (ReLU network): data gradient and partial weight gradient in one launch on the
    // caller's stream -- both read the block's activation and the output gradient
    auto can_fuse = [&](int k, int ch, int h, int w) {
        return (ch == 8 || ch == 16) && sl == 0.f && p.slices[k] > 0 && mmk_conv3x3_wgrad_slices(ch, ch, ch, B, h, w) == p.slices[k];
    };
    auto bwd_fused = [&](int k, int ch, const void *x, const void *g, void *dx, int h, int w) -> int {
        return mmk_conv_bwd_fused(x, g, at(sc, p.packs_t[k]), 1.f, B, h, w, ch, dx, part_ptr(k), 0, stream);
    };
    auto wgrad = [&](int k, const void *x1, int C1, const void *x2, int C2, const void *g, int h, int w) -> int {
        if (ss) {   // g was produced by the launch just enqueued on the caller's stream
            MMK_CHECK_HIP(hipEventRecord(ss->fork, st));
            MMK_CHECK_HIP(hipStreamWaitEvent(ss->st, ss->fork, 0));
        }
        // (the plan sized the slices for an unsplit input; a split input must give the same count)
        if (mmk_conv3x3_wgrad_slices(p.cout[k], p.cinn[k], C1, B, h, w) != p.slices[k]) {
            mmk::set_error("mmk_unet_backward: partial-slice count of layer %d depends on the input split", k);
            return MMK_ERR_ARG;
        }
        return mmk_conv3x3_wgrad_partial(x1, x2, C1, C2, g, p.cout[k], B, h, w, part_ptr(k), 0, wstream);
    };

    // ---- final layer
    const Tens &d2_4 = p.d2[4];
    if (d->norm)
        MMK_TRY(mmk_final_bwd_normalized(at(ws, d2_4.off), Wk(22), static_cast<const float *>(at(ws, p.mask_raw)), d->mask,
                                         static_cast<const float *>(at(ws, p.amax)), gmask, B, (int64_t)p.H * p.W, s, sl,
                                         static_cast<float *>(at(sc, p.fin_ws)), static_cast<float *>(at(sc, p.fin_ws)) + (size_t)B * 128,
                                         at(sc, p.gz_final.off), grads[44], grads[45], static_cast<float *>(at(sc, p.fin_red)), stream));
    else
        MMK_TRY(mmk_final_bwd(at(ws, d2_4.off), Wk(22), d->mask, gmask, (int64_t)B * p.H * p.W, s, sl, at(sc, p.gz_final.off), grads[44],
                              grads[45], static_cast<float *>(at(sc, p.fin_red)), stream));
    // ---- decoder, j = 4..0
    const void *gz = at(sc, p.gz_final.off);
    for (int j = 4; j >= 0; --j) {
        const int cs = ENC_CH[4 - j], h = p.rh[4 - j], w = p.rw[4 - j];
        const int k0 = 12 + 2 * j, k1 = 13 + 2 * j;
        const void *skip = at(ws, p.t[4 - j].off);
        // second application
        const bool fuse8 = can_fuse(k1, cs, h, w);
        if (fuse8) {
            MMK_TRY(bwd_fused(k1, cs, at(ws, p.a2[j].off), gz, at(sc, p.gz_a2[j].off), h, w));
        } else {
            MMK_TRY(wgrad(k1, at(ws, p.a2[j].off), cs, nullptr, 0, gz, h, w));
            ConvCall c;
            c.x1 = gz; c.C1 = cs; c.wpack = at(sc, p.packs_t[k1]); c.y1 = at(sc, p.gz_a2[j].off); c.O1 = cs;
            c.src1 = at(ws, p.a2[j].off); c.scale1 = 1.f;
            MMK_TRY(conv(p, h, w, sl, c, stream));
        }
        if (j == 4 && sl == 0.f && cs == 8 && p.slices[k0] > 0 && mmk_conv3x3_wgrad_slices(8, 16, 8, B, h, w) == p.slices[k0]) {
            // last decoder block: both halves of the data gradient are masked by the two halves of the weight gradient's input
            MMK_TRY(mmk_conv16x8_bwd_fused(skip, at(ws, p.d1[j].off), at(sc, p.gz_a2[j].off), at(sc, p.packs_t[k0]), s, B, h, w,
                                           at(sc, p.gsk[j].off), at(sc, p.gz_d1[j].off), part_ptr(k0), 0, stream));
        } else {
            MMK_TRY(wgrad(k0, skip, cs, at(ws, p.d1[j].off), cs, at(sc, p.gz_a2[j].off), h, w));
            ConvCall c2;       // one pass, two outputs: the skip's gradient and the first application's output gradient
            c2.x1 = at(sc, p.gz_a2[j].off); c2.C1 = cs; c2.wpack = at(sc, p.packs_t[k0]);
            c2.y1 = at(sc, p.gsk[j].off); c2.O1 = cs; c2.scale1 = s;
            c2.src1 = (j == 4) ? skip : nullptr;   // dec4's skip is the post-dropout activation of encoder block 0
            c2.y2 = at(sc, p.gz_d1[j].off); c2.O2 = cs; c2.src2 = at(ws, p.d1[j].off); c2.scale2 = s;
            MMK_TRY(conv(p, h, w, sl, c2, stream));
        }
        // first application
        if (fuse8) {
            MMK_TRY(bwd_fused(k1, cs, at(ws, p.a1[j].off), at(sc, p.gz_d1[j].off), at(sc, p.gz_a1[j].off), h, w));
        } else {
            MMK_TRY(wgrad(k1, at(ws, p.a1[j].off), cs, nullptr, 0, at(sc, p.gz_d1[j].off), h, w));
            ConvCall c3;
            c3.x1 = at(sc, p.gz_d1[j].off); c3.C1 = cs; c3.wpack = at(sc, p.packs_t[k1]); c3.y1 = at(sc, p.gz_a1[j].off); c3.O1 = cs;
            c3.src1 = at(ws, p.a1[j].off); c3.scale1 = 1.f;
            MMK_TRY(conv(p, h, w, sl, c3, stream));
        }
        if (j == 4 && sl == 0.f && cs == 8 && p.slices[k0] > 0 && mmk_conv3x3_wgrad_slices(8, 16, 16, B, h, w) == p.slices[k0]) {
            // (first application: the input is the up-sampled tensor, no ReLU source -- the launch shares the output gradient)
            MMK_TRY(mmk_conv16x8_bwd_fused(at(ws, p.u[j].off), nullptr, at(sc, p.gz_a1[j].off), at(sc, p.packs_t[k0]), 1.f, B, h, w,
                                           at(sc, p.g_u[j].off), nullptr, part_ptr(k0), 0, stream));
        } else {
            MMK_TRY(wgrad(k0, at(ws, p.u[j].off), 2 * cs, nullptr, 0, at(sc, p.gz_a1[j].off), h, w));
            ConvCall c4;
            c4.x1 = at(sc, p.gz_a1[j].off); c4.C1 = cs; c4.wpack = at(sc, p.packs_t[k0]); c4.y1 = at(sc, p.g_u[j].off); c4.O1 = 2 * cs;
            MMK_TRY(conv(p, h, w, sl, c4, stream));
        }
        if (j > 0) {
            const Tens &pd2 = p.d2[j - 1];
            MMK_TRY(mmk_upsample_bwd(at(sc, p.g_u[j].off), B, pd2.h, pd2.w, 2 * cs, h, w, at(ws, pd2.off), s, sl, at(sc, p.gz_up[j].off),
                                     stream));
        } else {
            MMK_TRY(mmk_upsample_bwd(at(sc, p.g_u[j].off), B, p.rh[5], p.rw[5], 2 * cs, h, w, nullptr, 1.f, sl, at(sc, p.gz_up[j].off),
                                     stream));
        }
        gz = at(sc, p.gz_up[j].off);
    }
    // the decoder's weight gradients are all enqueued: their slices are reduced under the >= 64-channel encoder levels, which
    // leave the HBM idle
    MMK_TRY(unpack(12, 21));
    // (the final layer's gradients were written on the caller's stream before the fork the reduction waited for)
    if (bucket_events) MMK_CHECK_HIP(hipEventRecord((hipEvent_t)bucket_events[0], (hipStream_t)wstream));
    // ---- encoder, i = 5..1 (g_t = gradient w.r.t. t[i])
    const void *g_t = gz;
    for (int i = 5; i >= 1; --i) {
        const int ch = ENC_CH[i], h = p.rh[i - 1], w = p.rw[i - 1];
        if (sl == 0.f)
            MMK_TRY(mmk_maxpool2_bwd_arg(at(ws, p.t_arg[i]), g_t, B, h, w, ch, s, at(sc, p.gz_d[i].off), stream));
        else
            MMK_TRY(mmk_maxpool2_bwd(at(ws, p.d_enc[i].off), g_t, B, h, w, ch, s, sl, at(sc, p.gz_d[i].off), stream));
        if (can_fuse(2 * i + 1, ch, h, w)) {
            MMK_TRY(bwd_fused(2 * i + 1, ch, at(ws, p.a_enc[i].off), at(sc, p.gz_d[i].off), at(sc, p.gz_a[i].off), h, w));
        } else {
            MMK_TRY(wgrad(2 * i + 1, at(ws, p.a_enc[i].off), ch, nullptr, 0, at(sc, p.gz_d[i].off), h, w));
            ConvCall c;
            c.x1 = at(sc, p.gz_d[i].off); c.C1 = ch; c.wpack = at(sc, p.packs_t[2 * i + 1]); c.y1 = at(sc, p.gz_a[i].off); c.O1 = ch;
            c.src1 = at(ws, p.a_enc[i].off); c.scale1 = 1.f;
            MMK_TRY(conv(p, h, w, sl, c, stream));
        }
        // data gradient accumulates into the skip gradient the decoder wrote for t[i-1]
        void *tgt = at(sc, p.gsk[5 - i].off);             // g_skip[i-1] was written by decoder block j = 4 - (i-1)
        if (i == 1 && sl == 0.f && ch == 16 && ENC_CH[0] == 8 && p.slices[2] > 0 &&
            mmk_conv3x3_wgrad_slices(16, 8, 8, B, h, w) == p.slices[2]) {
            // first convolution of block 1: weight gradient and (ReLU-masked, accumulating) data gradient in one launch
            MMK_TRY(mmk_conv8x16_bwd_fused(at(ws, p.t[0].off), at(sc, p.gz_a[1].off), at(sc, p.packs_t[2]), s, B, h, w, tgt,
                                           part_ptr(2), 0, stream));
        } else {
            MMK_TRY(wgrad(2 * i, at(ws, p.t[i - 1].off), ENC_CH[i - 1], nullptr, 0, at(sc, p.gz_a[i].off), h, w));
            ConvCall c2;
            c2.x1 = at(sc, p.gz_a[i].off); c2.C1 = ch; c2.wpack = at(sc, p.packs_t[2 * i]); c2.y1 = tgt; c2.O1 = ENC_CH[i - 1];
            c2.acc1 = 1;
            if (i == 1) { c2.src1 = at(ws, p.t[0].off); c2.scale1 = s; }   // t[0] is an activation: factor, then accumulate
            MMK_TRY(conv(p, h, w, sl, c2, stream));
        }
        g_t = tgt;
        // the >= 64-channel encoder layers have their weight gradients enqueued: reduce their slices now (the reduction reads
        // every partial slice; as one launch at the end it is the tail of the backward pass, and beside the 640 x 640 kernels
        // it competes for their HBM bandwidth)
        if (i == 3) {
            MMK_TRY(unpack(6, 11));
            if (bucket_events) MMK_CHECK_HIP(hipEventRecord((hipEvent_t)bucket_events[1], (hipStream_t)wstream));
        }
    }
    // ---- encoder block 0
    if (can_fuse(1, 8, p.H, p.W)) {
        MMK_TRY(bwd_fused(1, 8, at(ws, p.a_enc[0].off), g_t, at(sc, p.gz_a[0].off), p.H, p.W));
    } else {
        MMK_TRY(wgrad(1, at(ws, p.a_enc[0].off), 8, nullptr, 0, g_t, p.H, p.W));
        ConvCall c;
        c.x1 = g_t; c.C1 = 8; c.wpack = at(sc, p.packs_t[1]); c.y1 = at(sc, p.gz_a[0].off); c.O1 = 8;
        c.src1 = at(ws, p.a_enc[0].off); c.scale1 = 1.f;
        MMK_TRY(conv(p, p.H, p.W, sl, c, stream));
    }
    // ---- parameter gradients of the first five 3x3 layers (the rest was reduced on the way): on the weight-gradient stream,
    // beside the first layer's weight gradient
    MMK_TRY(unpack(1, 5));
    MMK_TRY(mmk_conv_first_wgrad(d->x, p.cin, at(sc, p.gz_a[0].off), d->pre, B, p.H, p.W, grads[0], grads[1],
                                 static_cast<float *>(at(sc, p.first_ws)), mmk_conv_first_wgrad_ws_bytes(p.cin), stream));
    if (ss) {       // join: the caller's stream continues only after every gradient is written
        MMK_CHECK_HIP(hipEventRecord(ss->join, ss->st));
        MMK_CHECK_HIP(hipStreamWaitEvent(st, ss->join, 0));
    }
    if (bucket_events) MMK_CHECK_HIP(hipEventRecord((hipEvent_t)bucket_events[2], st));
    return MMK_OK;
}
}  // namespace
