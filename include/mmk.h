/*
 * mmk.h — C ABI of libmmk_hip.so: the MI355X (gfx950) implementation of the
 * mm_masking learned-mask -> differentiable-ICP hot path.
 *
 * The reference (utiasASRL/mm_masking) is pure Python and has NO FFI of its own
 * (SURVEY.md §2.1, §8b): each entry point below therefore cites the reference
 * *Python* interface it replaces; the binding a maintainer adds is the ctypes
 * stub shown in INTEGRATION.md (shipped as mm_masking_amd/_lib.py).
 *
 * Conventions
 *  - Every pointer is a DEVICE pointer owned by the caller unless marked host.
 *    Kernels never allocate or free memory; scratch comes in through
 *    (workspace, workspace_bytes) whose size the *_workspace_bytes() helpers give.
 *  - All work is enqueued on `stream` (a hipStream_t passed as void*) and returns
 *    without synchronising (exception: mmk_icp_forward with check_every > 0).
 *  - Return value: 0 on success, negative MMK_ERR_* otherwise; a message for the
 *    calling thread is available from mmk_last_error().  No exceptions cross
 *    the ABI, no global mutable state: safe for one-process-per-GPU use.
 *  - Tensors are dense row-major fp32 unless stated; B = scan pairs in the batch.
 */
#ifndef MMK_H_
#define MMK_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMK_VERSION 500 /* 0.5.0: mmk_unet_backward_buckets / mmk_unet_grad_bucket (per-bucket completion events for an overlapped gradient all-reduce); 0.4.2: arg-max codes of the poolings (mmk_conv_desc.pool_arg, mmk_maxpool2_fwd_arg / _bwd_arg, mmk_unet_desc.keep_full_res); 0.4.1: mmk_pose_loss_*, mmk_bce_mean_*; 0.4.0: mmk_icp_status / _accumulate / _solve_update; 0.3.1: mmk_host_read_rows_batch; 0.3.0: no float atomics left (first / final layer gradients and the mask-gradient scatter take workspaces; mmk_conv3x3_wgrad + _unpack removed) */

#define MMK_OK 0
#define MMK_ERR_ARG (-1)
#define MMK_ERR_HIP (-2)
#define MMK_ERR_WORKSPACE (-3)

#define MMK_ICP_PT2PT 0
#define MMK_ICP_PT2PL 1

#define MMK_NN_BRUTE 0 /* exhaustive scan (north_star's brute-force kernel)                  */
#define MMK_NN_GRID 1  /* exact search through a uniform grid: same indices, fewer evaluations */

#define MMK_LOSS_NONE 0
#define MMK_LOSS_CAUCHY 1
#define MMK_LOSS_HUBER 2

int mmk_version(void);
const char *mmk_last_error(void);

/* ------------------------------------------------------------------ dICP (external/dICP, absent upstream)
 * Replaces dICP.ICP.ICP(...).icp(source, target, T_init=, weight=, trim_dist=,
 * loss_fn=, dim=) as called at mm_masking/icp_weight_policy.py:281-287 (ctor
 * :54-55; target_pad_val use mm_masking/icp_weight_dataset.py:59-61,395).
 * Arithmetic: DESIGN.md §3 == oracle/dicp_ref.py.                                     */
typedef struct {
    int32_t B;          /* scan pairs                                            */
    int32_t N;          /* padded source points per pair (zero rows carry weight 0) */
    int32_t M;          /* padded target points per pair (target_pad_val rows)   */
    int32_t tgt_cols;   /* 3 (xyz) or 6 (xyz|normal): icp_weight_dataset.py:397-398 */
    int32_t dim;        /* 2 or 3 (reference passes dim=2)                        */
    int32_t icp_type;   /* MMK_ICP_PT2PT | MMK_ICP_PT2PL (ctor icp_type)          */
    int32_t loss;       /* MMK_LOSS_* from loss_fn["name"]                        */
    float loss_k;       /* loss_fn["metric"]                                      */
    float trim_dist;    /* trim_dist (5.0 at icp_weight_policy.py:279)            */
    float tolerance;    /* ctor tolerance: a pair freezes once ||delta|| < tol    */
    int32_t max_iter;   /* ctor max_iterations (K)                                */
    int32_t save_state; /* 1: keep per-iteration correspondences for backward     */
    int32_t check_every;/* >0: every that many iterations read the active flags
                           back (synchronises `stream`) and stop when all pairs
                           froze; 0: run max_iter iterations, never synchronise   */
    int32_t nn_method;  /* MMK_NN_BRUTE | MMK_NN_GRID: both return identical indices     */
} mmk_icp_params;

size_t mmk_icp_workspace_bytes(const mmk_icp_params *p);

/* State arrays (device): idx_hist int32 (K,B,N) if save_state else (B,N);
 * T_hist fp32 (K+1,B,16): poses before iteration k, T_hist[K] = result;
 * delta_hist fp64 (K,B,6); A_hist fp64 (K,B,36) row-major p x p in the leading
 * block; active_hist int32 (K+1,B): 1 while the pair still iterates.
 * weight may be NULL (all ones).  iters_run (host int*, may be NULL).               */
int mmk_icp_forward(const mmk_icp_params *p, const float *source /*B,N,3*/,
                    const float *target /*B,M,tgt_cols*/, const float *weight /*B,N*/,
                    const float *T_init /*B,16*/, float *T_out /*B,16*/,
                    int32_t *idx_hist, float *T_hist, double *delta_hist, double *A_hist,
                    int32_t *active_hist, void *workspace, size_t workspace_bytes,
                    int *iters_run, void *stream);

/* Reverse sweep over the saved state: grad_weight (B,N) = dL/dweight,
 * grad_T_init (B,16, may be NULL) = dL/dT_init, given grad_T (B,16) = dL/dT_out.
 * Replaces autograd through the unrolled dICP iterations
 * (train_icp_weights.py:51 loss.backward()).                                           */
int mmk_icp_backward(const mmk_icp_params *p, const float *source, const float *target,
                     const float *weight, const int32_t *idx_hist, const float *T_hist,
                     const double *delta_hist, const double *A_hist,
                     const int32_t *active_hist, const float *grad_T /*B,16*/,
                     float *grad_weight /*B,N*/, float *grad_T_init /*B,16 or NULL*/,
                     void *workspace, size_t workspace_bytes, void *stream);

/* Status of the last mmk_icp_forward call that used `workspace`: the MMK_ICP_STATUS_* bits its kernels raised (0 = clean).
 * Asynchronous like everything else: enqueues a 4-byte copy on `stream` into status_out (a device pointer or pinned host
 * memory); the value is valid once the stream has passed it.  MMK_ICP_STATUS_UNARMED_KEY: a source row reached the
 * accumulation stage with a nearest-neighbour key no search kernel wrote (or an index outside the target) -- an internal
 * error, never the result of the caller's data; the row was clamped to stay in bounds and the poses of that call are
 * not to be trusted.                                                                                                     */
#define MMK_ICP_STATUS_UNARMED_KEY 1
int mmk_icp_status(const mmk_icp_params *p, const void *workspace, size_t workspace_bytes,
                   int32_t *status_out, void *stream);

/* The stages of one iteration on their own (SURVEY.md 8b names them; mmk_icp_forward is these in a loop behind the NN search).
 * mmk_icp_accumulate (stages I3 + I4): consumes nearest-neighbour keys ((bits of d2) << 32 | index, (B,N) uint64 -- the
 * layout the search kernels produce), writes the correspondences idx_out (B,N) and per-workgroup partial sums of the normal
 * equations, `partials` fp64 with mmk_icp_partials_count(p) elements ((B, ceil(N/256), 9 | 27): upper triangle of A row-major,
 * then b); ORs MMK_ICP_STATUS_* bits into the device word *status (caller zeroes it).  T (B,16) = the pose the keys were
 * found under.  mmk_icp_solve_update (stages I5 + I6): ordered sum of the partials, Cholesky solve, T_out = Exp(delta) T_in,
 * delta_out (B,6), A_out (B,36), active_out[b] = 0 once ||delta|| < tolerance (active_in[b] = 0: pair frozen, pose copied). */
size_t mmk_icp_partials_count(const mmk_icp_params *p);
int mmk_icp_accumulate(const mmk_icp_params *p, const float *source, const float *target,
                       const float *weight /*B,N or NULL*/, const float *T /*B,16*/,
                       const uint64_t *nn_keys /*B,N*/, int32_t *idx_out /*B,N*/, double *partials,
                       int32_t *status /*device, 1*/, void *stream);
int mmk_icp_solve_update(const mmk_icp_params *p, const double *partials, const float *T_in /*B,16*/,
                         float *T_out /*B,16*/, double *delta_out /*B,6*/, double *A_out /*B,36*/,
                         const int32_t *active_in /*B*/, int32_t *active_out /*B*/, void *stream);

/* Stand-alone brute-force nearest neighbour (stage I2), the roofline kernel.
 * target_planar: (B,dim,Mpad) produced by mmk_pack_target, Mpad = mmk_nn_padded_m(M).
 * T (B,16) is applied to the source first (stage I1).  idx int32 (B,N), d2 fp32 (B,N). */
int32_t mmk_nn_padded_m(int32_t M);
int mmk_pack_target(const float *target /*B,M,tgt_cols*/, int32_t B, int32_t M,
                    int32_t tgt_cols, int32_t dim, float *target_planar, void *stream);
size_t mmk_nn_workspace_bytes(int32_t B, int32_t N, int32_t M, int32_t dim);
int mmk_nn_search(const float *source /*B,N,3*/, const float *target_planar,
                  const float *T /*B,16*/, int32_t B, int32_t N, int32_t M, int32_t dim,
                  int32_t *idx, float *d2, void *workspace, size_t workspace_bytes,
                  void *stream);

/* Measurement hook (bench.py): while enabled, every nn_search launch made by this
 * process (inside mmk_icp_forward or mmk_nn_search) is bracketed by HIP events on its
 * launch stream.  _end() waits for the recorded events and returns the per-launch
 * durations in milliseconds (ms_out is a HOST array; n_out = launches seen).  Process-wide,
 * not thread-safe: measurement only.  Nothing upstream corresponds to it (the reference
 * times whole epochs, train_icp_weights.py:518-523).                                    */
int mmk_nn_profile_begin(int32_t capacity);
int mmk_nn_profile_end(float *ms_out /*host*/, int32_t max_out, int32_t *n_out /*host*/);

/* ------------------------------------------------------------------ loss terms of train_icp_weights.py:179-253
 * One launch each (plus an ordered final sum) instead of ~45 PyTorch launches per step; deterministic (no float atomics).
 * mmk_pose_loss_fwd: out2[0] = mean_b |T[b,1,0]|  (loss_rot, :199), out2[1] = mean_b ||(T[b,0,3], T[b,1,3])||  (loss_trans, :200)
 *   of xi = T_pred - I (the gt_eye form, :193).  _bwd: grad_T (B,16) = g_rot * d rot / dT + g_trans * d trans / dT, with g_rot /
 *   g_trans DEVICE scalars (NULL = 0): sign(.) / B and (x, y) / norm / B, zero where the norm is zero (as torch.norm's backward).
 * mmk_bce_mean_fwd: out[0] = torch.nn.BCELoss()(x, target) over n elements (logs clamped at -100), ws = mmk_bce_ws_bytes()
 *   bytes of device workspace.  _bwd: grad_x = grad_out[0] * (x - t) / max((1 - x) x, 1e-12) / n, grad_out a device scalar.    */
int mmk_pose_loss_fwd(const float *T_pred /*B,16*/, int32_t B, float *out2, void *stream);
int mmk_pose_loss_bwd(const float *T_pred, int32_t B, const float *g_rot, const float *g_trans,
                      float *grad_T /*B,16*/, void *stream);
size_t mmk_bce_ws_bytes(void);
int mmk_bce_mean_fwd(const float *x, const float *target, int64_t n, void *ws, size_t ws_bytes,
                     float *out, void *stream);
int mmk_bce_mean_bwd(const float *x, const float *target, int64_t n, const float *grad_out,
                     float *grad_x, void *stream);

/* ------------------------------------------------------------------ radar_utils.py
 * mmk_cfar_mask        <- cfar_mask                      radar_utils.py:29-69
 * mmk_extract_peaks    <- extract_pc (+mean_peaks_parallel_fast, pol_2_cart)
 *                                                        radar_utils.py:71-106,167-195
 * mmk_polar_to_cart    <- radar_polar_to_cartesian_diff  radar_utils.py:258-336
 * mmk_cart_to_polar    <- radar_cartesian_to_polar       radar_utils.py:338-372
 * mmk_sample_weights_* <- extract_weights (fwd + autograd bwd) radar_utils.py:108-128
 * mmk_bev_raster       <- extract_bev_from_pts           radar_utils.py:142-165      */

/* GO-CFAR on (B,A,R) polar power.  w2/guard/mincol/maxcol as derived at
 * radar_utils.py:34-39 by the caller.  diff: 0 hard (x>thres), 1 soft (tanh+hardshrink). */
int mmk_cfar_mask(const float *raw, int32_t B, int32_t A, int32_t R, int32_t w2,
                  int32_t guard, int32_t mincol, int32_t maxcol, float a_thresh,
                  float b_thresh, int32_t diff, float steep_fact, float *mask,
                  void *stream);

size_t mmk_extract_peaks_workspace_bytes(int32_t B, int32_t A, int32_t R, int32_t max_pts);
/* Blob-centre extraction.  out_pc (B,max_pts,3) zero padded in the reference's
 * azimuth-major order; out_count int32 (B) = points the reference would return
 * (may exceed max_pts: then the cloud is truncated); T_ab (B,16) may be NULL.          */
int mmk_extract_peaks(const float *mask, int32_t B, int32_t A, int32_t R, float res,
                      const float *azimuths /*B,A*/, const float *times /*B,A*/,
                      const float *T_ab, int32_t diff, float steep_fact, int32_t max_pts,
                      float *out_pc, int32_t *out_count, void *workspace,
                      size_t workspace_bytes, void *stream);

/* Polar (B,A,R) -> Cartesian (B,W,W) bilinear resample.  range_grid/angle_grid (W,W)
 * are form_cart_range_angle_grid's outputs (radar_utils.py:399-419), built by the host. */
int mmk_polar_to_cart(const float *polar, const float *azimuths /*B,A*/,
                      const float *range_grid, const float *angle_grid, int32_t B,
                      int32_t A, int32_t R, int32_t W, float radar_resolution,
                      int32_t interpolate_crossover, int32_t fix_wobble, float *cart,
                      void *stream);
/* Two images of one batch resampled with the same azimuths in one pass (FFT and CFAR image of a
 * scan, icp_weight_dataset.py:350-352): coordinates and tap weights are computed once. */
int mmk_polar_to_cart_pair(const float *polar, const float *polar2, const float *azimuths,
                           const float *range_grid, const float *angle_grid, int32_t B, int32_t A,
                           int32_t R, int32_t W, float radar_resolution, int32_t interpolate_crossover,
                           int32_t fix_wobble, float *cart, float *cart2, void *stream);

/* Cartesian (B,H,W) -> polar (B,A,R) bilinear resample in fp64 <- radar_cartesian_to_polar, radar_utils.py:338-372
 * (the reference casts its sampling grid to double at :370: fp64 images only).  sin_az / cos_az (B,A) and
 * range_coords (R) = linspace(0, (R-1) radar_resolution, R) are formed by the host with torch's CPU functions, as the
 * reference does; the output is bit-identical to the reference's. */
int mmk_cart_to_polar(const double *cart, const double *sin_az, const double *cos_az, const double *range_coords,
                      int32_t B, int32_t A, int32_t R, int32_t H, int32_t W, double cart_resolution, double *polar,
                      void *stream);

/* weights[b,n] = bilinear(mask[b], point n) with zero padding; fake points
 * (x==0 && y==0) get 0.  cart_resolution / cart_pixel_width as point_to_cart_idx
 * (radar_utils.py:374-397; extract_weights leaves them at 0.2384 / 640): the points are
 * normalised by the Cartesian grid's width, then F.grid_sample maps [-1,1] onto the mask's own
 * (H,W) -- so a polar (400,3360) mask is sampled exactly as the reference samples it. */
int mmk_sample_weights_fwd(const float *mask /*B,H,W*/, const float *pc /*B,N,pc_cols*/,
                           int32_t B, int32_t N, int32_t pc_cols, int32_t H, int32_t W,
                           int32_t cart_pixel_width, float cart_resolution,
                           float *weights /*B,N*/, void *stream);
/* grad_mask (B,H,W) is zero-filled here, then receives the scatter-add -- in a FIXED order: the taps that fall on one
 * pixel are chained (integer exchanges), and the chain's owner adds them in ascending (point, tap) order, which is
 * the order of a sequential loop over the points (what PyTorch's CPU grid_sample backward does): no float atomics,
 * bit-reproducible.  ws: mmk_sample_weights_bwd_ws_bytes(B, N) bytes. */
size_t mmk_sample_weights_bwd_ws_bytes(int32_t B, int32_t N);
int mmk_sample_weights_bwd(const float *grad_weights /*B,N*/, const float *pc, int32_t B,
                           int32_t N, int32_t pc_cols, int32_t H, int32_t W,
                           int32_t cart_pixel_width, float cart_resolution, float *grad_mask,
                           void *ws, size_t ws_bytes, void *stream);

/* ---- loader primitives (icp_weight_dataset.py:323-362: PNG rows -> load_radar -> augmentation roll -> polar -> Cartesian)
 * mmk_host_read_rows: HOST function, no GPU call, thread-safe: rows [0,rows) of a raw row-major byte file (header_bytes
 * skipped, row_bytes per row; the decoded-scan cache / prepared-cloud cache of ICPWeightDataset), columns
 * [col0, col0+ncols) kept, rows rotated as torch.roll(x, roll, dims=0) (the augmentation's azimuth roll, :446-452),
 * written densely to dst (rows*ncols bytes; typically a slice of the batch's pinned buffer).
 * mmk_u8_to_float: out[i] = lut256[in[i]] on the device -- load_radar's bytes / 255 (radar_utils.py:26) and the CFAR
 * cache's (:343) with the table computed by the host's own fp32 division, so that the values are the reference's bit
 * for bit whatever the device's division rounds to. */
int mmk_host_read_rows(const char *path, int64_t header_bytes, int32_t rows, int32_t row_bytes, int32_t col0,
                       int32_t ncols, int32_t roll, void *dst);
/* The same for a list of jobs (typically every tensor of every item of a batch), spread over `threads` host threads
 * that live for the duration of the call: one interpreter thread drives the whole loader (the reference's four worker
 * processes, train_icp_weights.py:454-455, become four C threads).  Fails if any job fails (first message kept). */
typedef struct mmk_read_job {
    const char *path;
    int64_t header_bytes;
    int32_t rows, row_bytes, col0, ncols, roll, reserved;
    void *dst;
} mmk_read_job;
int mmk_host_read_rows_batch(const mmk_read_job *jobs, int32_t n_jobs, int32_t threads);
int mmk_u8_to_float(const void *in /*n bytes*/, const float *lut256, int64_t n, float *out, void *stream);

/* Statistics extract_weights returns next to the weights (radar_utils.py:130-138) and the policy's
 * mean_all_pts (icp_weight_policy.py:209-212), over the real points (not x==0 && y==0):
 * out[0] diff_mean_num_non0 = sum(0.5 tanh(5w)+0.5)/B, out[1] mean_num_non0 = count(w>0.05)/B,
 * out[2] mean_w, out[3] max_w, out[4] min_w, out[5] mean_all_pts = count(x!=0 && y!=0)/B,
 * out[6] number of real points.  partial: B*8 floats of workspace; out: 8 floats. */
int mmk_weight_stats(const float *weights /*B,N*/, const float *pc /*B,N,pc_cols*/, int32_t B, int32_t N,
                     int32_t pc_cols, float *partial, float *out, void *stream);

int mmk_bev_raster(const float *pc /*B,M,pc_cols*/, int32_t B, int32_t M, int32_t pc_cols,
                   int32_t W, float cart_resolution, float *bev /*B,W,W*/, void *stream);

/* ------------------------------------------------------------------ mask U-Net building blocks
 * Replace nn.Conv2d(3x3, pad 1) + ReLU [+ Dropout] and their autograd of the reference's
 * mask predictor (mm_masking/icp_weight_policy.py:84-99 topology, :104-125 conv_block,
 * :161-184 forward) for its default configuration (ReLU, no batch norm).  Activations are
 * NHWC bf16 (B,H,W,C), C in {8,16,32,64,128,256}; accumulation fp32 on the matrix cores.   */
typedef struct {
    const void *x1;        /* bf16 (B,H,W,C1)                                              */
    const void *x2;        /* bf16 (B,H,W,C2) or NULL: the input is concat(x1, x2) — the
                              torch.cat([skip, x]) of icp_weight_policy.py:180 without a copy */
    int32_t C1, C2;
    const void *wpack;     /* from mmk_conv3x3_pack_weights                                */
    const float *bias;     /* [O1+O2] fp32 or NULL                                          */
    void *y1;              /* bf16 (B,H,W,O1): output channels [0,O1)                       */
    const void *relu_src1; /* optional bf16 (B,H,W,O1): y1 = result * (relu_src1 > 0 ? scale1 : 0)
                              (ReLU / dropout backward fused into the data-gradient pass)   */
    int32_t O1, accumulate1; /* accumulate: y1 += result                                    */
    float scale1;
    void *y2;              /* bf16 (B,H,W,O2) or NULL: output channels [O1,O1+O2)           */
    const void *relu_src2;
    int32_t O2, accumulate2;
    float scale2;
    int32_t B, H, W;
    int32_t relu;          /* forward: ReLU after the bias                                  */
    float leaky_slope;     /* > 0: the network's LeakyReLU variant (params["leaky"], nn.LeakyReLU(0.1),
                              icp_weight_policy.py:106): forward max(v, slope v) instead of ReLU; relu_src
                              epilogues use (src > 0 ? scale : src < 0 or -0.0 ? slope*scale : 0) -- a kept
                              zero is stored as -0.0, a dropped element as +0.0.  0 = ReLU              */
    float drop_p;          /* forward: inverted dropout with this probability (0 = none)    */
    uint32_t seed;
    void *pool_y;          /* optional bf16 (B,H/2,W/2,O1): nn.MaxPool2d(2,2) of y1 written by the same
                              pass (forward role, single output; layers for which
                              mmk_conv3x3_pool_fusable() returns 1), else NULL                  */
    void *pool_arg;        /* optional, with pool_y: (B,H/2,W/2,O1/2) bytes of arg-max codes, one nibble per channel
                              (channel c in bits 4 (c & 1) of byte c / 2) = position 0..3 of the window's first maximum in
                              scan order | (maximum > 0) << 2.  When set, y1 is NOT written (may be NULL): the pooled
                              tensor and the codes are all that the rest of the network and its backward pass
                              (mmk_maxpool2_bwd_arg) read of this layer's output                  */
} mmk_conv_desc;

/* 1 when mmk_conv3x3 can write the 2x2 max-pool of this layer's output itself (pool_y). */
int32_t mmk_conv3x3_pool_fusable(int32_t cin, int32_t cout, int32_t B, int32_t H, int32_t W);

/* Packed bf16 element count / packing of fp32 master weights W[cout][cin][3][3].
 * transposed = 1 packs the data-gradient operator (cout -> cin, taps flipped).            */
size_t mmk_conv3x3_packed_elems(int32_t cout, int32_t cin, int32_t transposed);
int mmk_conv3x3_pack_weights(const float *W, int32_t cout, int32_t cin, int32_t transposed,
                             void *packed, void *stream);
/* n layers in one launch: W[i] fp32 (cout[i],cin[i],3,3) -> packed[i] (mmk_conv3x3_packed_elems elements
 * each); replaces the per-layer .to(bf16) casts autocast performs on every nn.Conv2d call
 * (icp_weight_policy.py:104-125 run under torch.autocast). */
int mmk_conv3x3_pack_weights_batch(int32_t n, const float *const *W, const int32_t *cout, const int32_t *cin,
                                   int32_t transposed, void *const *packed, void *stream);
int mmk_conv3x3(const mmk_conv_desc *d, void *stream);
/* Weight + bias gradient of the same convolution (autograd of nn.Conv2d, train_icp_weights.py:51), g = gradient
 * w.r.t. the pre-activation, as per-workgroup partial sums (mmk_conv3x3_wgrad_slices() = number of slices for a shape
 * on the current device, 0 = unsupported shape): every workgroup stores its own slice of `partials`
 * (slices, 9*cout*cin + cout) -- the (9,cout,cin) weight sums followed by the cout bias sums -- with plain
 * stores (accumulate != 0: adds to it -- second application of shared weights) and
 * mmk_conv3x3_wgrad_unpack_batch sums the slices: no float atomics, bit-reproducible. */
int32_t mmk_conv3x3_wgrad_slices(int32_t cout, int32_t cin, int32_t c1, int32_t B, int32_t H, int32_t W);
int mmk_conv3x3_wgrad_partial(const void *x1, const void *x2, int32_t C1, int32_t C2, const void *g, int32_t cout,
                              int32_t B, int32_t H, int32_t W, float *partials, int32_t accumulate, void *stream);

/* Backward pass of a C -> C convolution (C = 8 or 16) whose ReLU source IS its input activation (the second convolution of
 * a U-Net block: icp_weight_policy.py:115-121) in one launch: dx = ((x > 0) ? scale : 0) * conv_T(g) as mmk_conv3x3 gives it
 * with relu_src = x, and the partial slices of mmk_conv3x3_wgrad_partial(x, NULL, C, 0, g, C, ...) -- bit-identical to those
 * two calls, reading x and g once instead of twice.  wpack_t = mmk_conv3x3_pack_weights(W, C, C, transposed = 1);
 * partials holds mmk_conv3x3_wgrad_slices(C, C, C, B, H, W) slices of 9*C*C + C floats. */
int mmk_conv_bwd_fused(const void *x, const void *g, const void *wpack_t, float scale, int32_t B, int32_t H, int32_t W,
                       int32_t C, void *dx, float *partials, int32_t accumulate, void *stream);

/* The same for the backward of an 8 -> 16 convolution whose data gradient (16 -> 8) takes the layer's input activation x as
 * ReLU source and ADDS its result to what dx holds (first convolution of encoder block 1: the skip gradient is there
 * already): dx += ((x > 0) ? scale : 0) * conv_T(g), as mmk_conv3x3 with relu_src = x, accumulate = 1, and the partial
 * slices of mmk_conv3x3_wgrad_partial(x, NULL, 8, 0, g, 16, ...): mmk_conv3x3_wgrad_slices(16, 8, 8, B, H, W) slices of
 * 9*16*8 + 16 floats.  x: (B,H,W,8), g: (B,H,W,16), wpack_t = mmk_conv3x3_pack_weights(W, 16, 8, transposed = 1). */
int mmk_conv8x16_bwd_fused(const void *x, const void *g, const void *wpack_t, float scale, int32_t B, int32_t H, int32_t W,
                           void *dx, float *partials, int32_t accumulate, void *stream);

/* ... and for the backward of a 16 -> 8 convolution on concat(x1, x2) (8 channels each) whose data gradient (8 -> 16) is
 * split the same way and masked by the two inputs: dx1 = ((x1 > 0) ? scale : 0) * conv_T(g)[0:8], dx2 likewise with x2 and
 * channels 8:16 (second application of the last decoder block's first convolution), as mmk_conv3x3 with two output parts,
 * relu_src = x1 / x2, and the partial slices of mmk_conv3x3_wgrad_partial(x1, x2, 8, 8, g, 8, ...):
 * mmk_conv3x3_wgrad_slices(8, 16, 8, B, H, W) slices of 9*8*16 + 8 floats.  g: (B,H,W,8).
 * x2 = dx2 = NULL: x1 is ONE (B,H,W,16) input and dx1 ONE (B,H,W,16) output without ReLU source (scale unused):
 * dx1 = conv_T(g), as mmk_conv3x3 without epilogue operands + mmk_conv3x3_wgrad_partial(x1, NULL, 16, 0, g, 8, ...). */
int mmk_conv16x8_bwd_fused(const void *x1, const void *x2, const void *g, const void *wpack_t, float scale, int32_t B, int32_t H,
                           int32_t W, void *dx1, void *dx2, float *partials, int32_t accumulate, void *stream);
/* n layers in one launch (no accumulation): dW[i] (cout,cin,3,3) = src[i] (9,cout,cin) when slices is NULL or
 * slices[i] == 0; else the sum over the slices[i] partial slices of src[i], and db[i] (cout, optional) their
 * bias sums */
int mmk_conv3x3_wgrad_unpack_batch(int32_t n, const float *const *src, const int32_t *slices, const int32_t *cout,
                                   const int32_t *cin, float *const *dW, float *const *db, void *stream);

/* First conv of the network (encoder.0.0): fp32 NCHW input (B,cin,H,W), cin = 1..4
 * (fft | cfar | range channels, icp_weight_policy.py:84), W[8][cin][3][3], + bias + ReLU ->
 * bf16 (B,H,W,8).  _wgrad: dW[8][cin][3][3] += , db[8] += (g = grad w.r.t. the pre-activation).
 * pre (may be NULL): 2 floats per input channel (offset, reciprocal scale); the kernels then read
 * (x - offset) * rscale — the policy's per-channel min-max normalisation
 * (icp_weight_policy.py:151-155) applied while loading.  mmk_channel_minmax fills pre with
 * (min, 1 / (max - min)) over (B,H,W) per channel; part: C*2048 floats of workspace; minmax (optional):
 * the raw (min, max) pairs, what a data-parallel job reduces over its ranks to keep the normalisation
 * batch-global (mm_masking_amd/icp_weight_policy.py: params["global_minmax"]). */
int mmk_channel_minmax(const float *x /*B,C,hw*/, int32_t B, int32_t C, int64_t hw, float *part,
                       float *pre /*C*2*/, float *minmax /*C*2 or NULL*/, void *stream);
int mmk_conv_first(const float *x, int32_t cin, const float *W, const float *bias, const float *pre,
                   int32_t B, int32_t H, int32_t Wd, float leaky_slope, void *y, void *stream);
/* dW[8][cin][3][3] and db[8] are WRITTEN (not added to): per-block partial sums into ws, then one ordered reduction --
 * no float atomics, bit-reproducible.  ws: mmk_conv_first_wgrad_ws_bytes(cin) bytes. */
size_t mmk_conv_first_wgrad_ws_bytes(int32_t cin);
int mmk_conv_first_wgrad(const float *x, int32_t cin, const void *g, const float *pre, int32_t B,
                         int32_t H, int32_t Wd, float *dW, float *db, float *ws, size_t ws_bytes, void *stream);

/* nn.MaxPool2d(2,2) on NHWC bf16 (icp_weight_policy.py:122-123).  _bwd fuses the backward of
 * the preceding Dropout(ReLU(.)): gz = route(gy) * (d > 0 ? scale : 0), d = the pooled tensor's
 * source (B,H,W,C).                                                                          */
int mmk_maxpool2_fwd(const void *x, int32_t B, int32_t H, int32_t W, int32_t C, void *y, void *stream);
int mmk_maxpool2_bwd(const void *d, const void *gy, int32_t B, int32_t H, int32_t W, int32_t C, float scale,
                     float leaky_slope, void *gz, void *stream);
/* The same pair through arg-max codes (layout: mmk_conv_desc.pool_arg): _fwd_arg also writes the codes, _bwd_arg routes
 * gy by them instead of re-deriving the arg-max from the full-resolution tensor d (ReLU network: leaky_slope = 0) --
 * gz is bit-identical to mmk_maxpool2_bwd's, from C/2 + 2C bytes per window instead of 10C.                          */
int mmk_maxpool2_fwd_arg(const void *x, int32_t B, int32_t H, int32_t W, int32_t C, void *y, void *arg, void *stream);
int mmk_maxpool2_bwd_arg(const void *arg, const void *gy, int32_t B, int32_t H, int32_t W, int32_t C, float scale,
                         void *gz, void *stream);

/* nn.UpsamplingBilinear2d(size) = bilinear, align_corners=True (icp_weight_policy.py:175-176).
 * _bwd is the adjoint in gather form; relu_src (optional, source-sized) applies
 * (relu_src > 0 ? scale : 0).                                                                */
int mmk_upsample_fwd(const void *x, int32_t B, int32_t Hs, int32_t Ws, int32_t C, int32_t Ho, int32_t Wo, void *y,
                     void *stream);
int mmk_upsample_bwd(const void *gy, int32_t B, int32_t Hs, int32_t Ws, int32_t C, int32_t Ho, int32_t Wo,
                     const void *relu_src, float scale, float leaky_slope, void *gx, void *stream);

/* final_layer: Conv2d(8,1,1x1) + Sigmoid (icp_weight_policy.py:96-99,184): bf16 (npix,8) -> fp32
 * mask (npix).  _bwd: gx = dL/dx * (x > 0 ? scale : 0) (bf16); dW[8] and db[1] are WRITTEN: per-block partial
 * sums into ws (MMK_FINAL_BWD_WS_FLOATS floats), then one ordered reduction -- no float atomics.  */
#define MMK_FINAL_BWD_WS_FLOATS 16384
int mmk_final_fwd(const void *x, const float *w, const float *bias, int64_t npix, float *mask, void *stream);
int mmk_final_bwd(const void *x, const float *w, const float *mask, const float *gmask, int64_t npix, float scale,
                  float leaky_slope, void *gx, float *dW, float *db, float *ws, void *stream);
/* mask_n = mask / amax(mask over the image) per image (icp_weight_policy.py:192-193), amax (B) out;
 * part: B*64 floats of workspace. */
int mmk_mask_normalize(const float *mask /*B,npix_per*/, int32_t B, int64_t npix_per, float *part,
                       float *mask_n, float *amax, void *stream);
/* mmk_final_bwd for a mask that went through mmk_mask_normalize: gmask_n is the gradient w.r.t.
 * mask_n; the adjoint of the division and of amax (spread evenly over tied maxima, as torch.amax)
 * is applied on the fly.  part: B*128 floats, coef: 2*B floats, ws: MMK_FINAL_BWD_WS_FLOATS floats of workspace. */
int mmk_final_bwd_normalized(const void *x, const float *w, const float *mask, const float *mask_n,
                             const float *amax, const float *gmask_n, int32_t B, int64_t npix_per,
                             float scale, float leaky_slope, float *part, float *coef, void *gx, float *dW,
                             float *db, float *ws, void *stream);

/* nn.BatchNorm2d of the network's batch-norm variant (params["batch_norm"], icp_weight_policy.py:108-113) on NHWC
 * bf16 (npix, C).  _forward_stats: batch statistics -> stat (C,2) = (mean, 1/sqrt(var + eps)) and affine (C,2) =
 * (scale, shift); running_mean / running_var (both or neither) updated as torch does (momentum, unbiased variance);
 * part: 512*C*2 floats of workspace.  _apply: y = a * scale + shift, then the block's inverted dropout (drop_p, seed;
 * 0 = none); a kept zero is stored as -0.0, a dropped value as +0.0.  _backward: gd = dL/d(output), d = the stored
 * post-dropout output when a dropout followed (else NULL; gy = gd * (d != +0 ? drop_scale : 0)), a = the BatchNorm
 * input (= ReLU / LeakyReLU output); writes gz = dL/d(pre-activation of the convolution in front of the ReLU),
 * dgamma / dbeta (accumulate != 0: adds -- decoder blocks run twice with shared modules); stat == NULL selects the
 * evaluation-mode form gz = gy * scale * act'(a).  coef: 3*C floats of workspace. */
int mmk_bn_forward_stats(const void *a, int64_t npix, int32_t C, const float *gamma, const float *beta, float eps,
                         float momentum, float *running_mean, float *running_var, float *part, float *stat,
                         float *affine, void *stream);
int mmk_bn_apply(const void *a, int64_t npix, int32_t C, const float *affine, float drop_p, uint32_t seed, void *y,
                 void *stream);
int mmk_bn_backward(const void *gd, const void *d, float drop_scale, const void *a, int64_t npix, int32_t C,
                    const float *stat, const float *affine, const float *gamma, float leaky_slope, int32_t accumulate,
                    float *part, float *coef, float *dgamma, float *dbeta, void *gz, void *stream);

/* ------------------------------------------------------------------ the mask U-Net as two calls
 * mmk_unet_forward replaces the network part of LearnICPWeightPolicy.forward
 * (mm_masking/icp_weight_policy.py:161-199: encoder, twice-applied decoder blocks, 1x1 + sigmoid, amax
 * normalisation) and mmk_unet_backward its autograd (train_icp_weights.py:51), sequencing the building
 * blocks above from the host side of the ABI: ~290 launches per training step without a Python call each.
 * Results are bit-identical to issuing the same building blocks one by one (tests/test_gpu_unet_driver.py). */
typedef struct {
    int32_t B, H, W, cin;       /* network input: fp32 NCHW (B,cin,H,W), cin = 1..4, H, W >= 32          */
    const float *x;
    const float *pre;           /* (cin,2) offset / reciprocal scale from mmk_channel_minmax, or NULL     */
    const float *const *params; /* HOST array of 46 device pointers in state_dict order: weight, bias of
                                   encoder.{0..5}.{0,2}, decoder.{0..4}.{0,2}, final_layer.0 (fp32 masters) */
    float drop_p;               /* dropout probability of this pass (0 in eval mode)                      */
    uint32_t seed;              /* dropout stream of this pass                                            */
    float leaky_slope;          /* 0 = ReLU, else nn.LeakyReLU(slope) (params["leaky"])                    */
    int32_t norm;               /* 1: mask / amax(mask) per image (params["norm_weights"])                */
    void *workspace;            /* mmk_unet_workspace_bytes(): packed weights + every activation; written by
                                   the forward, read by the backward -- the caller keeps it in between     */
    size_t workspace_bytes;
    float *mask;                /* (B,H,W) fp32: output of the forward, input of the backward             */
    int32_t keep_full_res;      /* 1: the encoder blocks whose second convolution pools in the same pass also write
                                   their pre-pool output (mmk_unet_tensor ids 7, 8; diagnostics).  0: they write the
                                   pooled tensor and its arg-max codes only -- nothing else reads that output     */
} mmk_unet_desc;

size_t mmk_unet_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t cin);
/* scratch of the backward pass (gradient tensors, partial sums); needs a HIP device (occupancy queries) */
size_t mmk_unet_scratch_bytes(int32_t B, int32_t H, int32_t W, int32_t cin);
int mmk_unet_forward(const mmk_unet_desc *d, void *stream);
/* gmask (B,H,W) fp32 = dL/dmask; grads: HOST array of 46 device pointers, same order and shapes as params,
 * overwritten with dL/dparam. */
int mmk_unet_backward(const mmk_unet_desc *d, const float *gmask, float *const *grads, void *scratch,
                      size_t scratch_bytes, void *stream);
/* Data-parallel form of mmk_unet_backward (nothing upstream to mirror: the reference has no multi-GPU path; BASELINE.json
 * configs[3] asks for the gradient all-reduce to overlap the backward pass).  The parameter gradients of a pass become final
 * in MMK_UNET_GRAD_BUCKETS groups; mmk_unet_grad_bucket tells which contiguous run of the 46 parameters (state_dict order)
 * group `bucket` holds -- 0: decoder + final layer (24..45), 1: encoder blocks 3-5 (12..23), 2: encoder blocks 0-2 (0..11),
 * the order in which they complete.  bucket_events: HOST array of MMK_UNET_GRAD_BUCKETS hipEvent_t the caller created;
 * event b is recorded behind the last kernel that writes a gradient of group b, so another stream that waits for it may
 * read (all-reduce) that group while the rest of the pass is still running.  Results are those of mmk_unet_backward. */
#define MMK_UNET_GRAD_BUCKETS 3
int32_t mmk_unet_grad_bucket(int32_t bucket, int32_t *first_param, int32_t *n_params);
int mmk_unet_backward_buckets(const mmk_unet_desc *d, const float *gmask, float *const *grads, void *scratch,
                              size_t scratch_bytes, void *const *bucket_events, void *stream);
/* Where an activation lives inside the workspace (tests / diagnostics): id 0..5 first conv output of encoder
 * block i, 6..11 second (post-dropout) output, 12..17 t[i] (block output after pooling), 18 + 5 j + {0..4}:
 * decoder block j's up-sampled input, a1, d1, a2, d2.  NHWC bf16 (B,h,w,c) at byte `offset`. */
int mmk_unet_tensor(int32_t B, int32_t H, int32_t W, int32_t cin, int32_t id, size_t *offset, int32_t *h,
                    int32_t *w, int32_t *c);

#ifdef __cplusplus
}
#endif
#endif /* MMK_H_ */
