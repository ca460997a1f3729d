/* et_hip.h -- C ABI of libet_hip.so, the MI355X (gfx950) kernels behind the Efficient-Teacher
 * SSOD training step.
 *
 * The reference (AlibabaResearch/efficientteacher) has no FFI of its own: every function below
 * replaces the stock-PyTorch call sequence at the cited reference file:line, and is bound from
 * Python with ctypes (efficientteacher_amd/_lib.py; stub a reference maintainer would add: see
 * INTEGRATION.md).  Conventions:
 *   - plain pointers + sizes, no torch types; all pointers are DEVICE pointers unless marked host;
 *   - every call is asynchronous on `stream` (a hipStream_t), allocates nothing, keeps no global
 *     state and is re-entrant; scratch comes from the caller (`*_workspace_bytes`);
 *   - return 0 on success, <0 on error: -1 bad pointer, -2 unsupported shape/argument,
 *     -3 workspace too small, <= -100 HIP launch error (-(100+hipError_t));
 *   - activations are NHWC ("channels last": N, H, W, C with C contiguous) in HBM;
 *     `dtype` is ET_F32 (parity mode), ET_BF16 (performance mode, fp32 accumulate) or ET_F16 (the reference's own
 *     reduced precision: torch.cuda.amp autocast + GradScaler, trainer/trainer.py:248,348,399-400; same MFMA rate).
 */
#ifndef ET_HIP_H
#define ET_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* et_stream_t; /* hipStream_t */

enum { ET_F32 = 0, ET_BF16 = 1, ET_F16 = 2 };   /* ET_F16 (r05): IEEE half storage, fp32 accumulate -- the reference's autocast dtype */

/* Library / device identification (host side). Returns the gfx arch the code objects were built
 * for ("gfx950") and the ABI version.  ET_ABI_VERSION changes whenever an entry point is added or a signature / workspace
 * contract changes; a binding must refuse a library whose et_abi_version() differs from the header it was written against
 * (efficientteacher_amd/_lib.py does): a stale libet_hip.so would otherwise read e.g. a new int argument as the stream. */
#define ET_ABI_VERSION 6
const char* et_build_arch(void);
int et_abi_version(void);

/* ---------------------------------------------------------------------------------------------
 * Pseudo-label filter.  Replaces utils/general.py:887-992 non_max_suppression_ssod (multi_label
 * False, classes None, labels ()) including the torchvision.ops.nms call at :976.
 *   pred   (B, A, no) fp32, rows [x, y, w, h, obj, cls_0..cls_{no-6}] (Detect eval output)
 *   dets   (B, max_det, 8) fp32 rows [x1,y1,x2,y2, conf, cls, obj_conf, cls_conf], zero padded
 *   counts (B) int32 number of valid rows per image
 *   keep   (B, max_det) int64: index of each kept row in the reference's pre-NMS candidate matrix
 *          (rows that passed both confidence tests, original anchor order); -1 padded
 *   n_candidates (B) int32, optional (may be NULL): size of that candidate matrix
 * Limits: 6 <= no <= 96, max_det <= 1024, A <= 262144.                                         */
int et_nms_ssod_workspace_bytes(int B, int A, size_t* bytes /*host out*/);
int et_nms_ssod(const float* pred, int B, int A, int no, float conf_thres, float iou_thres,
                int agnostic, int max_det, float* dets, int* counts, int64_t* keep,
                int* n_candidates, void* workspace, size_t ws_bytes, et_stream_t stream);

/* General non_max_suppression: reference utils/general.py:994-1100 (val.py:335 calls it with
 * multi_label=True, conf_thres=0.001).  Same outputs as et_nms_ssod; dets rows are
 * [x1,y1,x2,y2,conf,cls,0,0].  class_mask_{lo,hi}: bit c set = class c allowed (the `classes=` filter,
 * :1061); pass ~0 for no filter.  max_nms = 30000 and max_wh = 7680 in the reference (:1013-1014).
 * `keep` indexes the candidate matrix after the max_nms cut, in (anchor, class) order. */
int et_nms_workspace_bytes(int B, int A, int no, int multi_label, int max_nms, size_t* bytes /*host out*/);
int et_nms(const float* pred, int B, int A, int no, float conf_thres, float iou_thres, int agnostic,
           int multi_label, uint64_t class_mask_lo, uint64_t class_mask_hi, int max_nms, float max_wh,
           int max_det, float* dets, int* counts, int64_t* keep, int* n_candidates, void* workspace,
           size_t ws_bytes, et_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Detect head inference decode.  Replaces models/head/yolov5_head.py:68-78 (+ _make_grid_old
 * :127-136) for one level: reads the head conv output through element strides
 * raw[b*sb + a*sa + y*sy + x*sx + c] (so the NHWC GEMM output is consumed in place) and writes
 * z[b, a_offset + (a*ny + y)*nx + x, c] of z (B, A_total, no) fp32.
 *   anchor_px (na, 2) fp32 = anchors[level] * stride (the reference's anchor_grid values).        */
int et_detect_decode(const void* raw, int dtype, int B, int na, int ny, int nx, int no,
                     int64_t sb, int64_t sa, int64_t sy, int64_t sx, const float* anchor_px,
                     float stride, float* z, int64_t A_total, int64_t a_offset, et_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Flat-arena state updates (one launch instead of the reference's per-tensor python loops).
 * EMA: utils/torch_utils.py:330-338 / :366-375 / :406-416   v = v*d ; v += (1-d)*m   (fp32)
 * SGD: torch.optim.SGD(momentum, nesterov=True) built at trainer/trainer.py:215-223, with the
 *      GradScaler 1/scale (trainer.py:399-400) folded in and an optional 16-bit shadow copy of the
 *      updated values in the compute format (lp_shadow may be NULL; shadow_dtype ET_BF16 | ET_F16).
 *      `scaler` (may be NULL): the DEVICE loss-scaler state {scale, 1/scale, found_inf, growth_tracker} of fp16 mode
 *      (et_scaler_check / et_scaler_update below): found_inf != 0 skips the update (GradScaler.step), else the gradient is
 *      also multiplied by 1/scale (GradScaler.unscale_).  Pointers must be 16-byte aligned for EMA.      */
int et_ema_update(float* ema, const float* model, int64_t n, float d, float one_minus_d, et_stream_t stream);
/* AdamW over a flat arena: torch.optim.AdamW(betas=(hyp.momentum, 0.999)) as built at trainer/trainer.py:212 when cfg.adam
 * (decoupled weight decay, eps 1e-8); `step` = 1-based update count (bias corrections are formed on the host in double).   */
int et_adamw(float* p, const float* grad, float* exp_avg, float* exp_avg_sq, void* lp_shadow, int shadow_dtype, int64_t n, float lr,
             float beta1, float beta2, float eps, float weight_decay, int step, float inv_scale, const float* scaler,
             et_stream_t stream);
/* AdamW under the loss scaler (fp16 mode): the update count lives in device memory, tick = 3 doubles {t, 1 - beta1^t, 1 - beta2^t},
 * because GradScaler.step (trainer/trainer.py:400) does not call optimizer.step() on an overflow -- a skipped step must not
 * advance the bias corrections, and found_inf is only known on the device.  et_adamw_tick: once per optimizer step, advances the
 * count unless scaler's found_inf is set; et_adamw_dev: et_adamw with lr / bc1 and 1 / sqrt(bc2) formed from `tick`.              */
int et_adamw_tick(double* tick, float beta1, float beta2, const float* scaler, et_stream_t stream);
int et_adamw_dev(float* p, const float* grad, float* exp_avg, float* exp_avg_sq, void* lp_shadow, int shadow_dtype, int64_t n, float lr,
                 float beta1, float beta2, float eps, float weight_decay, const double* tick, float inv_scale, const float* scaler,
                 et_stream_t stream);
int et_sgd_nesterov(float* p, const float* grad, float* momentum_buf, void* lp_shadow, int shadow_dtype, int64_t n,
                    float lr, float momentum, float weight_decay, int first_step, float inv_scale, const float* scaler,
                    et_stream_t stream);
/* fp32 -> the 16-bit compute format (dtype ET_BF16 | ET_F16), round to nearest even: the weight shadow of the MFMA kernels */
int et_cast_f32_to_lp(const float* src, void* dst, int dtype, int64_t n, et_stream_t stream);
/* Loss scaler of fp16 mode, device resident (torch.cuda.amp.GradScaler as driven at trainer/trainer.py:248,348,399-400;
 * ssod_trainer.py:595,625).  scaler = 4 floats {scale, 1/scale, found_inf, growth_tracker}.
 *   et_scaler_check : found_inf = 1 if any of the n (scaled, fp32) gradients is inf / nan  (torch's non-finite check)
 *   et_scaler_update: GradScaler.update() -- found_inf ? scale *= backoff, tracker = 0 : (++tracker == interval ? scale *= growth);
 *                     then 1/scale is refreshed and found_inf cleared.                                                      */
int et_scaler_check(const float* grads, int64_t n, float* scaler, et_stream_t stream);
int et_scaler_update(float* scaler, float growth_factor, float backoff_factor, int growth_interval, et_stream_t stream);
/* The same two updates with their per-step scalars in DEVICE memory (d2 = {d, 1-d}; hp = {lr, momentum, weight_decay,
 * inv_scale}): the form a captured HIP graph of the step replays -- warm-up (trainer.py:386-395), the lr schedule and
 * ModelEMA's decay ramp (utils/torch_utils.py:324) change these every step, and kernel arguments are frozen at capture. */
int et_ema_update_dev(float* ema, const float* model, int64_t n, const float* d_and_one_minus_d, et_stream_t stream);
int et_sgd_nesterov_dev(float* p, const float* grad, float* momentum_buf, void* lp_shadow, int shadow_dtype, int64_t n,
                        const float* hp, int first_step, const float* scaler, et_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Convolution (the conv inside `Conv`, models/backbone/common.py:471-481; Detect.m,
 * models/head/yolov5_head.py:30; netD, models/detector/yolo_ssod.py:224-238) as implicit GEMM on
 * MFMA.  x (N, IH, IW, Cin) NHWC with pixel stride ldx (elements, so channel slices of a wider
 * buffer can be read in place); w (Cout, KH, KW, Cin); y (N, OH, OW, *) with pixel stride ldy.
 * Requirements: Cin % 8 == 0 (pad the 3-channel stem input to 8), 16-byte aligned pointers and
 * strides, KH*KW <= 36, tensors < 2^31 elements.  dgrad/wgrad additionally need Cout % 8 == 0.
 *   fwd epilogue: v = acc (* scale[co]) (+ bias[co]); act (0 none, 1 SiLU, 2 ReLU); (+ residual[pix*ldr + co]);
 *   scale/bias = the eval-mode BatchNorm affine of the EMA teacher (et_bn_eval_affine), or the head bias.
 *   stats_partial, if not NULL: (et_conv2d_stats_rows_for(0, ...), 2, Cout) fp32 partial per-channel
 *   sum / sum-of-squares of the raw accumulators (BatchNorm batch statistics, reduced later by
 *   et_bn_finalize) -- every row is fully overwritten.  The row count depends on the kernel the library selects for the
 *   problem (one row per 64 output pixels for the tiled kernels = et_conv2d_stats_rows(N,OH,OW); one row per resident
 *   workgroup and row group for the persistent 1x1 kernel): ask et_conv2d_stats_rows_for with the arguments of the call.
 *   residual MAY alias y exactly (same pointer, ldr == ldy): the in-place shortcut of the eval-mode C3 stem
 *   (models/backbone/common.py C3.forward) -- every lane loads the residual element it is about to overwrite; any OTHER overlap of
 *   the two ranges is rejected (-2).
 *   zero16: device pointer to >= 16 zero bytes (16-byte aligned).  When given, the K-chunks are staged with
 *   LDS-DMA (global_load_lds_dwordx4) and padding / tail lanes fetch from this page; NULL selects the
 *   register-staged kernel.
 *   dgrad: dx = conv_transpose(dy, w) with wT = (Cin, KH, KW, Cout) from et_weight_transpose;
 *          stride 2 is executed as 4 parity-class launches; accumulate != 0 adds into dx.
 *   wgrad: dw (Cout, KH, KW, Cin) fp32 += ... (split-K over pixels, atomicAdd: zero or reuse the
 *          gradient arena as the accumulator).                                                    */
int et_conv2d_stats_rows(int N, int OH, int OW);
/* rows of the partial-statistics buffer for one call: op 0 = stats_partial of et_conv2d_fwd (no residual), op 2 = stats_partial of
 * an et_conv2d_fwd call that also passes `residual` (another kernel form, another row count), op 1 = bn_stats_partial of
 * et_conv2d_dgrad_bn; the remaining arguments are those of the FORWARD conv (as for et_conv2d_kernel_name); have_zero_page = the
 * call passes zero16.  Host only. */
int et_conv2d_stats_rows_for(int op, int dtype, int N, int IH, int IW, int Cin, int Cout, int KH, int KW, int stride, int pad,
                             int have_zero_page);
/* the same question for a SHARDED call (stats_ld > 0): fp32 atomic additions per channel (all shards together) -- one per workgroup
 * that covers the channel (row tiles of the tiled kernels, resident workgroups of the persistent ones).  The caller decides with it
 * whether sharding pays (additions / ET_BN_SHARDS meet on one address and serialise in the memory-side atomic units).  Host only. */
int et_conv2d_stats_adds_for(int op, int dtype, int N, int IH, int IW, int Cin, int Cout, int KH, int KW, int stride, int pad,
                             int have_zero_page);
/* stats_partial with stats_ld == 0: partial rows (et_conv2d_stats_rows_for(...), 2, Cout), every row written -- the exact path
 * (fp64 combine in et_bn_finalize, bit-reproducible).  stats_ld > 0: stats_partial is a SHARDED ACCUMULATOR
 * [ET_BN_SHARDS][2][stats_ld] fp32 that must be ZERO before the launch; every WORKGROUP adds its column sums (its wave rows meet in
 * LDS first) into shard (workgroup index % ET_BN_SHARDS) with hardware fp32 atomics, channel c at [shard][t][c].  et_bn_act_fwd_sharded folds the shards
 * itself, so a Conv block is two launches (conv, normalise) instead of three, and one memset per STEP zeroes every layer's shards
 * when they live in one arena.  Sums are then order-dependent in the last fp32 bits (the 16-bit training modes use this form). */
#define ET_BN_SHARDS 16
int et_conv2d_fwd(const void* x, const void* w, void* y, int dtype, int N, int IH, int IW, int Cin,
                  int ldx, int Cout, int KH, int KW, int stride, int pad, int ldy, const float* scale,
                  const float* bias, int act, const void* residual, int ldr, float* stats_partial, int stats_ld,
                  const void* zero16, et_stream_t stream);
int et_conv2d_dgrad(const void* dy, const void* wT, void* dx, int dtype, int N, int IH, int IW, int Cin,
                    int ldx, int Cout, int KH, int KW, int stride, int pad, int ldy, int accumulate,
                    const void* residual /* optional, stride 1 only: dx = dgrad + residual */, int ldr,
                    const void* zero16, et_stream_t stream);
/* dgrad (stride 1) that ALSO does the reduce pass of the BatchNorm backward of the layer whose activation gradient it
 * produces: dx = dgrad (+ residual) is the dz of a Conv block z = act(BN(y)); with that block's raw conv output
 * bn_y (pixel stride ld_bn), its folded affine bn_scale / bn_shift (Cin values: the channels of dx) and activation,
 * the epilogue accumulates per-tile partial sums of du = dz * act'(y*scale+shift) and du*y into
 * bn_stats_partial (et_conv2d_stats_rows_for(1, ...), 2, Cin) -- computed on the ROUNDED dz it stores, i.e. exactly the
 * values et_bn_act_bwd_from_partials reads back.  Replaces bn_act_bwd's own reduce pass (a 4 B/element re-read of dz
 * and y) by one y read here.  Reference math: torch.nn.BatchNorm2d backward as used by Conv (models/backbone/common.py:480). */
int et_conv2d_dgrad_bn(const void* dy, const void* wT, void* dx, int dtype, int N, int IH, int IW, int Cin, int ldx, int Cout,
                       int KH, int KW, int pad, int ldy, const void* residual, int ldr, const void* bn_y, int ld_bn,
                       const float* bn_scale, const float* bn_shift, int bn_act, float* bn_stats_partial,
                       int bn_stats_ld /* 0: partial rows; > 0: sharded accumulator [ET_BN_SHARDS][2][bn_stats_ld], see et_conv2d_fwd */,
                       const void* zero16, et_stream_t stream);
int et_conv2d_wgrad(const void* x, const void* dy, float* dw, int dtype, int N, int IH, int IW, int Cin,
                    int ldx, int Cout, int KH, int KW, int stride, int pad, int ldy, const void* zero16,
                    et_stream_t stream);
/* The same for up to 16 layers of IDENTICAL geometry in ONE launch: the K-split that fills the chip is shared by
 * the group, so each dW address receives group-size times fewer atomics (the atomic epilogue is ~25 % of a
 * split-28..64 launch).  Items differ only in their pointers and pixel strides. */
typedef struct et_wgrad_item { const void* x; const void* dy; float* dw; int ldx, ldy; } et_wgrad_item;
int et_conv2d_wgrad_grouped(const et_wgrad_item* items /* host array */, int n_items, int dtype, int N, int IH, int IW,
                            int Cin, int Cout, int KH, int KW, int stride, int pad, const void* zero16,
                            et_stream_t stream);
int et_weight_transpose(const void* w, void* wT, int dtype, int Cout, int taps, int Cin, et_stream_t stream);
/* Introspection (host only, launches nothing): the kernel instantiation et_conv2d_fwd (op 0), et_conv2d_dgrad
 * (op 1; for stride 2 `parity_class` 0..3 selects one of its four launches) or et_conv2d_wgrad (op 2) would launch
 * for this problem -- arguments as for et_conv2d_fwd -- spelled the way rocprofv3 prints it.  The tile selection
 * depends on the shape, the dtype, the CU count of the current device and the ET_* tuning knobs; tests assert on
 * this name and bench.py tags its HIP-event timings with it, so there is ONE copy of the selection logic.
 * et_env_knobs: every ET_* tuning knob set in the environment, "NAME=value;..." ("" = all defaults). */
int et_conv2d_kernel_name(int op, int dtype, int N, int IH, int IW, int Cin, int Cout, int KH, int KW, int stride,
                          int pad, int have_zero_page, int parity_class, char* buf /*host out*/, int buflen);
int et_env_knobs(char* buf /*host out*/, int buflen);
/* every layer of a flat weight arena at once: table = n_layers x {element offset, Cout, taps, Cin} (int32, device,
 * sorted by offset); wT_arena has the arena's layout with each layer stored (Cin, taps, Cout). */
int et_weight_transpose_all(const void* w_arena, void* wT_arena, int dtype, const int* table, int n_layers,
                            long long total_elems, et_stream_t stream);
/* out[c] += sum_p x[p*ld + c]  (bias gradient of the Detect convs) */
int et_colsum(const void* x, int dtype, int P, int C, int ld, float* out, et_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * BatchNorm2d + activation (+ residual) of the reference `Conv` block
 * (models/backbone/common.py:471-481, Bottleneck add :544; eps / momentum from
 * utils/torch_utils.py:162-169), NHWC, elementwise 16 B per lane.
 *   et_bn_finalize: (rows,2,C) partial sums from et_conv2d_fwd -> scale = g*invstd,
 *       shift = b - mean*scale, saved mean / invstd for backward, running stats update
 *       (running_* may be NULL; unbiased variance as torch).
 *   et_bn_eval_affine: eval-mode (teacher) scale/shift from the running statistics.
 *   et_bn_act_fwd: z = act(y*scale + shift) (+ residual).
 *   et_bn_act_bwd: dy = dBN/dSiLU(dz) in two passes; dgamma/dbeta (fp32) are ACCUMULATED.
 *       workspace: >= et_bn_reduce_rows(P,C,dtype)*2*C + 3*C floats.  C <= 2048.
 *   `ws` / `totals` (2*C fp64 totals followed by (C+31)/32 int32 tickets, i.e. 2*C*8 + 4*((C+31)/32) bytes):
 *       MUST BE ZERO ON ENTRY and is left zero on return (the last workgroup of each channel group reads the
 *       totals back with an atomic exchange and finalizes), so one zero-initialised scratch per stream serves
 *       every layer: partial-sum reduction + finalize are ONE launch, no memset.
 *   et_act_bwd: dy = dz * act'(y)   (netD ReLU, models/detector/yolo_ssod.py:231-238).          */
int et_bn_reduce_rows(int P, int C, int dtype);
int et_bn_finalize(const float* stats_partial, int rows, int C, double count, const float* gamma,
                   const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                   float* scale, float* shift, float* save_mean, float* save_invstd,
                   double* ws /* totals + tickets, zero in / zero out */, et_stream_t stream);
int et_bn_eval_affine(int C, const float* gamma, const float* beta, const float* running_mean,
                      const float* running_var, float eps, float* scale, float* shift, et_stream_t stream);
int et_bn_act_fwd(const void* y, int ldy, void* z, int ldz, const void* residual, int ldr, int dtype, int P,
                  int C, const float* scale, const float* shift, int act, et_stream_t stream);
int et_bn_act_bwd(const void* dz, int lddz, const void* y, int ldy, void* dy, int lddy, int dtype, int P, int C,
                  const float* gamma, const float* scale, const float* shift, const float* save_mean,
                  const float* save_invstd, int act, float* dgamma, float* dbeta,
                  double* totals /* totals + tickets, zero in / zero out */, float* workspace, size_t ws_floats,
                  et_stream_t stream);
/* et_bn_act_bwd without its reduce pass: `partials` (partial_rows, 2, C) are the per-tile sums of du and du*y that
 * et_conv2d_dgrad_bn left when it produced dz.  workspace: 3*C floats.  totals as above (zero in / zero out). */
int et_bn_act_bwd_from_partials(const void* dz, int lddz, const void* y, int ldy, void* dy, int lddy, int dtype, int P, int C,
                                const float* gamma, const float* scale, const float* shift, const float* save_mean,
                                const float* save_invstd, int act, float* dgamma, float* dbeta, double* totals,
                                const float* partials, int partial_rows, float* workspace, et_stream_t stream);
/* The Conv block's normalise pass and its backward on SHARDED sums (et_conv2d_fwd stats_ld > 0): no finalize launch.
 * et_bn_act_fwd_sharded = et_bn_finalize + et_bn_act_fwd in one launch: every workgroup folds the ET_BN_SHARDS shards of its channels
 * (fp64), derives scale / shift into LDS and streams z = act(y*scale + shift) (+ residual); workgroup 0 also writes scale / shift /
 * save_mean / save_invstd and updates the running statistics.  shards: [ET_BN_SHARDS][2][shard_ld], this layer's channel 0 at
 * shards[0].  C <= 1024.
 * et_bn_act_bwd_sharded: reduce != 0 runs the reduce pass first (sums of du and du*xhat ADDED into the zeroed shards); reduce == 0
 * means et_conv2d_dgrad_bn (bn_stats_ld > 0) already left sums of du and du*y there.  The apply pass folds the shards per workgroup
 * and workgroup 0 accumulates dgamma / dbeta.  Same math as et_bn_finalize / et_bn_act_bwd (one device function each). */
int et_bn_act_fwd_sharded(const void* y, int ldy, void* z, int ldz, const void* residual, int ldr, int dtype, int P, int C,
                          const float* shards, int shard_ld, double count, const float* gamma, const float* beta, float eps,
                          float momentum, float* running_mean, float* running_var, float* scale, float* shift, float* save_mean,
                          float* save_invstd, int act, et_stream_t stream);
int et_bn_act_bwd_sharded(const void* dz, int lddz, const void* y, int ldy, void* dy, int lddy, int dtype, int P, int C,
                          const float* gamma, const float* scale, const float* shift, const float* save_mean,
                          const float* save_invstd, int act, float* dgamma, float* dbeta, float* shards, int shard_ld, int reduce,
                          et_stream_t stream);
int et_act_bwd(const void* dz, int lddz, const void* y, int ldy, void* dy, int lddy, int dtype, int P, int C,
               int act, et_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Spatial data movement on NHWC tensors.
 *   et_pack_input: (B,C<=8,H,W) fp32 NCHW image -> (B,H,W,8) NHWC, zero padded channels (feeds the
 *       stem conv, models/backbone/yolov5_backbone.py:56).
 *   et_maxpool5_*: nn.MaxPool2d(5,1,2) of SPPF (models/backbone/common.py:702-708); argmax is
 *       (B,H,W,C) uint8 window codes ky*5+kx (first maximum in scan order, torch's tie rule);
 *       bwd: dx = base + gather(dy) (base may be NULL).
 *   et_upsample2x_*: nn.Upsample(scale_factor=2, 'nearest') (models/neck/yolov5_neck.py:60,64).   */
int et_pack_input(const float* x_nchw, void* y_nhwc8, int dtype, int B, int C, int H, int W, et_stream_t stream);
/* the loader's uint8 batch directly: y = (float)x / norm_scale (IEEE division == torch `imgs.float() / 255.0`,
 * trainer/trainer.py:411, trainer/ssod_trainer.py:694-696) packed to NHWC8 in one pass */
int et_pack_input_u8(const uint8_t* x_nchw, void* y_nhwc8, int dtype, int B, int C, int H, int W, float norm_scale,
                     et_stream_t stream);
int et_maxpool5_fwd(const void* x, int ldx, void* y, int ldy, uint8_t* argmax, int dtype, int B, int H, int W, int C,
                    et_stream_t stream);
int et_maxpool5_bwd(const void* dy, int lddy, const uint8_t* argmax, const void* base, int ldb, void* dx, int lddx,
                    int dtype, int B, int H, int W, int C, et_stream_t stream);
int et_upsample2x_fwd(const void* x, int ldx, void* y, int ldy, int dtype, int B, int H, int W, int C, et_stream_t stream);
int et_upsample2x_bwd(const void* dy, int lddy, void* dx, int lddx, int dtype, int B, int H, int W, int C,
                      int accumulate /* dx += (the other consumer's gradient is already there) */, et_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Strong view of an unlabeled batch from its weak view: the per-pixel work of the reference's data-loader
 * workers -- cv2.warpAffine(M[:2], borderValue 114) of random_perspective_with_M (utils/datasets_ssod.py:902-945),
 * augment_hsv's LUTs between BGR2HSV / HSV2BGR (utils/augmentations.py:48-61), cutout rectangles (:382-398), flipud / fliplr
 * (datasets_ssod.py:552-563) -- fused, one thread per output pixel.  weak / strong: (B,3,H,W) uint8 RGB planes;
 * minv [B][6] fp64: dst->src affine map (OpenCV's inverse of M[:2]); lut [B][3][256] uint8 (hue, sat, val) or NULL;
 * cutouts [B][32][7] int32 (x0,y0,x1,y1 half open, r,g,b); flags [B][3] int32 (n_cutouts, flipud, fliplr).
 * PARITY UNPINNED (cv2 absent from the build image): restates OpenCV's published 8-bit algorithms, see csrc/augment.hip.   */
int et_strong_view_u8(const uint8_t* weak, uint8_t* strong, int B, int H, int W, const double* minv, const uint8_t* lut,
                      const int* cutouts, const int* flags, int border_value, et_stream_t stream);
/* 4-image mosaic of the SSOD loader (load_mosaic_with_M, utils/datasets_ssod.py:732-782) without its 2s x 2s canvas: out (B, 3, S, S)
 * uint8 = the 2:1 box average ((a + b + c + d + 2) >> 2: what cv2.resize(img4, (S, S)) computes at this exact ratio) of four images
 * pasted around a centre over border_value.  tiles: DEVICE table [B][4][8] int64 = {device pointer of the tile's (3, h, w) uint8
 * planes, h, w, x1a, y1a, x2a, y2a, x1b << 32 | y1b}: canvas rectangle and its origin inside the image, tile order top-left,
 * top-right, bottom-left, bottom-right (x2a / y2a of tile 0 = the centre).  Host recipe: efficientteacher_amd/utils/augment.py. */
int et_mosaic4_u8(const int64_t* tiles, uint8_t* out, int B, int S, int border_value, et_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Online pseudo labels.  Replaces FairPseudoLabel.create_pseudo_label_online_with_gt
 * (utils/self_supervised_utils.py:194-245) after the NMS: per detection xyxy->xywh (fp32),
 * then in fp64 the affine warp by M_s (B,13) [img, M00..M22, s, ud, lr], clip, box_candidates
 * filter, normalise, flips.  targets9 (B*max_det, 9) fp64 rows
 * [img, cls, x, y, w, h, conf, obj_conf, cls_conf] stay PADDED, valid (B*max_det) uint8 marks
 * the surviving rows (the reference's row order == ascending padded index).                    */
int et_pseudo_label_transform(const float* dets, const int* counts, const double* M_s, int B, int max_det,
                              int width, int height, int clip01 /* LabelMatch: utils/labelmatch.py:333 */,
                              double* targets9, uint8_t* valid, et_stream_t stream);
/* LabelMatch's per-class score lists (utils/labelmatch.py:279-287, read by update_epoch_cls_thr :188-240): append the
 * (confidence, class) of every NMS detection to a device log; *log_count (device, 64-bit) counts all appends, entries
 * beyond `cap` are dropped (the host checks the counter at the end of the epoch).                                       */
int et_score_log_append(const float* dets, const int* counts, int B, int max_det, float* conf_log, int* cls_log,
                        uint64_t* log_count, int64_t cap, et_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Detection losses with fused anchor assignment, forward + gradient.  Replaces
 * YOLOAnchorAssigner.build_targets / build_uc_targets_aug
 * (models/assigner/yolo_anchor_assigner.py:319-372, 640-696), bbox_iou CIoU (utils/metrics.py:207),
 * ComputeLoss.default_loss (models/loss/loss.py:138-208) and
 * ComputeStudentMatchLoss.default_loss (models/loss/ssod/ssod_loss.py:194-288).
 *   targets table (NT, 8) fp32 rows [img, cls, x, y, w, h, score, flags]; flags bit p = row takes part
 *   in pass p: 0 reliable / labelled (box + cls + tobj = CIoU), 1 uncertain (tobj = score, or -1
 *   "ignore" if ignore_obj), 2 uncertain with obj_conf >= .99 (extra box term), 3 uncertain with
 *   cls_conf >= .99 (extra cls term).  Supervised loss: every row has flags = 1.
 *   level[l].p: logits viewed as [b*sb + a*sa + y*sy + x*sx + c]; level[l].dp: fp32 gradient buffer
 *   with the same strides, ZEROED by the caller, receives d(lbox*box_w + lobj*obj_w + lcls*cls_w)/dp
 *   (without the reference's final *bs); tobj_ws: B*na*ny*nx uint64 scratch; acc_ws: 64 floats.
 *   out (8 floats, device): [lbox*box_w, lobj*obj_w, lcls*cls_w, (sum)*B, n_pos pass0..3].
 * et_select_targets builds the table from (N,9) fp64 pseudo labels
 * (ComputeStudentMatchLoss.select_targets, ssod_loss.py:130-192; thresholds per class, fp64).
 * et_scale_cast: dst = (T)(src * scale * (dev_scale ? *dev_scale : 1)): hands dp to the head's
 * dgrad/wgrad with the autograd upstream factor (a device scalar) folded in, no host sync.      */
typedef struct {
    const void* p;
    float* dp;
    void* tobj_ws;
    int64_t sb, sa, sy, sx;
    int ny, nx;
    float anchors[6];
    float balance;
} et_loss_level;
typedef struct {
    int dtype, B, na, nc, NT, nl;
    float anchor_t, gr, cp, cn, cls_pw, obj_pw, box_w, obj_w, cls_w;
    int pass_mask, ignore_obj;
    const float* targets;
    float* acc_ws;
    float* out;
    const int* ota_match;   /* NULL, or the result of et_ota_assign: pass 0 takes its positives from it */
    int obj_channel;        /* 0 = default (4); the SimOTA half of ComputeLoss.ota_loss reads objectness from no-1 */
    float* balance_dev;     /* NULL (level[].balance is used), or device [nl] objectness balance weights ... */
    int autobalance_ssi;    /* ... updated after use as Loss.autobalance does (loss.py:193-197): index of the stride-16 level, -1 = leave them */
    float fl_gamma;         /* > 0: FocalLoss(gamma, alpha 0.25) around the class / objectness BCE (loss.py:37-62, :112-114) */
    et_loss_level level[4];
} et_loss_desc;
int et_yolo_loss(const et_loss_desc* desc /*host*/, et_stream_t stream);
/* SimOTA dynamic-k matching: YOLOAnchorAssigner.build_ota_targets (models/assigner/yolo_anchor_assigner.py:104-264) over the
 * candidates of find_3_positive (:266-317), for the logits / targets / anchors of `desc` (dp, tobj_ws, acc_ws, out unused).
 * strides[nl] = Detect.stride, img_size = the literal 640 of :128, top_k = Loss.top_k (<= 13).
 * match[nl][5*na*NT] (device int32): index of the target matched to each candidate slot (slot order = reference candidate
 * order: offset-major, anchor, target), -1 = not a positive.  Feed it to et_yolo_loss through desc->ota_match.
 * Equal costs: the smaller slot index / the earlier target of the image wins (torch.topk / torch.min leave it open).     */
int et_ota_workspace_bytes(int B, int na, int nl, int NT, size_t* bytes);
int et_ota_assign(const et_loss_desc* desc /*host*/, const float* strides /*host, nl*/, float img_size, int top_k,
                  void* workspace, int* match, et_stream_t stream);
int et_select_targets(const double* targets9, const uint8_t* valid, int N, const double* thr_low,
                      const double* thr_high, int nc, int with_obj, float* table, et_stream_t stream);
int et_scale_cast(const float* src, void* dst, int dtype, int64_t n, float scale,
                  const float* dev_scale /*device scalar or NULL*/, et_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Domain-adaptation branch (SSOD.with_da_loss): DomainLoss / TargetLoss
 * (models/loss/loss.py:376-421, DomainFocalLoss :312-368: softmax over the 2 netD logits, alpha 1,
 * gamma 2) for one pyramid level: loss_sum[0] += sum_pixels -(1-p_label)^2 log p_label, and
 * grad[pix*ldg + {0,1}] = dL/dlogit * gscale (grad may be NULL).  feat/grad: NHWC, channels 0,1 used.
 * et_scale_inplace: x *= alpha * (dev_scale ? *dev_scale : 1)  (GradReverse backward,
 * models/detector/yolo_ssod.py:158-171, alpha = -1; upstream factor of the focal gradient).       */
int et_domain_focal(const void* feat, int ldf, int dtype, int64_t P, int label, float gscale, void* grad,
                    int ldg, float* loss_sum, et_stream_t stream);
int et_scale_inplace(void* x, int dtype, int64_t n, float alpha, const float* dev_scale /*or NULL*/,
                     et_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * YOLOv8 anchor-free head (BASELINE.json configs[4]).
 * et_v8_decode: inference branch of YoloV8Detect.forward (models/head/yolov8_head.py:172-214) for one level: DFL
 *   expectation of the four sides, dist2bbox 'xywh' around the cell centre (x + cell_offset, y + cell_offset), * stride;
 *   rows [cx, cy, w, h, 1, sigmoid(cls)] of out (B, A_total, 5 + nc) fp32 at anchor offset a_offset.  reg (B,H,W,>=4*(reg_max+1))
 *   and cls (B,H,W,>=nc) are the NHWC outputs of the two head branches with pixel strides ld_reg / ld_cls.
 * et_tal_assign: TaskAlignedAssigner.forward (models/assigner/tal_assigner.py:30-158; alpha 1, beta 6, top_k 13 in
 *   ComputeTalLoss, models/loss/tal_loss.py:43).  pd_scores (B,A,nc) sigmoid scores, pd_bboxes (B,A,4) xyxy, anc_points (A,2),
 *   gt_labels (B,G), gt_bboxes (B,G,4) xyxy, mask_gt (B,G), all fp32.  Outputs: target_labels (B,A) int64 (label of the
 *   assigned gt; of gt 0 for background anchors, as the reference), target_bboxes (B,A,4), target_scores (B,A,nc)
 *   (one-hot * normalised alignment metric), fg_mask (B,A) uint8.  Equal metrics: the smaller anchor index is taken first
 *   (torch.topk leaves ties unspecified; they only occur at metric 0, where the target score is 0).  A <= 16384. */
int et_v8_decode(const void* reg, int ld_reg, const void* cls, int ld_cls, int dtype, int B, int H, int W, int reg_max, int nc,
                 float stride, float cell_offset, float* out, int64_t A_total, int64_t a_offset, et_stream_t stream);
/* et_tal_loss: ComputeTalLoss after the assigner (models/loss/tal_loss.py:104-128), forward + gradient: varifocal-free BCE class
 *   term, GIoU (iou_kind 1) or IoU (0) box term and the DFL term, each normalised by max(sum target_scores, 1) and weighted;
 *   out = [w_iou*loss_iou, w_dfl*loss_dfl, w_class*loss_cls, total]; grad_* = d total / d logits.  The reference's BboxLoss /
 *   VarifocalLoss classes are absent from its tree: the spec is oracle/v8.py::tal_loss (parity unpinned, self-validated). */
int et_tal_loss(const float* pred_scores, const float* pred_distri, const float* anchor_points_s, const float* stride_tensor,
                const float* target_bboxes_px, const float* target_scores, const uint8_t* fg_mask, int B, int A, int nc, int reg_max,
                int iou_kind, float w_class, float w_iou, float w_dfl, float* grad_scores, float* grad_distri,
                float* acc_ws /* 4 floats, zero on entry */, float* out /* 4 floats */, et_stream_t stream);
int et_tal_assign_workspace_bytes(int B, int A, int G, size_t* bytes /*host out*/);
int et_tal_assign(const float* pd_scores, const float* pd_bboxes, const float* anc_points, const float* gt_labels,
                  const float* gt_bboxes, const float* mask_gt, int B, int A, int G, int nc, int topk, float alpha, float beta,
                  float eps, int64_t* target_labels, float* target_bboxes, float* target_scores, uint8_t* fg_mask,
                  void* workspace, size_t ws_bytes, et_stream_t stream);
/* et_tal_targets_pad: ComputeTalLoss.preprocess (models/loss/tal_loss.py:131-143) without the host loop: targets (n,6) fp32
 *   [img, cls, x, y, w, h] normalised -> out (B,G,5) [cls, x1, y1, x2, y2] pixels (padded rows: cls -1, zero box) and mask (B,G)
 *   = (x1 + y1 + x2 + y2 > 0), a row's slot = the number of earlier rows of its image.  G >= the largest per-image count (G = n is
 *   always enough). */
int et_tal_targets_pad(const float* targets, int n, int B, int G, float img_w, float img_h, float* out, float* mask, et_stream_t stream);

/* ---- EXTENSION (no counterpart in the reference): pseudo labels on the anchor-free head.  The reference's
 * ComputeStudentMatchLoss needs det.anchors (models/loss/ssod/ssod_loss.py:69) and its SSOD trainer raises for model types other
 * than yolov5 (trainer/ssod_trainer.py:598-606), although update_train_logger anticipates a 'tal' variant (:271-272).  These three
 * entry points carry the reliable / uncertain split of ssod_loss.py:130-192 over to TaskAlignedAssigner targets; the specification
 * is oracle/v8.py::tal_student_match_loss (parity unpinned by construction).
 * et_tal_pseudo_split: padded pseudo-label rows targets9 (B*G, 9) fp64 [img, cls, x, y, w, h (normalised), conf, obj_conf, cls_conf]
 *   (+ valid mask or NULL) and the per-class thresholds (fp64, nc each) -> ONE padded gt table per set, as ComputeTalLoss.preprocess
 *   builds it (rows outside the set: label -1, zero box): gt_labels_r/_u (B,G), gt_bboxes_r/_u (B,G,4) xyxy pixels,
 *   mask_reliable / mask_uncertain (B,G) 0/1, u_score (B,G) = obj_conf (with_obj) or conf of the uncertain rows, u_flags (B,G):
 *   bit 0 = contributes box / DFL terms (with_bbox and obj_conf >= 0.99), bit 1 = full-strength class target (with_cls and
 *   cls_conf >= 0.99); both only under with_obj, as the reference forms those subsets.
 * et_tal_assigned_gt: the (B,A) gt index of the last et_tal_assign that used `workspace` (valid where its fg_mask is set).
 * et_tal_merge_pseudo: per anchor: uncertain owner (fg_u) wins over reliable owner (fg_r) -- ssod_loss.py:231 then :248 --;
 *   target_scores = ts_u * (u_score or 1) | ts_r | 0, target_bboxes from the winner, fg_box = reliable, or uncertain with bit 0. */
int et_tal_pseudo_split(const double* targets9, const uint8_t* valid, const double* thr_low, const double* thr_high, int B, int G, int nc,
                        int with_obj, int with_bbox, int with_cls, float img_w, float img_h, float* gt_labels_r, float* gt_bboxes_r,
                        float* gt_labels_u, float* gt_bboxes_u, float* mask_reliable, float* mask_uncertain, float* u_score,
                        uint8_t* u_flags, et_stream_t stream);
int et_tal_assigned_gt(const void* workspace, int B, int A, int G, int32_t* gt_idx, et_stream_t stream);
int et_tal_merge_pseudo(const float* ts_r, const float* tb_r, const uint8_t* fg_r, const float* ts_u, const float* tb_u,
                        const uint8_t* fg_u, const int32_t* gt_idx_u, const float* u_score, const uint8_t* u_flags, int B, int A, int G,
                        int nc, float* target_scores, float* target_bboxes, uint8_t* fg_box, et_stream_t stream);


#ifdef __cplusplus
}
#endif
#endif /* ET_HIP_H */
