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
 *     `dtype` is ET_F32 (parity mode) or ET_BF16 (performance mode, fp32 accumulate).
 */
#ifndef ET_HIP_H
#define ET_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* et_stream_t; /* hipStream_t */

enum { ET_F32 = 0, ET_BF16 = 1 };

/* Library / device identification (host side). Returns the gfx arch the code objects were built
 * for ("gfx950") and the ABI version. */
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

#ifdef __cplusplus
}
#endif
#endif /* ET_HIP_H */
