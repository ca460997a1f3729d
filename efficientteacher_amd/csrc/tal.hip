// YOLOv8 anchor-free head (BASELINE.json configs[4]): inference decode of YoloV8Detect and the TaskAlignedAssigner.
//
//   v8_decode      reference models/head/yolov8_head.py:172-214 (eval branch): per anchor point the DFL expectation
//                  softmax(reg[side, 0..reg_max]) . [0..reg_max] of the four sides, dist2bbox(..., 'xywh')
//                  (models/module/nanodet_utils.py:92-103) around the cell centre (x + 0.5, y + 0.5), times the stride;
//                  row = [cx, cy, w, h, 1, sigmoid(cls_0..nc-1)].  Reads the NHWC outputs of the two head branches in place.
//   tal_*          reference models/assigner/tal_assigner.py:13-158 + select_candidates_in_gts / select_highest_overlaps /
//                  iou_calculator (models/module/nanodet_utils.py:181-243): metric = score^alpha * IoU^beta inside the gt box,
//                  top-k anchors per gt, an anchor claimed by several gts goes to the one with the highest IoU, targets
//                  scaled by the normalised alignment metric.  Four launches, everything stays on the device:
//                    tal_topk     one workgroup per (image, gt): metric of all anchors in LDS, k rounds of argmax
//                    tal_resolve  one thread per (image, anchor): count claims, break multi-claims by IoU (first max)
//                    tal_gtmax    one workgroup per (image, gt): max metric / max IoU over the anchors it finally owns
//                    tal_emit     one thread per (image, anchor): labels, boxes, normalised one-hot scores, fg mask
//                  Ties: torch.topk leaves the order of EQUAL values unspecified; here the smaller anchor index wins.  Equal
//                  metrics only occur at exactly 0 (anchor inside the gt whose predicted box misses it), where the pick has
//                  no effect on the loss (its target score is 0); tests compare on the anchors with a non-zero target.
// Compiled with -ffp-contract=off (index decisions depend on fp32 results).
#include "et_device.h"
#include "../../include/et_hip.h"
#include <math.h>

template <typename T>
__global__ __launch_bounds__(256) void v8_decode_kernel(const T* __restrict__ reg, int ld_reg, const T* __restrict__ cls, int ld_cls,
                                                        int B, int H, int W, int reg_max, int nc, float stride, float cell_offset,
                                                        float* __restrict__ out, long long A_total, long long a_offset) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;     // (b, y, x)
    const long long total = (long long)B * H * W;
    if (i >= total) return;
    const int x = (int)(i % W), y = (int)((i / W) % H);
    const long long b = i / ((long long)W * H);
    const T* r = reg + i * ld_reg;
    float d[4];
    const int nb = reg_max + 1;
    for (int s = 0; s < 4; ++s) {
        float m = -INFINITY;
        for (int k = 0; k < nb; ++k) m = fmaxf(m, et_elem<T>::ld(r[s * nb + k]));
        float den = 0.f, num = 0.f;
        for (int k = 0; k < nb; ++k) {
            const float e = expf(et_elem<T>::ld(r[s * nb + k]) - m);
            den += e;
            num += e * (float)k;
        }
        d[s] = num / den;
    }
    const float ax = (float)x + cell_offset, ay = (float)y + cell_offset;
    const float x1 = ax - d[0], y1 = ay - d[1], x2 = ax + d[2], y2 = ay + d[3];
    float* o = out + (b * A_total + a_offset + (long long)y * W + x) * (5 + nc);
    o[0] = (x1 + x2) / 2 * stride;
    o[1] = (y1 + y2) / 2 * stride;
    o[2] = (x2 - x1) * stride;
    o[3] = (y2 - y1) * stride;
    o[4] = 1.0f;
    const T* c = cls + i * ld_cls;
    for (int k = 0; k < nc; ++k) o[5 + k] = et_sigmoid(et_elem<T>::ld(c[k]));
}

extern "C" int et_v8_decode(const void* reg, int ld_reg, const void* cls, int ld_cls, int dtype, int B, int H, int W, int reg_max,
                            int nc, float stride, float cell_offset, float* out, int64_t A_total, int64_t a_offset,
                            et_stream_t stream) {
    if (!reg || !cls || !out) return -1;
    if (B <= 0 || H <= 0 || W <= 0 || reg_max < 0 || reg_max > 31 || nc <= 0) return -2;
    const long long total = (long long)B * H * W;
    const dim3 grid(et_cdiv(total, 256));
    if (dtype == ET_F32)
        hipLaunchKernelGGL((v8_decode_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)reg, ld_reg, (const float*)cls,
                           ld_cls, B, H, W, reg_max, nc, stride, cell_offset, out, (long long)A_total, (long long)a_offset);
    else if (dtype == ET_BF16)
        hipLaunchKernelGGL((v8_decode_kernel<uint16_t>), grid, dim3(256), 0, (hipStream_t)stream, (const uint16_t*)reg, ld_reg,
                           (const uint16_t*)cls, ld_cls, B, H, W, reg_max, nc, stride, cell_offset, out, (long long)A_total,
                           (long long)a_offset);
    else if (dtype == ET_F16)
        hipLaunchKernelGGL((v8_decode_kernel<et_f16>), grid, dim3(256), 0, (hipStream_t)stream, (const et_f16*)reg, ld_reg,
                           (const et_f16*)cls, ld_cls, B, H, W, reg_max, nc, stride, cell_offset, out, (long long)A_total,
                           (long long)a_offset);
    else return -2;
    ET_CHECK_LAUNCH();
    return 0;
}

// ---- TaskAlignedAssigner ----------------------------------------------------------------------------------------
struct TalArgs {
    const float* scores;   // (B, A, nc) sigmoid class scores
    const float* boxes;    // (B, A, 4) predicted xyxy
    const float* anc;      // (A, 2) anchor points
    const float* glabel;   // (B, G)
    const float* gbox;     // (B, G, 4) xyxy
    const float* gmask;    // (B, G) 1 = real gt
    int B, A, G, nc, topk;
    float alpha, beta, eps;
    unsigned char* pos;    // (B, G, A) workspace: 1 = anchor a is a top-k candidate of gt g inside its box
    int* idx;              // (B, A) final gt index
    int* fg;               // (B, A) final foreground flag
    float* gmax;           // (B, G, 2) max metric / max IoU over the anchors the gt finally owns
};

// iou_calculator (nanodet_utils.py:181-200): box1 = gt, box2 = prediction
__device__ __forceinline__ float tal_iou(const float* g, const float* p, float eps) {
    const float x1 = fmaxf(g[0], p[0]), y1 = fmaxf(g[1], p[1]), x2 = fminf(g[2], p[2]), y2 = fminf(g[3], p[3]);
    const float overlap = fmaxf(x2 - x1, 0.f) * fmaxf(y2 - y1, 0.f);
    const float a1 = fmaxf(g[2] - g[0], 0.f) * fmaxf(g[3] - g[1], 0.f);
    const float a2 = fmaxf(p[2] - p[0], 0.f) * fmaxf(p[3] - p[1], 0.f);
    const float uni = a1 + a2 - overlap + eps;
    return overlap / uni;
}
__device__ __forceinline__ float tal_metric(const TalArgs& t, int b, int g, int a, float* iou_out) {
    const float* gb = t.gbox + ((size_t)b * t.G + g) * 4;
    const float* pb = t.boxes + ((size_t)b * t.A + a) * 4;
    int lab = (int)t.glabel[(size_t)b * t.G + g];
    if (lab < 0) lab += t.nc;                                  // torch indexing with -1 (padded rows) wraps around
    const float sc = t.scores[((size_t)b * t.A + a) * t.nc + lab];
    const float iou = tal_iou(gb, pb, t.eps);
    if (iou_out) *iou_out = iou;
    return powf(sc, t.alpha) * powf(iou, t.beta);
}
__device__ __forceinline__ bool tal_in_gt(const TalArgs& t, int b, int g, int a) {
    const float* gb = t.gbox + ((size_t)b * t.G + g) * 4;
    const float ax = t.anc[2 * a], ay = t.anc[2 * a + 1];
    const float m = fminf(fminf(ax - gb[0], ay - gb[1]), fminf(gb[2] - ax, gb[3] - ay));
    return m > t.eps;
}

#define TAL_MAX_A 16384
__global__ __launch_bounds__(256) void tal_topk_kernel(TalArgs t) {
    __shared__ float met[TAL_MAX_A];
    __shared__ float rv[256];
    __shared__ int ri[256];
    const int b = blockIdx.x / t.G, g = blockIdx.x % t.G;
    unsigned char* pos = t.pos + ((size_t)b * t.G + g) * t.A;
    if (t.gmask[(size_t)b * t.G + g] <= 0.f) {                  // padded gt: its k picks all collapse onto index 0 and are dropped
        for (int a = threadIdx.x; a < t.A; a += 256) pos[a] = 0;   // (:139-142); block-uniform, so no metric scan at all
        return;
    }
    for (int a = threadIdx.x; a < t.A; a += 256) {
        pos[a] = 0;
        const bool in = tal_in_gt(t, b, g, a);
        met[a] = in ? tal_metric(t, b, g, a, nullptr) : 0.f;   // metric * mask_in_gts (tal_assigner.py:93)
    }
    __syncthreads();
    for (int k = 0; k < t.topk && k < t.A; ++k) {
        float bv = -1.f;
        int bi = 0x7fffffff;
        for (int a = threadIdx.x; a < t.A; a += 256) {
            const float v = met[a];
            if (v > bv) { bv = v; bi = a; }                      // ascending scan: the smaller index wins ties
        }
        rv[threadIdx.x] = bv; ri[threadIdx.x] = bi;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (threadIdx.x < s) {
                const float ov = rv[threadIdx.x + s];
                const int oi = ri[threadIdx.x + s];
                if (ov > rv[threadIdx.x] || (ov == rv[threadIdx.x] && oi < ri[threadIdx.x])) { rv[threadIdx.x] = ov; ri[threadIdx.x] = oi; }
            }
            __syncthreads();
        }
        const int win = ri[0];
        if (threadIdx.x == 0) {
            if (tal_in_gt(t, b, g, win)) pos[win] = 1;          // mask_topk * mask_in_gts * mask_gt
            met[win] = -2.f;                                    // taken
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void tal_resolve_kernel(TalArgs t) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)t.B * t.A) return;
    const int b = (int)(i / t.A), a = (int)(i % t.A);
    int cnt = 0, first = 0;
    for (int g = 0; g < t.G; ++g)
        if (t.pos[((size_t)b * t.G + g) * t.A + a]) { if (!cnt) first = g; ++cnt; }
    int idx = first;                                            // mask_pos.argmax(-2): first gt with a 1, or 0
    if (cnt > 1) {                                              // select_highest_overlaps: argmax of the IoU over ALL gts
        float best = -1.f;
        for (int g = 0; g < t.G; ++g) {
            float iou;
            tal_metric(t, b, g, a, &iou);
            if (iou > best) { best = iou; idx = g; }
        }
    }
    t.idx[i] = idx;
    t.fg[i] = cnt > 0 ? 1 : 0;
}

__global__ __launch_bounds__(256) void tal_gtmax_kernel(TalArgs t) {
    __shared__ float rm[256], ro[256];
    const int b = blockIdx.x / t.G, g = blockIdx.x % t.G;
    float bm = 0.f, bo = 0.f;                                   // align_metric * mask_pos and overlaps * mask_pos are >= 0
    for (int a = threadIdx.x; a < t.A; a += 256) {
        const size_t i = (size_t)b * t.A + a;
        if (t.fg[i] && t.idx[i] == g) {
            float iou;
            const float m = tal_metric(t, b, g, a, &iou);
            bm = fmaxf(bm, m); bo = fmaxf(bo, iou);
        }
    }
    rm[threadIdx.x] = bm; ro[threadIdx.x] = bo;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) { rm[threadIdx.x] = fmaxf(rm[threadIdx.x], rm[threadIdx.x + s]); ro[threadIdx.x] = fmaxf(ro[threadIdx.x], ro[threadIdx.x + s]); }
        __syncthreads();
    }
    if (threadIdx.x == 0) { t.gmax[((size_t)b * t.G + g) * 2] = rm[0]; t.gmax[((size_t)b * t.G + g) * 2 + 1] = ro[0]; }
}

__global__ __launch_bounds__(256) void tal_emit_kernel(TalArgs t, long long* __restrict__ tlabel, float* __restrict__ tbox,
                                                       float* __restrict__ tscore, unsigned char* __restrict__ fg_out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)t.B * t.A) return;
    const int b = (int)(i / t.A), a = (int)(i % t.A);
    const int g = t.idx[i], fg = t.fg[i];
    long long lab = (long long)t.glabel[(size_t)b * t.G + g];
    if (lab < 0) lab = 0;                                       // tal_assigner.py:153
    tlabel[i] = lab;
    const float* gb = t.gbox + ((size_t)b * t.G + g) * 4;
    for (int k = 0; k < 4; ++k) tbox[i * 4 + k] = gb[k];
    float norm = 0.f;
    if (fg) {
        const float m = tal_metric(t, b, g, a, nullptr);
        norm = m * t.gmax[((size_t)b * t.G + g) * 2 + 1] / (t.gmax[((size_t)b * t.G + g) * 2] + t.eps);
    }
    float* ts = tscore + i * t.nc;
    for (int k = 0; k < t.nc; ++k) ts[k] = (fg && k == (int)lab) ? norm : 0.f;
    fg_out[i] = (unsigned char)fg;
}

extern "C" int et_tal_assign_workspace_bytes(int B, int A, int G, size_t* bytes) {
    if (!bytes || B <= 0 || A <= 0 || G < 0) return -2;
    *bytes = (size_t)B * (G > 0 ? G : 1) * A + 16 + (size_t)B * A * 8 + (size_t)B * (G > 0 ? G : 1) * 8 + 64;
    return 0;
}

extern "C" int et_tal_assign(const float* pd_scores, const float* pd_bboxes, const float* anc_points, const float* gt_labels,
                             const float* gt_bboxes, const float* mask_gt, int B, int A, int G, int nc, int topk, float alpha,
                             float beta, float eps, int64_t* target_labels, float* target_bboxes, float* target_scores,
                             uint8_t* fg_mask, void* workspace, size_t ws_bytes, et_stream_t stream) {
    if (!pd_scores || !pd_bboxes || !anc_points || !target_labels || !target_bboxes || !target_scores || !fg_mask || !workspace) return -1;
    if (B <= 0 || A <= 0 || A > TAL_MAX_A || G <= 0 || nc <= 0 || topk <= 0) return -2;
    if (!gt_labels || !gt_bboxes || !mask_gt) return -1;
    size_t need;
    et_tal_assign_workspace_bytes(B, A, G, &need);
    if (ws_bytes < need) return -3;
    TalArgs t;
    t.scores = pd_scores; t.boxes = pd_bboxes; t.anc = anc_points; t.glabel = gt_labels; t.gbox = gt_bboxes; t.gmask = mask_gt;
    t.B = B; t.A = A; t.G = G; t.nc = nc; t.topk = topk; t.alpha = alpha; t.beta = beta; t.eps = eps;
    char* w = (char*)workspace;
    t.pos = (unsigned char*)w; w += (((size_t)B * G * A + 15) / 16) * 16;
    t.idx = (int*)w; w += (size_t)B * A * 4;
    t.fg = (int*)w; w += (size_t)B * A * 4;
    t.gmax = (float*)w;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(tal_topk_kernel, dim3(B * G), dim3(256), 0, s, t);
    hipLaunchKernelGGL(tal_resolve_kernel, dim3(et_cdiv((long long)B * A, 256)), dim3(256), 0, s, t);
    hipLaunchKernelGGL(tal_gtmax_kernel, dim3(B * G), dim3(256), 0, s, t);
    hipLaunchKernelGGL(tal_emit_kernel, dim3(et_cdiv((long long)B * A, 256)), dim3(256), 0, s, t, (long long*)target_labels, target_bboxes,
                       target_scores, fg_mask);
    ET_CHECK_LAUNCH();
    return 0;
}


// ---- ComputeTalLoss: forward + gradient in one pass per anchor ---------------------------------------------------------
// reference models/loss/tal_loss.py:51-146 after the assigner (:95-102).  The two loss classes it imports
// (models.loss.gfocal_loss.VarifocalLoss / BboxLoss) are absent from the reference tree; their definitions are the
// written spec of oracle/v8.py::tal_loss (parity unpinned for this function, see there):
//     loss_cls = sum BCEWithLogits(pred_scores, target_scores) / S                      S = max(sum target_scores, 1)
//     loss_iou = sum_fg (1 - GIoU(pred_box, target_box)) * w / S                         w = sum_c target_scores[a, c]
//     loss_dfl = sum_fg mean_side( CE(l, floor t) (ceil t - t) + CE(l, ceil t) (t - floor t) ) * w / S,  t = clip(dist, 0, reg_max - .01)
//     loss     = w_class loss_cls + w_iou loss_iou + w_dfl loss_dfl
// pred_box = anchor -/+ DFL expectation of the side logits (grid units).  One thread per (image, anchor) computes its terms and
// writes d loss / d pred_scores, d loss / d pred_distri; S comes from tal_sum_kernel (device scalar, no host sync).
struct TalLossArgs {
    const float* ps;        // (B, A, nc) class logits
    const float* pd;        // (B, A, 4*(reg_max+1)) DFL logits
    const float* anc_s;     // (A, 2) anchor points in grid units
    const float* stride;    // (A)
    const float* tb;        // (B, A, 4) target boxes in PIXELS (assigner output)
    const float* ts;        // (B, A, nc) target scores
    const unsigned char* fg;
    int B, A, nc, reg_max, iou_kind;
    float w_class, w_iou, w_dfl;
    float* gps;             // gradients, same shapes
    float* gpd;
    float* acc;             // [0] sum target scores  [1] cls  [2] iou  [3] dfl  (fp32 atomics)
};

__global__ __launch_bounds__(256) void tal_sum_kernel(const float* __restrict__ ts, long long n, float* __restrict__ acc) {
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) s += ts[i];
    s = et_wave_sum(s);
    if ((threadIdx.x & 63) == 0) atomicAdd(acc, s);
}

__global__ __launch_bounds__(256) void tal_loss_kernel(TalLossArgs t) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    float l_cls = 0.f, l_iou = 0.f, l_dfl = 0.f;
    if (i < (long long)t.B * t.A) {
        const int a = (int)(i % t.A);
        const float S = fmaxf(t.acc[0], 1.0f);
        const float* x = t.ps + i * t.nc;
        const float* y = t.ts + i * t.nc;
        float* gx = t.gps + i * t.nc;
        float w = 0.f;
        for (int c = 0; c < t.nc; ++c) {
            const float xv = x[c], yv = y[c];
            // BCEWithLogits: max(x,0) - x*y + log(1 + exp(-|x|))
            l_cls += fmaxf(xv, 0.f) - xv * yv + log1pf(expf(-fabsf(xv)));
            gx[c] = t.w_class * (et_sigmoid(xv) - yv) / S;
            w += yv;
        }
        const int nb = t.reg_max + 1;
        const float* l = t.pd + i * 4 * nb;
        float* gl = t.gpd + i * 4 * nb;
        if (!t.fg[i]) {
            for (int k = 0; k < 4 * nb; ++k) gl[k] = 0.f;
        } else {
            const float ax = t.anc_s[2 * a], ay = t.anc_s[2 * a + 1], st = t.stride[a];
            float d[4], mx[4], den[4];
            for (int s = 0; s < 4; ++s) {
                float m = -INFINITY;
                for (int k = 0; k < nb; ++k) m = fmaxf(m, l[s * nb + k]);
                float dn = 0.f, nm = 0.f;
                for (int k = 0; k < nb; ++k) { const float e = expf(l[s * nb + k] - m); dn += e; nm += e * (float)k; }
                d[s] = nm / dn; mx[s] = m; den[s] = dn;
            }
            const float x1 = ax - d[0], y1 = ay - d[1], x2 = ax + d[2], y2 = ay + d[3];
            const float X1 = t.tb[i * 4] / st, Y1 = t.tb[i * 4 + 1] / st, X2 = t.tb[i * 4 + 2] / st, Y2 = t.tb[i * 4 + 3] / st;
            const float eps = 1e-7f;
            const float w1 = x2 - x1, h1 = y2 - y1 + eps, w2 = X2 - X1, h2 = Y2 - Y1 + eps;
            const float iw_ = fminf(x2, X2) - fmaxf(x1, X1), ih_ = fminf(y2, Y2) - fmaxf(y1, Y1);
            const float iw = fmaxf(iw_, 0.f), ih = fmaxf(ih_, 0.f);
            const float inter = iw * ih;
            const float uni = w1 * h1 + w2 * h2 - inter + eps;
            const float iou = inter / uni;
            // d inter / d (x1, y1, x2, y2)
            const float di[4] = {(iw_ > 0.f && x1 > X1) ? -ih : 0.f, (ih_ > 0.f && y1 > Y1) ? -iw : 0.f,
                                 (iw_ > 0.f && x2 < X2) ? ih : 0.f, (ih_ > 0.f && y2 < Y2) ? iw : 0.f};
            const float da[4] = {-h1, -w1, h1, w1};                 // d area1
            float giou = iou, dg[4];
            for (int k = 0; k < 4; ++k) {
                const float du = da[k] - di[k];
                dg[k] = (di[k] * uni - inter * du) / (uni * uni);
            }
            if (t.iou_kind == 1) {                                   // GIoU
                const float cw = fmaxf(x2, X2) - fminf(x1, X1), ch = fmaxf(y2, Y2) - fminf(y1, Y1);
                const float C = cw * ch + eps;
                giou = iou - (C - uni) / C;
                const float dc[4] = {(x1 < X1) ? -ch : 0.f, (y1 < Y1) ? -cw : 0.f, (x2 > X2) ? ch : 0.f, (y2 > Y2) ? cw : 0.f};
                for (int k = 0; k < 4; ++k) {
                    const float du = da[k] - di[k];
                    dg[k] += (du * C - uni * dc[k]) / (C * C);
                }
            }
            l_iou = (1.0f - giou) * w;
            // box coordinate k depends on side k: x1 = ax - d0, y1 = ay - d1, x2 = ax + d2, y2 = ay + d3
            const float sgn[4] = {-1.f, -1.f, 1.f, 1.f};
            const float tdist[4] = {ax - X1, ay - Y1, X2 - ax, Y2 - ay};
            for (int s = 0; s < 4; ++s) {
                const float dL_dd = t.w_iou * (-dg[s]) * sgn[s] * w / S;          // d(w_iou * loss_iou) / d d_s
                const float tt = fminf(fmaxf(tdist[s], 0.f), (float)t.reg_max - 0.01f);
                const int tl = (int)tt, tr = tl + 1;
                const float wl = (float)tr - tt, wr = tt - (float)tl;
                const float lse = mx[s] + logf(den[s]);
                l_dfl += ((lse - l[s * nb + tl]) * wl + (lse - l[s * nb + tr]) * wr) * 0.25f * w;
                const float cd = t.w_dfl * 0.25f * w / S;
                for (int k = 0; k < nb; ++k) {
                    const float p = expf(l[s * nb + k] - mx[s]) / den[s];
                    float gk = dL_dd * p * ((float)k - d[s]);                     // through the expectation
                    gk += cd * (p - (k == tl ? wl : 0.f) - (k == tr ? wr : 0.f)); // DFL cross-entropy
                    gl[s * nb + k] = gk;
                }
            }
        }
    }
    l_cls = et_wave_sum(l_cls); l_iou = et_wave_sum(l_iou); l_dfl = et_wave_sum(l_dfl);
    if ((threadIdx.x & 63) == 0) { atomicAdd(t.acc + 1, l_cls); atomicAdd(t.acc + 2, l_iou); atomicAdd(t.acc + 3, l_dfl); }
}

__global__ void tal_loss_finalize_kernel(const float* __restrict__ acc, float w_class, float w_iou, float w_dfl, float* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const float S = fmaxf(acc[0], 1.0f);
        out[0] = w_iou * acc[2] / S;
        out[1] = w_dfl * acc[3] / S;
        out[2] = w_class * acc[1] / S;
        out[3] = out[0] + out[1] + out[2];
    }
}

extern "C" int et_tal_loss(const float* pred_scores, const float* pred_distri, const float* anchor_points_s, const float* stride_tensor,
                           const float* target_bboxes_px, const float* target_scores, const uint8_t* fg_mask, int B, int A, int nc,
                           int reg_max, int iou_kind, float w_class, float w_iou, float w_dfl, float* grad_scores, float* grad_distri,
                           float* acc_ws /* 4 floats, ZERO on entry */, float* out /* 4 floats */, et_stream_t stream) {
    if (!pred_scores || !pred_distri || !anchor_points_s || !stride_tensor || !target_bboxes_px || !target_scores || !fg_mask ||
        !grad_scores || !grad_distri || !acc_ws || !out) return -1;
    if (B <= 0 || A <= 0 || nc <= 0 || reg_max < 1 || reg_max > 31 || (iou_kind != 0 && iou_kind != 1)) return -2;
    TalLossArgs t;
    t.ps = pred_scores; t.pd = pred_distri; t.anc_s = anchor_points_s; t.stride = stride_tensor; t.tb = target_bboxes_px;
    t.ts = target_scores; t.fg = fg_mask; t.B = B; t.A = A; t.nc = nc; t.reg_max = reg_max; t.iou_kind = iou_kind;
    t.w_class = w_class; t.w_iou = w_iou; t.w_dfl = w_dfl; t.gps = grad_scores; t.gpd = grad_distri; t.acc = acc_ws;
    hipStream_t s = (hipStream_t)stream;
    const long long n = (long long)B * A * nc;
    hipLaunchKernelGGL(tal_sum_kernel, dim3((int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024)), dim3(256), 0, s, target_scores, n, acc_ws);
    hipLaunchKernelGGL(tal_loss_kernel, dim3(et_cdiv((long long)B * A, 256)), dim3(256), 0, s, t);
    hipLaunchKernelGGL(tal_loss_finalize_kernel, dim3(1), dim3(64), 0, s, acc_ws, w_class, w_iou, w_dfl, out);
    ET_CHECK_LAUNCH();
    return 0;
}


// ---- pseudo labels on the anchor-free head (EXTENSION: the reference has no TAL variant of ComputeStudentMatchLoss) ----------------
// Specification: oracle/v8.py::tal_student_match_loss (parity unpinned by construction; DESIGN.md "a-14").  It mirrors
// ComputeStudentMatchLoss (models/loss/ssod/ssod_loss.py:130-296) on the YOLOv8 head:
//   split   (:130-192)  reliable  = conf >= high[cls]                      -> ordinary TaskAlignedAssigner targets
//                       uncertain = low[cls] <= conf < high[cls]           -> soft class target: aligned score * s,
//                                   s = obj_conf (pseudo_label_with_obj) or conf; box / DFL terms only for uncertain labels with
//                                   obj_conf >= 0.99 (pseudo_label_with_bbox), full-strength class target for cls_conf >= 0.99
//                                   (pseudo_label_with_cls)
//   merge   (:231,:248) where a reliable and an uncertain label both own an anchor the UNCERTAIN one wins (the reference writes
//                       tobj for the reliable cells first and for the uncertain cells afterwards)
// The padded pseudo-label table (B * G rows of 9 fp64 + valid mask, utils/self_supervised_utils.py) never leaves the device.
__global__ __launch_bounds__(256) void tal_pseudo_split_kernel(const double* __restrict__ t9, const unsigned char* __restrict__ valid,
                                                               const double* __restrict__ lo, const double* __restrict__ hi, int n, int nc,
                                                               int with_obj, int with_bbox, int with_cls, float img_w, float img_h,
                                                               float* __restrict__ glabel_r, float* __restrict__ gbox_r,
                                                               float* __restrict__ glabel_u, float* __restrict__ gbox_u,
                                                               float* __restrict__ mask_r, float* __restrict__ mask_u,
                                                               float* __restrict__ u_score, unsigned char* __restrict__ u_flags) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double* r = t9 + (size_t)i * 9;
    const bool ok = valid == nullptr || valid[i] != 0;
    int c = ok ? (int)r[1] : 0;
    c = c < 0 ? 0 : (c >= nc ? nc - 1 : c);
    const double conf = r[6], oc = r[7], cc = r[8];
    const bool rel = ok && conf >= hi[c];
    const bool unc = ok && !rel && conf >= lo[c];
    // normalised xywh -> xyxy pixels, the arithmetic of ComputeTalLoss.preprocess (tal_loss.py:139-142) in fp32.  Each set gets its
    // OWN table: a row that is not in the set is a padded row (label -1, zero box) exactly as in the supervised loss -- the assigner's
    // multi-owner resolution takes the arg-max IoU over ALL rows of the table it is given (tal_assigner.py:100-104)
    const float x = (float)r[2] * img_w, y = (float)r[3] * img_h, w = (float)r[4] * img_w, h = (float)r[5] * img_h;
    const float x1 = x - w * 0.5f, y1 = y - h * 0.5f;
    const float bx[4] = {x1, y1, x1 + w, y1 + h};
    glabel_r[i] = rel ? (float)c : -1.0f;
    glabel_u[i] = unc ? (float)c : -1.0f;
    for (int k = 0; k < 4; ++k) { gbox_r[i * 4 + k] = rel ? bx[k] : 0.f; gbox_u[i * 4 + k] = unc ? bx[k] : 0.f; }
    mask_r[i] = rel ? 1.f : 0.f;
    mask_u[i] = unc ? 1.f : 0.f;
    u_score[i] = unc ? (float)(with_obj ? oc : conf) : 0.f;
    unsigned char f = 0;
    if (unc && with_obj) {                                    // the reference forms the two subsets only under pseudo_label_with_obj
        if (with_bbox && oc >= 0.99) f |= 1;
        if (with_cls && cc >= 0.99) f |= 2;
    }
    u_flags[i] = f;
}

__global__ __launch_bounds__(256) void tal_merge_pseudo_kernel(const float* __restrict__ ts_r, const float* __restrict__ tb_r,
                                                               const unsigned char* __restrict__ fg_r, const float* __restrict__ ts_u,
                                                               const float* __restrict__ tb_u, const unsigned char* __restrict__ fg_u,
                                                               const int* __restrict__ idx_u, const float* __restrict__ u_score,
                                                               const unsigned char* __restrict__ u_flags, int B, int A, int G, int nc,
                                                               float* __restrict__ ts, float* __restrict__ tb,
                                                               unsigned char* __restrict__ fg_box) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)B * A) return;
    const int b = (int)(i / A);
    const bool u = fg_u[i] != 0, r = fg_r[i] != 0;
    float scale = 0.f;
    bool box = false;
    const float* src_s = ts_r + i * nc;
    const float* src_b = tb_r + i * 4;
    if (u) {
        const int g = idx_u[i];
        const unsigned char f = u_flags[(size_t)b * G + g];
        scale = (f & 2) ? 1.0f : u_score[(size_t)b * G + g];
        box = (f & 1) != 0;
        src_s = ts_u + i * nc; src_b = tb_u + i * 4;
    } else if (r) {
        scale = 1.0f; box = true;
    }
    for (int k = 0; k < nc; ++k) ts[i * nc + k] = (u || r) ? src_s[k] * scale : 0.f;
    for (int k = 0; k < 4; ++k) tb[i * 4 + k] = src_b[k];
    fg_box[i] = box ? 1 : 0;
}

__global__ __launch_bounds__(256) void copy_i32_kernel(const int* __restrict__ src, int* __restrict__ dst, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

extern "C" int et_tal_pseudo_split(const double* targets9, const uint8_t* valid, const double* thr_low, const double* thr_high, int B,
                                   int G, int nc, int with_obj, int with_bbox, int with_cls, float img_w, float img_h, float* gt_labels_r,
                                   float* gt_bboxes_r, float* gt_labels_u, float* gt_bboxes_u, float* mask_reliable,
                                   float* mask_uncertain, float* u_score, uint8_t* u_flags, et_stream_t stream) {
    if (!targets9 || !thr_low || !thr_high || !gt_labels_r || !gt_bboxes_r || !gt_labels_u || !gt_bboxes_u || !mask_reliable ||
        !mask_uncertain || !u_score || !u_flags) return -1;
    if (B <= 0 || G <= 0 || nc <= 0) return -2;
    const int n = B * G;
    hipLaunchKernelGGL(tal_pseudo_split_kernel, dim3(et_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, targets9, valid, thr_low, thr_high,
                       n, nc, with_obj, with_bbox, with_cls, img_w, img_h, gt_labels_r, gt_bboxes_r, gt_labels_u, gt_bboxes_u, mask_reliable,
                       mask_uncertain, u_score, u_flags);
    ET_CHECK_LAUNCH();
    return 0;
}

extern "C" int et_tal_assigned_gt(const void* workspace, int B, int A, int G, int32_t* gt_idx, et_stream_t stream) {
    // the (B, A) gt index et_tal_assign left in its workspace (valid where its fg_mask is set)
    if (!workspace || !gt_idx) return -1;
    if (B <= 0 || A <= 0 || G <= 0) return -2;
    const char* w = (const char*)workspace + (((size_t)B * G * A + 15) / 16) * 16;
    hipLaunchKernelGGL(copy_i32_kernel, dim3(et_cdiv((long long)B * A, 256)), dim3(256), 0, (hipStream_t)stream, (const int*)w, gt_idx,
                       (long long)B * A);
    ET_CHECK_LAUNCH();
    return 0;
}

extern "C" int et_tal_merge_pseudo(const float* ts_r, const float* tb_r, const uint8_t* fg_r, const float* ts_u, const float* tb_u,
                                   const uint8_t* fg_u, const int32_t* gt_idx_u, const float* u_score, const uint8_t* u_flags, int B, int A,
                                   int G, int nc, float* target_scores, float* target_bboxes, uint8_t* fg_box, et_stream_t stream) {
    if (!ts_r || !tb_r || !fg_r || !ts_u || !tb_u || !fg_u || !gt_idx_u || !u_score || !u_flags || !target_scores || !target_bboxes || !fg_box)
        return -1;
    if (B <= 0 || A <= 0 || G <= 0 || nc <= 0) return -2;
    hipLaunchKernelGGL(tal_merge_pseudo_kernel, dim3(et_cdiv((long long)B * A, 256)), dim3(256), 0, (hipStream_t)stream, ts_r, tb_r, fg_r, ts_u,
                       tb_u, fg_u, gt_idx_u, u_score, u_flags, B, A, G, nc, target_scores, target_bboxes, fg_box);
    ET_CHECK_LAUNCH();
    return 0;
}


// ---- ComputeTalLoss.preprocess on the device (models/loss/tal_loss.py:131-143) ------------------------------------------------
// (n, 6) rows [img, cls, x, y, w, h] (normalised) -> per-image padded table (B, G, 5) [cls, x1, y1, x2, y2] in pixels + mask (B, G).
// The reference builds it with a python loop over targets.cpu() (a host synchronisation per step when the targets live on the
// device); here G = n (the capacity that can never overflow -- padded rows cost the assigner one early-exiting workgroup each) and
// a row's slot inside its image is the number of earlier rows of the same image (O(n^2) over at most a few thousand rows).
__global__ __launch_bounds__(256) void tal_targets_pad_kernel(const float* __restrict__ t, int n, int B, int G, float img_w, float img_h,
                                                              float* __restrict__ out, float* __restrict__ mask) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < B * G) {                                              // initialise every slot: label -1, zero box, mask 0
        out[(size_t)i * 5] = -1.f;
        for (int k = 1; k < 5; ++k) out[(size_t)i * 5 + k] = 0.f;
        mask[i] = 0.f;
    }
}
__global__ __launch_bounds__(256) void tal_targets_scatter_kernel(const float* __restrict__ t, int n, int B, int G, float img_w, float img_h,
                                                                  float* __restrict__ out, float* __restrict__ mask) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int img = (int)t[(size_t)i * 6];
    if (img < 0 || img >= B) return;
    int rank = 0;
    for (int j = 0; j < i; ++j) rank += ((int)t[(size_t)j * 6] == img) ? 1 : 0;
    if (rank >= G) return;
    const float x = t[(size_t)i * 6 + 2] * img_w, y = t[(size_t)i * 6 + 3] * img_h, w = t[(size_t)i * 6 + 4] * img_w, h = t[(size_t)i * 6 + 5] * img_h;
    const float x1 = x - w * 0.5f, y1 = y - h * 0.5f;
    float* o = out + ((size_t)img * G + rank) * 5;
    o[0] = t[(size_t)i * 6 + 1]; o[1] = x1; o[2] = y1; o[3] = x1 + w; o[4] = y1 + h;
    // mask_gt = (gt_bboxes.sum(-1) > 0) (tal_loss.py:84)
    mask[(size_t)img * G + rank] = (x1 + y1 + (x1 + w) + (y1 + h)) > 0.f ? 1.f : 0.f;
}

extern "C" int et_tal_targets_pad(const float* targets, int n, int B, int G, float img_w, float img_h, float* out /* (B,G,5) */,
                                  float* mask /* (B,G) */, et_stream_t stream) {
    if (!out || !mask || (n > 0 && !targets)) return -1;
    if (n < 0 || B <= 0 || G <= 0) return -2;
    hipLaunchKernelGGL(tal_targets_pad_kernel, dim3(et_cdiv(B * G, 256)), dim3(256), 0, (hipStream_t)stream, targets, n, B, G, img_w, img_h, out, mask);
    if (n > 0)
        hipLaunchKernelGGL(tal_targets_scatter_kernel, dim3(et_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, targets, n, B, G, img_w, img_h,
                           out, mask);
    ET_CHECK_LAUNCH();
    return 0;
}
