// Anchor assignment + detection losses, forward AND gradient in the same pass, no host round trips.
//
// Replaces (reference file:line):
//   YOLOAnchorAssigner.build_targets / build_uc_targets_aug   models/assigner/yolo_anchor_assigner.py:319-372, 640-696
//   bbox_iou(x1y1x2y2=False, CIoU=True)                       utils/metrics.py:207-245
//   ComputeLoss.default_loss                                  models/loss/loss.py:138-208
//   ComputeStudentMatchLoss.select_targets / default_loss     models/loss/ssod/ssod_loss.py:130-296
//
// Work decomposition (per pyramid level, all levels/passes resolved inside one launch each):
//   count  : one thread per candidate slot (pass, offset o, anchor a, target k); counts the positives
//            of each (pass, level) -- the denominators of the reference's .mean() calls.
//   poscls : same slots; for each positive gathers the 5+nc logits of its cell, evaluates CIoU and the
//            class BCE, accumulates the loss sums and atomically adds d(loss)/d(logits) into dp, and
//            publishes tobj with a 64-bit atomicMax keyed by the slot's sequence number, which
//            reproduces the reference's duplicate-cell rule (last writer in assigner order wins; the
//            uncertain pass writes after the reliable pass) deterministically.
//   obj    : dense over every cell (b, a, y, x): BCE(logit_4, tobj) and its gradient.
// Slot order == reference output order: offset-major [centre, left, up, right, down], then anchor,
// then target.  Compiled with -ffp-contract=off.
#include "et_device.h"
#include "../../include/et_hip.h"

#define LOSS_NPASS 4   // 0 reliable (box+cls+tobj=iou) 1 uncertain (tobj=score) 2 uc_obj (box) 3 uc_cls (cls)
#define LOSS_MAXL 4

struct LossLevel {
    const void* p;         // logits, element strides below, channel stride 1
    float* dp;             // fp32 gradient, same strides
    unsigned long long* tobj;   // [B][na][ny][nx] packed (seq << 32 | float bits); 0 = untouched
    long long sb, sa, sy, sx;
    int ny, nx;
    float anchors[3][2];   // stride-normalised (na <= 3)
    float balance;
};

struct LossArgs {
    int dtype, B, na, nc, no, NT, nl;
    float anchor_t, gr, cp, cn, cls_pw, obj_pw, box_w, obj_w, cls_w;
    int pass_mask;         // bit p set: pass p enabled
    int ignore_obj;
    float* balance_dev;    // NULL, or [nl] objectness balance weights in device memory (Loss.autobalance: they change every step)
    int ssi;               // autobalance: index of the stride-16 level the weights are renormalised by; -1 = fixed weights
    float fl_gamma;        // > 0: FocalLoss(BCE, gamma, alpha 0.25) around the class and objectness terms (loss.py:37-62, :112-114)
    int obj_ch;            // channel of the objectness logit: 4, or no-1 in the SimOTA half (loss.py:246)
    const int* ota_match;  // NULL, or [nl][5*na*NT]: pass-0 slot -> index of the matched target, -1 = not a positive
    const float* tgt;      // [NT][8]: img, cls, x, y, w, h, score, flags(bit p = member of pass p)
    float* acc;            // [nl][16] accumulators, see ACC_* ; zeroed by the caller (et_yolo_loss does it)
};

enum { ACC_CNT0 = 0, ACC_BOX0 = 4, ACC_CLS0 = 8, ACC_OBJ = 12, ACC_OBJN = 13 };
// acc[l][ACC_CNT0+p] positives of pass p; acc[l][ACC_BOX0+p] sum(1-ciou); acc[l][ACC_CLS0+p] sum BCE cls;
// acc[l][ACC_OBJ] sum BCE obj; acc[l][ACC_OBJN] number of non-ignored cells.

struct Slot {
    bool valid;
    int b, c, a, gj, gi;
    float tb[4];       // dx, dy, gw, gh
    float score;
    unsigned seq;
};

__device__ __forceinline__ float ld_logit(const void* p, int dtype, long long off) {
    return dtype == ET_F32 ? ((const float*)p)[off] : (dtype == ET_F16 ? et_h2f(((const uint16_t*)p)[off]) : et_bf2f(((const uint16_t*)p)[off]));
}

// candidate slot s of pass `pass` on this level -> assignment result
__device__ __forceinline__ Slot eval_slot(const LossArgs& A, const LossLevel& L, int pass, int s, int level = 0) {
    Slot r;
    r.valid = false;
    const int k = s % A.NT;
    const int a = (s / A.NT) % A.na;
    const int o = s / (A.NT * A.na);
    const float* t = A.tgt + (size_t)k * 8;
    const int flags = (int)t[7];
    if (!((flags >> pass) & 1)) return r;
    const float nx = (float)L.nx, ny = (float)L.ny;
    const float gx = t[2] * nx, gy = t[3] * ny, gw = t[4] * nx, gh = t[5] * ny;   // t = targets * gain
    const float rw = gw / L.anchors[a][0], rh = gh / L.anchors[a][1];
    const float m = fmaxf(fmaxf(rw, 1.0f / rw), fmaxf(rh, 1.0f / rh));
    if (!(m < A.anchor_t)) return r;
    float ox = 0.f, oy = 0.f;
    if (o != 0) {
        const float gxi = nx - gx, gyi = ny - gy;
        bool ok;
        if (o == 1) { ok = ((gx - floorf(gx)) < 0.5f) && (gx > 1.0f); ox = 0.5f; }
        else if (o == 2) { ok = ((gy - floorf(gy)) < 0.5f) && (gy > 1.0f); oy = 0.5f; }
        else if (o == 3) { ok = ((gxi - floorf(gxi)) < 0.5f) && (gxi > 1.0f); ox = -0.5f; }
        else { ok = ((gyi - floorf(gyi)) < 0.5f) && (gyi > 1.0f); oy = -0.5f; }
        if (!ok) return r;
    }
    int gi = (int)(gx - ox), gj = (int)(gy - oy);          // .long(): truncation toward zero
    gi = min(max(gi, 0), L.nx - 1);
    gj = min(max(gj, 0), L.ny - 1);
    r.valid = true;
    r.b = (int)t[0]; r.c = (int)t[1]; r.a = a; r.gi = gi; r.gj = gj;
    r.tb[0] = gx - (float)gi; r.tb[1] = gy - (float)gj; r.tb[2] = gw; r.tb[3] = gh;
    r.score = t[6];
    r.seq = (unsigned)s + 1u + (pass == 1 ? (1u << 27) : 0u);
    if (pass == 0 && A.ota_match) {
        // SimOTA (ComputeLoss.ota_loss, loss.py:218-244): the slot keeps its cell and anchor, box and class come from the
        // target the dynamic-k matching gave it: selected_tbox = ota_targets[:, 2:6] * gain ; selected_tbox[:, :2] -= grid
        const int g = A.ota_match[(size_t)level * (5 * A.na * A.NT) + s];
        if (g < 0) { r.valid = false; return r; }
        const float* tg = A.tgt + (size_t)g * 8;
        r.c = (int)tg[1];
        r.tb[0] = tg[2] * nx - (float)gi; r.tb[1] = tg[3] * ny - (float)gj; r.tb[2] = tg[4] * nx; r.tb[3] = tg[5] * ny;
    }
    return r;
}

// r06: the pyramid levels of one loss run in ONE launch per kernel (blockIdx.z = level; they are independent of each other, count ->
// pos -> obj is the only order that matters): 8 launches per loss instead of 14 in the low-power gap between the step's forward and
// backward (profiles/r06_loss_levels_one_launch_ab.txt)
struct LossLevels { LossLevel l[LOSS_MAXL]; float fixed_n[LOSS_MAXL]; };

__global__ __launch_bounds__(256) void loss_count_kernel(LossArgs A, LossLevels Ls) {
    const int level = blockIdx.z;
    const LossLevel& L = Ls.l[level];
    const int nslot = 5 * A.na * A.NT;
    const int s = blockIdx.x * 256 + threadIdx.x;
    const int pass = blockIdx.y;
    int v = 0;
    if (((A.pass_mask >> pass) & 1) && s < nslot) v = eval_slot(A, L, pass, s, level).valid ? 1 : 0;
    v = et_wave_sum_i(v);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(A.acc + level * 16 + ACC_CNT0 + pass, (float)v);
}

// BCEWithLogits (pos_weight pw): value and d/dx
__device__ __forceinline__ float bce_logits(float x, float t, float pw, float& grad) {
    const float lw = 1.0f + (pw - 1.0f) * t;
    const float sp = log1pf(expf(-fabsf(x))) + fmaxf(-x, 0.0f);     // softplus(-x)
    const float sg = 1.0f / (1.0f + expf(-x));
    grad = (1.0f - t) + lw * (sg - 1.0f);
    return (1.0f - t) * x + lw * sp;
}

// FocalLoss wrapper of the reference (loss.py:46-62, TF formulation): bce * alpha_t * (1 - p_t)^gamma, alpha 0.25.
__device__ __forceinline__ float focal_bce_logits(float x, float t, float pw, float gamma, float& grad) {
    float gb;
    const float b = bce_logits(x, t, pw, gb);
    if (!(gamma > 0.f)) { grad = gb; return b; }
    const float p = 1.0f / (1.0f + expf(-x));
    const float pt = t * p + (1.0f - t) * (1.0f - p);
    const float af = t * 0.25f + (1.0f - t) * 0.75f;
    const float om = 1.0f - pt;
    const float mf = powf(om, gamma);
    // d mf / dx = -gamma (1 - p_t)^(gamma-1) * d p_t / dx ,  d p_t / dx = (2t - 1) p (1 - p)
    const float dmf = om > 0.f ? -gamma * powf(om, gamma - 1.0f) * (2.0f * t - 1.0f) * p * (1.0f - p) : 0.f;
    grad = af * (gb * mf + b * dmf);
    return b * af * mf;
}

__device__ __forceinline__ float min_grad_a(float a, float b) { return a < b ? 1.f : (a == b ? 0.5f : 0.f); }
__device__ __forceinline__ float max_grad_a(float a, float b) { return a > b ? 1.f : (a == b ? 0.5f : 0.f); }

// CIoU of pbox (xywh, from logits) vs tbox (xywh); returns ciou and d ciou / d (px,py,pw,ph)
__device__ __forceinline__ float ciou_fwd_bwd(const float pb[4], const float tb[4], float g[4]) {
    const float eps = 1e-7f;
    const float x1 = pb[0] - pb[2] / 2, x2 = pb[0] + pb[2] / 2, y1 = pb[1] - pb[3] / 2, y2 = pb[1] + pb[3] / 2;
    const float X1 = tb[0] - tb[2] / 2, X2 = tb[0] + tb[2] / 2, Y1 = tb[1] - tb[3] / 2, Y2 = tb[1] + tb[3] / 2;
    const float mx = fminf(x2, X2), Mx = fmaxf(x1, X1), my = fminf(y2, Y2), My = fmaxf(y1, Y1);
    const float iwr = mx - Mx, ihr = my - My;
    const float iw = fmaxf(iwr, 0.f), ih = fmaxf(ihr, 0.f);
    const float inter = iw * ih;
    const float w1 = x2 - x1, h1 = y2 - y1 + eps, w2 = X2 - X1, h2 = Y2 - Y1 + eps;
    const float uni = w1 * h1 + w2 * h2 - inter + eps;
    const float iou = inter / uni;
    const float cxx = fmaxf(x2, X2), cxn = fminf(x1, X1), cyx = fmaxf(y2, Y2), cyn = fminf(y1, Y1);
    const float cw = cxx - cxn, ch = cyx - cyn;
    const float c2 = cw * cw + ch * ch + eps;
    const float sx = X1 + X2 - x1 - x2, sy = Y1 + Y2 - y1 - y2;
    const float rho2 = (sx * sx + sy * sy) / 4;
    const float kpi = 0.40528473456935108578f;   // 4 / pi^2
    const float q1 = w1 / h1;
    const float dA = atanf(w2 / h2) - atanf(q1);
    const float v = kpi * (dA * dA);
    const float alpha = v / (v - iou + (1 + eps));
    const float ciou = iou - (rho2 / c2 + v * alpha);
    // reverse mode, upstream 1
    const float d_iou = 1.f, d_rho2 = -1.f / c2, d_c2 = rho2 / (c2 * c2), d_v = -alpha;
    float d_inter = d_iou / uni;
    const float d_uni = -d_iou * inter / (uni * uni);
    float d_w1 = d_uni * h1, d_h1 = d_uni * w1;
    d_inter -= d_uni;
    const float d_iw = iwr >= 0.f ? d_inter * ih : 0.f, d_ih = ihr >= 0.f ? d_inter * iw : 0.f;
    float d_x1 = 0.f, d_x2 = 0.f, d_y1 = 0.f, d_y2 = 0.f;
    d_x2 += d_iw * min_grad_a(x2, X2);  d_x1 -= d_iw * max_grad_a(x1, X1);
    d_y2 += d_ih * min_grad_a(y2, Y2);  d_y1 -= d_ih * max_grad_a(y1, Y1);
    const float d_cw = d_c2 * 2 * cw, d_ch = d_c2 * 2 * ch;
    d_x2 += d_cw * max_grad_a(x2, X2);  d_x1 -= d_cw * min_grad_a(x1, X1);
    d_y2 += d_ch * max_grad_a(y2, Y2);  d_y1 -= d_ch * min_grad_a(y1, Y1);
    const float d_sx = d_rho2 * sx / 2, d_sy = d_rho2 * sy / 2;
    d_x1 -= d_sx; d_x2 -= d_sx; d_y1 -= d_sy; d_y2 -= d_sy;
    const float d_A1 = -d_v * 2 * kpi * dA;
    const float d_q = d_A1 / (1 + q1 * q1);
    d_w1 += d_q / h1;
    d_h1 -= d_q * w1 / (h1 * h1);
    d_x2 += d_w1; d_x1 -= d_w1; d_y2 += d_h1; d_y1 -= d_h1;
    g[0] = d_x1 + d_x2; g[2] = (d_x2 - d_x1) / 2;
    g[1] = d_y1 + d_y2; g[3] = (d_y2 - d_y1) / 2;
    return ciou;
}

// Class term of the positive slots of one wave: the wave walks its slots that have one (do_cls) and spreads the nc class logits of
// each over the lanes (coalesced loads and atomics; one lane per slot looping over 80 classes touched 64 different cache lines per
// iteration).  EIGHT slots per trip: their logit loads are issued together (unconditionally; a missing slot re-reads the first
// one's row) before any of the focal / BCE arithmetic -- one slot per trip was a chain of up to 64 dependent L2 round trips per
// wave, most of loss_pos_kernel's 87 us on the bench batch (r04).  The element type is a template parameter: a run-time dtype
// switch inside the loop puts every load behind its own branch and wait.  Per-lane summation order: slots ascending, as before.
template <typename T> __device__ __forceinline__ float ld_logit_t(const void* p, long long off);
template <> __device__ __forceinline__ float ld_logit_t<float>(const void* p, long long off) { return ((const float*)p)[off]; }
template <> __device__ __forceinline__ float ld_logit_t<uint16_t>(const void* p, long long off) { return et_bf2f(((const uint16_t*)p)[off]); }
template <> __device__ __forceinline__ float ld_logit_t<et_f16>(const void* p, long long off) { return et_h2f(((const uint16_t*)p)[off]); }

template <typename T>
__device__ __forceinline__ float loss_class_walk(const LossArgs& A, const LossLevel& L, bool do_cls, long long cls_off, int cls_c,
                                                 float cls_wgt) {
    constexpr int NB = 8;                                    // slots per trip
    const int lane = threadIdx.x & 63;
    float cls_sum = 0.f;
    unsigned long long m = __ballot(do_cls);
    while (m) {
        long long off_s[NB];
        int c_s[NB];
        float wgt_s[NB];
        int n = 0;
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            int src = 0;
            if (m) { src = __ffsll((long long)m) - 1; m &= m - 1; n = k + 1; }
            const unsigned lo = __shfl((unsigned)(cls_off & 0xffffffffll), src);
            const unsigned hi = __shfl((unsigned)((unsigned long long)cls_off >> 32), src);
            off_s[k] = k < n ? (long long)(((unsigned long long)hi << 32) | lo) : off_s[0];
            c_s[k] = __shfl(cls_c, src);
            wgt_s[k] = __shfl(cls_wgt, src);
        }
        for (int c = lane; c < A.nc; c += 64) {
            float x[NB];
#pragma unroll
            for (int k = 0; k < NB; ++k) x[k] = ld_logit_t<T>(L.p, off_s[k] + 5 + c);
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                if (k < n) {                                     // wave-uniform
                    float gr_;
                    cls_sum += focal_bce_logits(x[k], c == c_s[k] ? A.cp : A.cn, A.cls_pw, A.fl_gamma, gr_);
                    atomicAdd(L.dp + off_s[k] + 5 + c, wgt_s[k] * gr_);
                }
            }
        }
    }
    return cls_sum;
}

__global__ __launch_bounds__(256) void loss_pos_kernel(LossArgs A, LossLevels Ls) {
    const int level = blockIdx.z;
    const LossLevel& L = Ls.l[level];
    const int nslot = 5 * A.na * A.NT;
    const int s = blockIdx.x * 256 + threadIdx.x;
    const int pass = blockIdx.y;
    float box_sum = 0.f, cls_sum = 0.f;
    bool do_cls = false;
    long long cls_off = 0;
    int cls_c = 0;
    float cls_wgt = 0.f;
    if (((A.pass_mask >> pass) & 1) && s < nslot) {
        const Slot r = eval_slot(A, L, pass, s, level);
        if (r.valid && r.b >= 0 && r.b < A.B) {
            const float npos = A.acc[level * 16 + ACC_CNT0 + pass];
            const long long off = r.b * L.sb + r.a * L.sa + r.gj * L.sy + r.gi * L.sx;
            const long long cell = (((long long)r.b * A.na + r.a) * L.ny + r.gj) * L.nx + r.gi;
            const bool want_box = pass == 0 || pass == 2;
            const bool want_cls = (pass == 0 || pass == 3) && A.nc > 1;
            if (want_box) {
                float lg[4], sg[4], pb[4], g[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) lg[i] = 0.f;
                if (A.dtype == ET_F32) {                  // the dtype switch OUTSIDE the loads: four loads in flight, one wait
#pragma unroll
                    for (int i = 0; i < 4; ++i) lg[i] = ld_logit_t<float>(L.p, off + i);
                } else if (A.dtype == ET_F16) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) lg[i] = ld_logit_t<et_f16>(L.p, off + i);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) lg[i] = ld_logit_t<uint16_t>(L.p, off + i);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) sg[i] = 1.0f / (1.0f + expf(-lg[i]));
                pb[0] = sg[0] * 2.f - 0.5f; pb[1] = sg[1] * 2.f - 0.5f;
                const float tw = sg[2] * 2, th = sg[3] * 2;
                pb[2] = tw * tw * L.anchors[r.a][0]; pb[3] = th * th * L.anchors[r.a][1];
                const float ciou = ciou_fwd_bwd(pb, r.tb, g);
                box_sum = 1.0f - ciou;
                // d lbox / d logits : lbox_l += mean(1 - ciou) ; total weight box_w / npos
                const float wgt = -A.box_w / npos;
                atomicAdd(L.dp + off + 0, wgt * g[0] * 2.f * sg[0] * (1.f - sg[0]));
                atomicAdd(L.dp + off + 1, wgt * g[1] * 2.f * sg[1] * (1.f - sg[1]));
                atomicAdd(L.dp + off + 2, wgt * g[2] * L.anchors[r.a][0] * 8.f * sg[2] * sg[2] * (1.f - sg[2]));
                atomicAdd(L.dp + off + 3, wgt * g[3] * L.anchors[r.a][1] * 8.f * sg[3] * sg[3] * (1.f - sg[3]));
                if (pass == 0) {
                    const float tv = (1.0f - A.gr) + A.gr * fmaxf(ciou, 0.f);
                    atomicMax(L.tobj + cell, ((unsigned long long)r.seq << 32) | __float_as_uint(tv));
                }
            }
            if (pass == 1) {
                const float tv = A.ignore_obj ? -1.0f : r.score;
                atomicMax(L.tobj + cell, ((unsigned long long)r.seq << 32) | __float_as_uint(tv));
            }
            if (want_cls) {
                do_cls = true;
                cls_off = off;
                cls_c = r.c;
                cls_wgt = A.cls_w / (npos * (float)A.nc);
            }
        }
    }
    // class term: the wave walks its slots that have one and spreads the nc class logits of each over the lanes
    if (A.dtype == ET_F32) cls_sum += loss_class_walk<float>(A, L, do_cls, cls_off, cls_c, cls_wgt);
    else if (A.dtype == ET_F16) cls_sum += loss_class_walk<et_f16>(A, L, do_cls, cls_off, cls_c, cls_wgt);
    else cls_sum += loss_class_walk<uint16_t>(A, L, do_cls, cls_off, cls_c, cls_wgt);
    box_sum = et_wave_sum(box_sum);
    cls_sum = et_wave_sum(cls_sum);
    if ((threadIdx.x & 63) == 0) {
        if (box_sum != 0.f) atomicAdd(A.acc + level * 16 + ACC_BOX0 + pass, box_sum);
        if (cls_sum != 0.f) atomicAdd(A.acc + level * 16 + ACC_CLS0 + pass, cls_sum);
    }
}

// mode 0: count non-ignored cells only; mode 1: loss sum + gradient (denominator read from acc)
__global__ __launch_bounds__(256) void loss_obj_kernel(LossArgs A, LossLevels Ls, int mode) {
    const int level = blockIdx.z;
    const LossLevel& L = Ls.l[level];
    const float fixed_n = Ls.fixed_n[level];
    // grid-stride over the cells, ONE atomic per workgroup: with a thread per cell and an atomic per wave the 80x80 level
    // sent 9600 adds to the same address (tools/probe/probe_stat_atomics: same-address atomics serialise at ~20-100 ns each)
    const unsigned ncell = (unsigned)A.B * A.na * L.ny * L.nx;          // host: < 2^31
    const unsigned plane = (unsigned)L.nx * L.ny;
    float lsum = 0.f, cnt = 0.f;
    const float n = (mode == 1) ? (fixed_n > 0.f ? fixed_n : A.acc[level * 16 + ACC_OBJN]) : 1.f;
    const float bal = (mode == 1) ? (A.balance_dev ? A.balance_dev[level] : L.balance) : 0.f;
    if (blockIdx.x * 256u >= ncell) return;            // the launch is sized for the largest level: workgroups beyond this level's cells
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < ncell; i += gridDim.x * 256u) {
        const unsigned long long w = L.tobj[i];
        const float t = w ? __uint_as_float((unsigned)(w & 0xffffffffull)) : 0.0f;
        if (t >= 0.0f) {
            cnt += 1.f;
            if (mode == 1) {
                const unsigned q = i / plane, rem = i - q * plane;       // q = b * na + a
                const unsigned gj = rem / (unsigned)L.nx, gi = rem - gj * (unsigned)L.nx;
                const unsigned b = q / (unsigned)A.na, a = q - b * (unsigned)A.na;
                const long long off = (long long)b * L.sb + (long long)a * L.sa + (long long)gj * L.sy + (long long)gi * L.sx + A.obj_ch;
                const float x = ld_logit(L.p, A.dtype, off);
                float g;
                lsum += focal_bce_logits(x, t, A.obj_pw, A.fl_gamma, g);
                L.dp[off] += g * (A.obj_w * bal / n);   // += : in the SimOTA half this channel is also a class logit
            }
        }
    }
    __shared__ float red[2][4];
    lsum = et_wave_sum(lsum);
    cnt = et_wave_sum(cnt);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = lsum; red[1][threadIdx.x >> 6] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (mode == 1) atomicAdd(A.acc + level * 16 + ACC_OBJ, red[0][0] + red[0][1] + red[0][2] + red[0][3]);
        else atomicAdd(A.acc + level * 16 + ACC_OBJN, red[1][0] + red[1][1] + red[1][2] + red[1][3]);
    }
}

// out[0..3] = lbox*box_w, lobj*obj_w, lcls*cls_w, (sum)*bs ; out[4+p] / out[8+p]: per-pass positives (level sums)
__global__ void loss_finalize_kernel(LossArgs A, float b0, float b1, float b2, float b3, long long n0, long long n1,
                                     long long n2, long long n3, float* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float bal[4] = {b0, b1, b2, b3};
    const long long ncell[4] = {n0, n1, n2, n3};
    float lbox = 0.f, lobj = 0.f, lcls = 0.f;
    for (int p = 0; p < 4; ++p) out[4 + p] = 0.f;
    for (int l = 0; l < A.nl; ++l) {
        const float* a = A.acc + l * 16;
        for (int p = 0; p < LOSS_NPASS; ++p) {
            const float n = a[ACC_CNT0 + p];
            out[4 + p] += n;
            if (n > 0.f) {
                if (p == 0 || p == 2) lbox += a[ACC_BOX0 + p] / n;
                if ((p == 0 || p == 3) && A.nc > 1) lcls += a[ACC_CLS0 + p] / (n * (float)A.nc);
            }
        }
        const float nobj = A.ignore_obj ? a[ACC_OBJN] : (float)ncell[l];
        const float obji = a[ACC_OBJ] / nobj;
        lobj += obji * (A.balance_dev ? A.balance_dev[l] : bal[l]);
        // Loss.autobalance (loss.py:193-194): balance[l] = balance[l] * 0.9999 + 0.0001 / obji, AFTER its use above
        if (A.balance_dev && A.ssi >= 0) A.balance_dev[l] = A.balance_dev[l] * 0.9999f + 0.0001f / obji;
    }
    if (A.balance_dev && A.ssi >= 0) {                  // loss.py:196-197
        const float ref = A.balance_dev[A.ssi];
        for (int l = 0; l < A.nl; ++l) A.balance_dev[l] = A.balance_dev[l] / ref;
    }
    lbox *= A.box_w; lobj *= A.obj_w; lcls *= A.cls_w;
    out[0] = lbox; out[1] = lobj; out[2] = lcls;
    out[3] = (lbox + lobj + lcls) * (float)A.B;
}

// ---- SimOTA matching (YOLOAnchorAssigner.build_ota_targets, models/assigner/yolo_anchor_assigner.py:104-264) ------------
// The reference loops over the images on the host; per image it gathers the candidates of find_3_positive (:266-317, the
// same slots as build_targets), builds the (gt x candidate) IoU and cost matrices and runs dynamic-k matching with
// torch.topk / .item() per gt.  Here, with no host round trip:
//   ota_order    stable order of the targets by image (the reference's `targets[:, 0] == batch_idx` subsets) + image starts
//   ota_cand     one thread per slot (level, offset, anchor, target): decoded box in pixels (:160-164), logits offset of
//                its cell, sum over the classes of the t=0 BCE term of the pair-wise class cost (:196-205)
//   ota_gt       one workgroup per gt: scans the slots of its image, every thread keeps its 13 largest IoUs and 13
//                cheapest (cost, slot) in registers; the workgroup merges them, dynamic_k = max(1, int(sum of the top-13
//                IoUs)) (:185-186) and marks the dynamic_k cheapest slots (:215-219)
//   ota_resolve  one thread per slot: unmarked -> -1, marked once -> that gt, marked by several gts -> argmin of the cost
//                over ALL gts of the image (:223-226)
// Quirks kept from the reference: gt boxes are scaled by a literal 640 (:128 "TODO"), the "objectness" factor of the class
// cost is the LAST channel of the prediction (p_obj_e2e = fg_pred[:, -1:], :157,199) -- with no = 5 + nc that is the logit
// of class nc-1.  Ties (equal costs / IoUs): torch.topk and torch.min leave the pick unspecified; here the smaller slot
// index (level-major, then reference candidate order) resp. the earlier gt wins.
#define OTA_K 13

struct OtaArgs {
    LossArgs A;
    LossLevel L[LOSS_MAXL];
    float stride[LOSS_MAXL];
    float img_size;
    int topk;
    int* order;        // [NT]   target indices grouped by image, input order inside an image; targets outside [0,B) last
    int* istart;       // [B+1]
    float* cbox;       // [nq][4] xyxy in pixels      nq = nl * 5 * na * NT
    float* cs0;        // [nq]
    long long* coff;   // [nq]   logits offset of the slot's cell
    int* cimg;         // [nq]   image of the slot, -1 = not a candidate
    int* cnt;          // [nq]
    int* who;          // [nq]
    int* match;        // [nq]   out
};

__device__ __forceinline__ int ota_key(const LossArgs& A, int k) {      // image of target k, or B for rows no image owns
    const float* t = A.tgt + (size_t)k * 8;
    const int b = (int)t[0];
    return ((((int)t[7]) & 1) && t[0] == (float)b && b >= 0 && b < A.B) ? b : A.B;
}

__global__ __launch_bounds__(256) void ota_order_kernel(OtaArgs O) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    const LossArgs& A = O.A;
    if (k < A.NT) {
        const int key = ota_key(A, k);
        int rank = 0;
        for (int j = 0; j < A.NT; ++j) {
            const int kj = ota_key(A, j);
            rank += (kj < key || (kj == key && j < k)) ? 1 : 0;
        }
        O.order[rank] = k;
    }
    if (k <= A.B) {
        int c = 0;
        for (int j = 0; j < A.NT; ++j) c += ota_key(A, j) < k ? 1 : 0;
        O.istart[k] = c;
    }
}

__global__ __launch_bounds__(256) void ota_cand_kernel(OtaArgs O) {
    const LossArgs& A = O.A;
    const int nslot = 5 * A.na * A.NT;
    const int s = blockIdx.x * 256 + threadIdx.x;
    const int level = blockIdx.y;
    const LossLevel& L = O.L[level];
    const size_t q = (size_t)level * nslot + s;
    bool live = false;
    long long off = 0;
    if (s < nslot) {
        const Slot r = eval_slot(A, L, 0, s);          // O.A.ota_match is NULL: the plain find_3_positive slot
        live = r.valid && r.b >= 0 && r.b < A.B && ota_key(A, s % A.NT) == r.b;
        O.cnt[q] = 0;
        O.who[q] = -1;
        O.cimg[q] = live ? r.b : -1;
        if (live) {
            off = r.b * L.sb + r.a * L.sa + r.gj * L.sy + r.gi * L.sx;
            float sg[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) sg[i] = 1.0f / (1.0f + expf(-ld_logit(L.p, A.dtype, off + i)));
            const float st = O.stride[level];
            const float px = (sg[0] * 2.f - 0.5f + (float)r.gi) * st, py = (sg[1] * 2.f - 0.5f + (float)r.gj) * st;
            const float tw = sg[2] * 2, th = sg[3] * 2;
            const float pw = tw * tw * L.anchors[r.a][0] * st, ph = th * th * L.anchors[r.a][1] * st;
            float* cb = O.cbox + q * 4;
            cb[0] = px - pw / 2; cb[1] = py - ph / 2; cb[2] = px + pw / 2; cb[3] = py + ph / 2;
            O.coff[q] = off;
        }
    }
    // class walk as in loss_pos_kernel: the wave spreads the nc logits of each live slot over its lanes
    const int lane = threadIdx.x & 63;
    unsigned long long m = __ballot(live);
    float mine = 0.f;
    while (m) {
        const int src = __ffsll((long long)m) - 1;
        m &= m - 1;
        const unsigned lo = __shfl((unsigned)(off & 0xffffffffll), src);
        const unsigned hi = __shfl((unsigned)((unsigned long long)off >> 32), src);
        const long long off_s = (long long)(((unsigned long long)hi << 32) | lo);
        const float se = 1.0f / (1.0f + expf(-ld_logit(L.p, A.dtype, off_s + A.no - 1)));
        float acc = 0.f;
        for (int c = lane; c < A.nc; c += 64) {
            const float sc = 1.0f / (1.0f + expf(-ld_logit(L.p, A.dtype, off_s + 5 + c)));
            const float y = sqrtf(sc * se);
            float g_;
            acc += bce_logits(logf(y / (1.0f - y)), 0.f, 1.f, g_);
        }
        acc = et_wave_sum(acc);
        if (lane == src) mine = acc;
    }
    if (live) O.cs0[q] = mine;
}

// IoU (utils/metrics.py:252-274 box_iou) and cost (:183, :196-211) of gt (box gb, class gc) against slot q
__device__ __forceinline__ float ota_pair(const OtaArgs& O, int level, size_t q, const float gb[4], int gc, float& iou) {
    const float* cb = O.cbox + q * 4;
    const float a1 = (gb[2] - gb[0]) * (gb[3] - gb[1]), a2 = (cb[2] - cb[0]) * (cb[3] - cb[1]);
    const float iw = fmaxf(fminf(gb[2], cb[2]) - fmaxf(gb[0], cb[0]), 0.f);
    const float ih = fmaxf(fminf(gb[3], cb[3]) - fmaxf(gb[1], cb[1]), 0.f);
    const float inter = iw * ih;
    iou = inter / (a1 + a2 - inter);
    const LossLevel& L = O.L[level];
    const long long off = O.coff[q];
    const float sc = 1.0f / (1.0f + expf(-ld_logit(L.p, O.A.dtype, off + 5 + gc)));
    const float se = 1.0f / (1.0f + expf(-ld_logit(L.p, O.A.dtype, off + O.A.no - 1)));
    const float y = sqrtf(sc * se);
    const float x = logf(y / (1.0f - y));
    float g_;
    const float cls = O.cs0[q] - bce_logits(x, 0.f, 1.f, g_) + bce_logits(x, 1.f, 1.f, g_);
    float cost = cls + 3.0f * (-logf(iou + 1e-8f));
    if (!(cost == cost)) cost = INFINITY;              // torch.topk orders NaN as the largest value
    return cost;
}

__device__ __forceinline__ void ota_gt_box(const OtaArgs& O, int g, float gb[4], int& gc) {
    const float* t = O.A.tgt + (size_t)g * 8;
    const float x = t[2] * O.img_size, y = t[3] * O.img_size, w = t[4] * O.img_size, h = t[5] * O.img_size;   // :128
    gb[0] = x - w / 2; gb[1] = y - h / 2; gb[2] = x + w / 2; gb[3] = y + h / 2;
    gc = min(max((int)t[1], 0), O.A.nc - 1);
}

// slot index of candidate number ci of image range [r0, r1): level-major, then slot order (offset, anchor, target)
__device__ __forceinline__ size_t ota_slot_of(const OtaArgs& O, int r0, int nimg, long long ci, int& level) {
    const int per_level = 5 * O.A.na * nimg;
    level = (int)(ci / per_level);
    const int rem = (int)(ci % per_level);
    const int oa = rem / nimg, j = rem % nimg;
    const int k = O.order[r0 + j];
    return (size_t)level * (5 * O.A.na * O.A.NT) + (size_t)oa * O.A.NT + k;
}

__global__ __launch_bounds__(256) void ota_gt_kernel(OtaArgs O) {
    __shared__ float sh_i[256 * OTA_K];
    __shared__ float sh_c[256 * OTA_K];
    __shared__ int sh_q[256 * OTA_K];
    __shared__ float rv[256];
    __shared__ int ri[256];
    __shared__ int rt[256];
    const LossArgs& A = O.A;
    const int g = O.order[blockIdx.x];
    const int b = ota_key(A, g);
    if (b >= A.B) return;
    const int r0 = O.istart[b], nimg = O.istart[b + 1] - r0;
    float gb[4];
    int gc;
    ota_gt_box(O, g, gb, gc);
    float ti[OTA_K], tc[OTA_K];
    int tq[OTA_K];
#pragma unroll
    for (int i = 0; i < OTA_K; ++i) { ti[i] = -1.f; tc[i] = INFINITY; tq[i] = 0x7fffffff; }
    const long long ncand = (long long)A.nl * 5 * A.na * nimg;
    const int tid = threadIdx.x;
    for (long long ci = tid; ci < ncand; ci += 256) {
        int level;
        const size_t q = ota_slot_of(O, r0, nimg, ci, level);
        if (O.cimg[q] != b) continue;
        float iou;
        float cost = ota_pair(O, level, q, gb, gc, iou);
        int qi = (int)q;
        float v = (iou == iou) ? iou : INFINITY;          // torch.topk: NaN sorts as the largest
#pragma unroll
        for (int i = 0; i < OTA_K; ++i) {
            if (v > ti[i]) { const float x = ti[i]; ti[i] = v; v = x; }
            if (cost < tc[i] || (cost == tc[i] && qi < tq[i])) {
                const float x = tc[i]; const int y = tq[i];
                tc[i] = cost; tq[i] = qi; cost = x; qi = y;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < OTA_K; ++i) { sh_i[tid * OTA_K + i] = ti[i]; sh_c[tid * OTA_K + i] = tc[i]; sh_q[tid * OTA_K + i] = tq[i]; }
    __syncthreads();
    // dynamic_k from the top-k IoUs of the whole workgroup
    int ptr = 0;
    float sum = 0.f;
    const int K = min(O.topk, OTA_K);
    for (int r = 0; r < K; ++r) {
        rv[tid] = ptr < OTA_K ? sh_i[tid * OTA_K + ptr] : -1.f;
        rt[tid] = tid;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if (tid < st && rv[tid + st] > rv[tid]) { rv[tid] = rv[tid + st]; rt[tid] = rt[tid + st]; }
            __syncthreads();
        }
        const float best = rv[0];
        const int wt = rt[0];
        __syncthreads();
        if (best < 0.f) break;                            // fewer than k candidates
        sum += best;
        if (tid == wt) ++ptr;
    }
    int dyn = (int)sum;                                    // .int() truncation; NaN/inf IoU sums are not meaningful in the reference either
    if (!(sum == sum) || sum > 1e9f) dyn = K;
    dyn = max(dyn, 1);
    ptr = 0;
    for (int r = 0; r < dyn; ++r) {
        rv[tid] = ptr < OTA_K ? sh_c[tid * OTA_K + ptr] : INFINITY;
        ri[tid] = ptr < OTA_K ? sh_q[tid * OTA_K + ptr] : 0x7fffffff;
        rt[tid] = tid;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if (tid < st) {
                const float ov = rv[tid + st];
                const int oq = ri[tid + st];
                if (ov < rv[tid] || (ov == rv[tid] && oq < ri[tid])) { rv[tid] = ov; ri[tid] = oq; rt[tid] = rt[tid + st]; }
            }
            __syncthreads();
        }
        const int wq = ri[0], wt = rt[0];
        __syncthreads();
        if (wq == 0x7fffffff) break;
        if (tid == wt) {
            ++ptr;
            atomicAdd(O.cnt + wq, 1);
            O.who[wq] = g;
        }
    }
}

__global__ __launch_bounds__(256) void ota_resolve_kernel(OtaArgs O, long long nq) {
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
    if (q >= nq) return;
    const int b = O.cimg[q];
    int res = -1;
    if (b >= 0) {
        const int c = O.cnt[q];
        if (c == 1) res = O.who[q];
        else if (c > 1) {
            const int level = (int)(q / (5 * O.A.na * O.A.NT));
            float best = INFINITY;
            bool have = false;
            for (int r = O.istart[b]; r < O.istart[b + 1]; ++r) {
                const int g = O.order[r];
                float gb[4], iou;
                int gc;
                ota_gt_box(O, g, gb, gc);
                const float cost = ota_pair(O, level, (size_t)q, gb, gc, iou);
                if (!have || cost < best) { best = cost; res = g; have = true; }
            }
        }
    }
    O.match[q] = res;
}

extern "C" int et_ota_workspace_bytes(int B, int na, int nl, int NT, size_t* bytes) {
    if (!bytes) return -1;
    if (B < 0 || na < 1 || nl < 1 || NT < 0) return -2;
    const size_t nq = (size_t)nl * 5 * na * NT;
    *bytes = sizeof(int) * (NT + B + 1 + 3 * nq) + sizeof(float) * 5 * nq + sizeof(long long) * nq + 64;
    return 0;
}

extern "C" int et_ota_assign(const et_loss_desc* d, const float* strides, float img_size, int top_k, void* workspace, int* match,
                             et_stream_t stream) {
    if (!d || !d->targets || !strides || !workspace || !match) return -1;
    if (d->nl < 1 || d->nl > LOSS_MAXL || d->na < 1 || d->na > 3 || d->nc < 1 || d->NT < 0 || top_k < 1 || top_k > OTA_K) return -2;
    if (d->dtype != ET_F32 && d->dtype != ET_BF16 && d->dtype != ET_F16) return -2;
    const long long nq = (long long)d->nl * 5 * d->na * d->NT;
    if (nq >= (1ll << 31)) return -2;
    if (d->NT == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    OtaArgs O;
    LossArgs& A = O.A;
    A.dtype = d->dtype; A.B = d->B; A.na = d->na; A.nc = d->nc; A.no = d->nc + 5; A.NT = d->NT; A.nl = d->nl;
    A.anchor_t = d->anchor_t; A.gr = d->gr; A.cp = d->cp; A.cn = d->cn; A.cls_pw = d->cls_pw; A.obj_pw = d->obj_pw;
    A.box_w = d->box_w; A.obj_w = d->obj_w; A.cls_w = d->cls_w;
    A.pass_mask = 1; A.ignore_obj = 0; A.obj_ch = 4; A.ota_match = nullptr; A.fl_gamma = 0.f; A.balance_dev = nullptr; A.ssi = -1;
    A.tgt = d->targets; A.acc = nullptr;
    for (int l = 0; l < LOSS_MAXL; ++l) {
        LossLevel& L = O.L[l];
        O.stride[l] = 0.f;
        L.p = nullptr; L.dp = nullptr; L.tobj = nullptr; L.sb = L.sa = L.sy = L.sx = 0; L.ny = L.nx = 1; L.balance = 0.f;
        for (int a = 0; a < 3; ++a) L.anchors[a][0] = L.anchors[a][1] = 1.f;
        if (l >= d->nl) continue;
        const et_loss_level* e = &d->level[l];
        if (!e->p) return -1;
        L.p = e->p; L.sb = e->sb; L.sa = e->sa; L.sy = e->sy; L.sx = e->sx; L.ny = e->ny; L.nx = e->nx;
        for (int a = 0; a < d->na; ++a) { L.anchors[a][0] = e->anchors[2 * a]; L.anchors[a][1] = e->anchors[2 * a + 1]; }
        O.stride[l] = strides[l];
    }
    O.img_size = img_size; O.topk = top_k;
    char* w = (char*)workspace;
    O.coff = (long long*)w;  w += sizeof(long long) * nq;
    O.cbox = (float*)w;      w += sizeof(float) * 4 * nq;
    O.cs0 = (float*)w;       w += sizeof(float) * nq;
    O.cimg = (int*)w;        w += sizeof(int) * nq;
    O.cnt = (int*)w;         w += sizeof(int) * nq;
    O.who = (int*)w;         w += sizeof(int) * nq;
    O.order = (int*)w;       w += sizeof(int) * d->NT;
    O.istart = (int*)w;
    O.match = match;
    const int nslot = 5 * d->na * d->NT;
    hipLaunchKernelGGL(ota_order_kernel, dim3((max(d->NT, d->B + 1) + 255) / 256), dim3(256), 0, s, O);
    hipLaunchKernelGGL(ota_cand_kernel, dim3((nslot + 255) / 256, d->nl), dim3(256), 0, s, O);
    hipLaunchKernelGGL(ota_gt_kernel, dim3(d->NT), dim3(256), 0, s, O);
    hipLaunchKernelGGL(ota_resolve_kernel, dim3(et_cdiv(nq, 256)), dim3(256), 0, s, O, nq);
    ET_CHECK_LAUNCH();
    return 0;
}

// ---- pseudo-label -> target table (ComputeStudentMatchLoss.select_targets, ssod_loss.py:130-192) ----------
// t9 (N,9) fp64 [img, cls, x, y, w, h, conf, obj_conf, cls_conf]; valid (N) uint8 or NULL;
// thresholds per class (fp64, compared in fp64 as the reference does); table (N,8) fp32.
__global__ __launch_bounds__(256) void select_targets_kernel(const double* __restrict__ t9, const unsigned char* __restrict__ valid,
                                                             int N, const double* __restrict__ thr_low,
                                                             const double* __restrict__ thr_high, int nc, int with_obj,
                                                             float* __restrict__ table) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= N) return;
    const double* t = t9 + (size_t)k * 9;
    float* o = table + (size_t)k * 8;
    int flags = 0;
    float score = 0.f;
    if (!valid || valid[k]) {
        int c = (int)t[1];
        c = min(max(c, 0), nc - 1);
        if (t[6] >= thr_high[c]) {
            flags = 1; score = (float)t[6];
        } else if (t[6] >= thr_low[c]) {
            flags = 2;
            if (with_obj) {
                score = (float)t[7];
                if (t[7] >= 0.99) flags |= 4;
                if (t[8] >= 0.99) flags |= 8;
            } else {
                score = (float)t[6];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) o[i] = (float)t[i];
    o[6] = score;
    o[7] = (float)flags;
}

// fp32 gradient (loss kernels) -> activation dtype with the upstream scale folded in
template <typename T>
__global__ __launch_bounds__(256) void scale_cast_kernel(const float* __restrict__ s, T* __restrict__ d, long long n, float scale,
                                                         const float* __restrict__ dev_scale) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const float f = dev_scale ? scale * dev_scale[0] : scale;
    if (i < n) d[i] = et_elem<T>::st(s[i] * f);
}

// bf16 destination, eight elements per thread (two 16-byte loads, one 16-byte store; the element-per-thread form moves 128 bytes
// per wave instruction: 2.7 TB/s on the head gradients).  Same arithmetic per element: one multiplication, round to nearest even.
template <typename T = uint16_t>          // T: the 16-bit destination format
__global__ __launch_bounds__(256) void scale_cast_bf16_vec8_kernel(const float* __restrict__ s, T* __restrict__ d, long long n8,
                                                                   float scale, const float* __restrict__ dev_scale) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const float f = dev_scale ? scale * dev_scale[0] : scale;
    const float4 a = *(const float4*)(s + i * 8), b = *(const float4*)(s + i * 8 + 4);
    *(uint4*)(d + i * 8) = make_uint4(et_lp<T>::pack(a.x * f, a.y * f), et_lp<T>::pack(a.z * f, a.w * f), et_lp<T>::pack(b.x * f, b.y * f),
                                      et_lp<T>::pack(b.z * f, b.w * f));
}

// ---- host -------------------------------------------------------------------------------------------------
extern "C" int et_yolo_loss(const et_loss_desc* d, et_stream_t stream) {
    if (!d || !d->targets || !d->acc_ws || !d->out) return -1;
    if (d->nl < 1 || d->nl > LOSS_MAXL || d->na < 1 || d->na > 3 || d->nc < 1 || d->NT < 0) return -2;
    if ((long long)5 * d->na * d->NT >= (1ll << 27)) return -2;
    hipStream_t s = (hipStream_t)stream;
    LossArgs A;
    A.dtype = d->dtype; A.B = d->B; A.na = d->na; A.nc = d->nc; A.no = d->nc + 5; A.NT = d->NT; A.nl = d->nl;
    A.anchor_t = d->anchor_t; A.gr = d->gr; A.cp = d->cp; A.cn = d->cn; A.cls_pw = d->cls_pw; A.obj_pw = d->obj_pw;
    A.box_w = d->box_w; A.obj_w = d->obj_w; A.cls_w = d->cls_w;
    A.pass_mask = d->pass_mask; A.ignore_obj = d->ignore_obj;
    A.ota_match = d->ota_match; A.obj_ch = d->obj_channel ? d->obj_channel : 4;
    A.fl_gamma = d->fl_gamma;
    A.balance_dev = d->balance_dev; A.ssi = d->balance_dev ? d->autobalance_ssi : -1;
    if (A.balance_dev && A.ssi >= A.nl) return -2;
    if (A.obj_ch < 4 || A.obj_ch >= A.no) return -2;
    A.tgt = d->targets; A.acc = d->acc_ws;
    if (A.dtype != ET_F32 && A.dtype != ET_BF16 && A.dtype != ET_F16) return -2;
    (void)hipMemsetAsync(d->acc_ws, 0, sizeof(float) * 16 * LOSS_MAXL, s);
    float bal[4] = {0, 0, 0, 0};
    long long ncell[4] = {0, 0, 0, 0};
    LossLevels Ls;
    long long ob_max = 1;
    for (int l = 0; l < d->nl; ++l) {
        const et_loss_level* e = &d->level[l];
        if (!e->p || !e->dp || !e->tobj_ws) return -1;
        LossLevel& L = Ls.l[l];
        L.p = e->p; L.dp = e->dp; L.tobj = (unsigned long long*)e->tobj_ws;
        L.sb = e->sb; L.sa = e->sa; L.sy = e->sy; L.sx = e->sx; L.ny = e->ny; L.nx = e->nx;
        for (int a = 0; a < d->na; ++a) { L.anchors[a][0] = e->anchors[2 * a]; L.anchors[a][1] = e->anchors[2 * a + 1]; }
        L.balance = e->balance;
        bal[l] = e->balance;
        ncell[l] = (long long)d->B * d->na * e->ny * e->nx;
        if (ncell[l] >= (1ll << 31)) return -2;
        Ls.fixed_n[l] = d->ignore_obj ? 0.f : (float)ncell[l];
        const long long ob = et_cdiv(ncell[l], 256);
        ob_max = ob > ob_max ? ob : ob_max;
        (void)hipMemsetAsync(e->tobj_ws, 0, (size_t)ncell[l] * 8, s);
    }
    for (int l = d->nl; l < LOSS_MAXL; ++l) { Ls.l[l] = Ls.l[0]; Ls.fixed_n[l] = 0.f; }
    const int nslot = 5 * d->na * d->NT;
    if (nslot > 0) {
        const dim3 grid((nslot + 255) / 256, LOSS_NPASS, d->nl);
        hipLaunchKernelGGL(loss_count_kernel, grid, dim3(256), 0, s, A, Ls);
        hipLaunchKernelGGL(loss_pos_kernel, grid, dim3(256), 0, s, A, Ls);
    }
    const dim3 og((unsigned)(ob_max < 512 ? ob_max : 512), 1, d->nl);     // two workgroups per CU and level, grid-stride
    if (d->ignore_obj) hipLaunchKernelGGL(loss_obj_kernel, og, dim3(256), 0, s, A, Ls, 0);
    hipLaunchKernelGGL(loss_obj_kernel, og, dim3(256), 0, s, A, Ls, 1);
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, s, A, bal[0], bal[1], bal[2], bal[3], ncell[0], ncell[1],
                       ncell[2], ncell[3], d->out);
    ET_CHECK_LAUNCH();
    return 0;
}

extern "C" int et_select_targets(const double* targets9, const uint8_t* valid, int N, const double* thr_low,
                                 const double* thr_high, int nc, int with_obj, float* table, et_stream_t stream) {
    if (!targets9 || !thr_low || !thr_high || !table) return -1;
    if (N < 0 || nc < 1) return -2;
    if (N == 0) return 0;
    hipLaunchKernelGGL(select_targets_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, targets9, valid, N,
                       thr_low, thr_high, nc, with_obj, table);
    ET_CHECK_LAUNCH();
    return 0;
}

extern "C" int et_scale_cast(const float* src, void* dst, int dtype, int64_t n, float scale, const float* dev_scale,
                             et_stream_t stream) {
    if (!src || !dst) return -1;
    if (n <= 0) return n == 0 ? 0 : -2;
    const dim3 grid(et_cdiv(n, 256));
    if (dtype == ET_F32) hipLaunchKernelGGL((scale_cast_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, src, (float*)dst, (long long)n, scale, dev_scale);
    else if (dtype == ET_BF16 || dtype == ET_F16) {
        long long done = 0;
        if (((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0 && n >= 8) {
            const long long n8 = n / 8;
            if (dtype == ET_BF16) hipLaunchKernelGGL((scale_cast_bf16_vec8_kernel<uint16_t>), dim3(et_cdiv(n8, 256)), dim3(256), 0, (hipStream_t)stream, src, (uint16_t*)dst, n8, scale, dev_scale);
            else hipLaunchKernelGGL((scale_cast_bf16_vec8_kernel<et_f16>), dim3(et_cdiv(n8, 256)), dim3(256), 0, (hipStream_t)stream, src, (et_f16*)dst, n8, scale, dev_scale);
            done = n8 * 8;
        }
        if (done < n) {
            if (dtype == ET_BF16) hipLaunchKernelGGL((scale_cast_kernel<uint16_t>), dim3(et_cdiv(n - done, 256)), dim3(256), 0, (hipStream_t)stream, src + done,
                                                     (uint16_t*)dst + done, (long long)(n - done), scale, dev_scale);
            else hipLaunchKernelGGL((scale_cast_kernel<et_f16>), dim3(et_cdiv(n - done, 256)), dim3(256), 0, (hipStream_t)stream, src + done,
                                    (et_f16*)dst + done, (long long)(n - done), scale, dev_scale);
        }
    } else return -2;
    ET_CHECK_LAUNCH();
    return 0;
}

// ---- domain-adaptation focal loss (DomainLoss / TargetLoss, models/loss/loss.py:376-421 with
// DomainFocalLoss :312-368, class_num 2, alpha 1, gamma 2, mean): per pixel p = softmax(l0,l1)[label],
// L = -(1-p)^2 log p.  One launch per pyramid level: accumulates sum(L) into out[0] (atomic) and writes
// dL/dlogits * gscale into the (B,H,W,ldg) gradient buffer (channels 0,1), other channels untouched.
template <typename T>
__global__ __launch_bounds__(256) void domain_focal_kernel(const T* __restrict__ f, int ldf, long long P, int label,
                                                           float gscale, T* __restrict__ g, int ldg, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    float l = 0.f;
    if (i < P) {
        const float a = et_elem<T>::ld(f[i * ldf + 0]), b = et_elem<T>::ld(f[i * ldf + 1]);
        const float m = fmaxf(a, b);
        const float ea = expf(a - m), eb = expf(b - m);
        const float inv = 1.0f / (ea + eb);
        const float p = (label == 0 ? ea : eb) * inv;
        const float q = 1.0f - p;
        const float lp = logf(p);
        l = -(q * q) * lp;
        // dL/dp = 2 q log p - q^2 / p ; dp/dx_label = p q ; dp/dx_other = -p q
        const float dldp = 2.0f * q * lp - (q * q) / p;
        const float gt = dldp * p * q * gscale;
        if (g) {
            g[i * ldg + label] = et_elem<T>::st(gt);
            g[i * ldg + (1 - label)] = et_elem<T>::st(-gt);
        }
    }
    l = et_wave_sum(l);
    if ((threadIdx.x & 63) == 0 && l != 0.f) atomicAdd(out, l);
}

// x *= alpha in place (GradReverse backward, models/detector/yolo_ssod.py:158-171: alpha = -1)
template <typename T>
__global__ __launch_bounds__(256) void scale_inplace_kernel(T* __restrict__ x, long long n, float alpha,
                                                            const float* __restrict__ dev_scale) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const float f = dev_scale ? alpha * dev_scale[0] : alpha;
    if (i < n) x[i] = et_elem<T>::st(et_elem<T>::ld(x[i]) * f);
}

extern "C" int et_domain_focal(const void* feat, int ldf, int dtype, int64_t P, int label, float gscale, void* grad,
                               int ldg, float* loss_sum, et_stream_t stream) {
    if (!feat || !loss_sum) return -1;
    if (P <= 0 || (label != 0 && label != 1)) return -2;
    const dim3 grid(et_cdiv(P, 256));
    if (dtype == ET_F32) hipLaunchKernelGGL((domain_focal_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)feat, ldf, (long long)P, label, gscale, (float*)grad, ldg, loss_sum);
    else if (dtype == ET_BF16) hipLaunchKernelGGL((domain_focal_kernel<uint16_t>), grid, dim3(256), 0, (hipStream_t)stream, (const uint16_t*)feat, ldf, (long long)P, label, gscale, (uint16_t*)grad, ldg, loss_sum);
    else if (dtype == ET_F16) hipLaunchKernelGGL((domain_focal_kernel<et_f16>), grid, dim3(256), 0, (hipStream_t)stream, (const et_f16*)feat, ldf, (long long)P, label, gscale, (et_f16*)grad, ldg, loss_sum);
    else return -2;
    ET_CHECK_LAUNCH();
    return 0;
}

extern "C" int et_scale_inplace(void* x, int dtype, int64_t n, float alpha, const float* dev_scale, et_stream_t stream) {
    if (!x) return -1;
    if (n <= 0) return n == 0 ? 0 : -2;
    const dim3 grid(et_cdiv(n, 256));
    if (dtype == ET_F32) hipLaunchKernelGGL((scale_inplace_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, (float*)x, (long long)n, alpha, dev_scale);
    else if (dtype == ET_BF16) hipLaunchKernelGGL((scale_inplace_kernel<uint16_t>), grid, dim3(256), 0, (hipStream_t)stream, (uint16_t*)x, (long long)n, alpha, dev_scale);
    else if (dtype == ET_F16) hipLaunchKernelGGL((scale_inplace_kernel<et_f16>), grid, dim3(256), 0, (hipStream_t)stream, (et_f16*)x, (long long)n, alpha, dev_scale);
    else return -2;
    ET_CHECK_LAUNCH();
    return 0;
}
