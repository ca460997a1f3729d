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
    return dtype == ET_F32 ? ((const float*)p)[off] : et_bf2f(((const uint16_t*)p)[off]);
}

// candidate slot s of pass `pass` on this level -> assignment result
__device__ __forceinline__ Slot eval_slot(const LossArgs& A, const LossLevel& L, int pass, int s) {
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
    return r;
}

__global__ __launch_bounds__(256) void loss_count_kernel(LossArgs A, LossLevel L, int level) {
    const int nslot = 5 * A.na * A.NT;
    const int s = blockIdx.x * 256 + threadIdx.x;
    const int pass = blockIdx.y;
    int v = 0;
    if (((A.pass_mask >> pass) & 1) && s < nslot) v = eval_slot(A, L, pass, s).valid ? 1 : 0;
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

__global__ __launch_bounds__(256) void loss_pos_kernel(LossArgs A, LossLevel L, int level) {
    const int nslot = 5 * A.na * A.NT;
    const int s = blockIdx.x * 256 + threadIdx.x;
    const int pass = blockIdx.y;
    float box_sum = 0.f, cls_sum = 0.f;
    bool do_cls = false;
    long long cls_off = 0;
    int cls_c = 0;
    float cls_wgt = 0.f;
    if (((A.pass_mask >> pass) & 1) && s < nslot) {
        const Slot r = eval_slot(A, L, pass, s);
        if (r.valid && r.b >= 0 && r.b < A.B) {
            const float npos = A.acc[level * 16 + ACC_CNT0 + pass];
            const long long off = r.b * L.sb + r.a * L.sa + r.gj * L.sy + r.gi * L.sx;
            const long long cell = (((long long)r.b * A.na + r.a) * L.ny + r.gj) * L.nx + r.gi;
            const bool want_box = pass == 0 || pass == 2;
            const bool want_cls = (pass == 0 || pass == 3) && A.nc > 1;
            if (want_box) {
                float lg[4], sg[4], pb[4], g[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) { lg[i] = ld_logit(L.p, A.dtype, off + i); sg[i] = 1.0f / (1.0f + expf(-lg[i])); }
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
    // (coalesced loads and atomics; one lane per slot looping over 80 classes touched 64 different cache lines
    // per iteration)
    {
        const int lane = threadIdx.x & 63;
        unsigned long long m = __ballot(do_cls);
        while (m) {
            const int src = __ffsll((long long)m) - 1;
            m &= m - 1;
            const unsigned lo = __shfl((unsigned)(cls_off & 0xffffffffll), src);
            const unsigned hi = __shfl((unsigned)((unsigned long long)cls_off >> 32), src);
            const long long off_s = (long long)(((unsigned long long)hi << 32) | lo);
            const int c_s = __shfl(cls_c, src);
            const float wgt_s = __shfl(cls_wgt, src);
            for (int c = lane; c < A.nc; c += 64) {
                const float x = ld_logit(L.p, A.dtype, off_s + 5 + c);
                float gr_;
                cls_sum += bce_logits(x, c == c_s ? A.cp : A.cn, A.cls_pw, gr_);
                atomicAdd(L.dp + off_s + 5 + c, wgt_s * gr_);
            }
        }
    }
    box_sum = et_wave_sum(box_sum);
    cls_sum = et_wave_sum(cls_sum);
    if ((threadIdx.x & 63) == 0) {
        if (box_sum != 0.f) atomicAdd(A.acc + level * 16 + ACC_BOX0 + pass, box_sum);
        if (cls_sum != 0.f) atomicAdd(A.acc + level * 16 + ACC_CLS0 + pass, cls_sum);
    }
}

// mode 0: count non-ignored cells only; mode 1: loss sum + gradient (denominator read from acc)
__global__ __launch_bounds__(256) void loss_obj_kernel(LossArgs A, LossLevel L, int level, int mode, float fixed_n) {
    const long long ncell = (long long)A.B * A.na * L.ny * L.nx;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    float lsum = 0.f, cnt = 0.f;
    if (i < ncell) {
        const unsigned long long w = L.tobj[i];
        const float t = w ? __uint_as_float((unsigned)(w & 0xffffffffull)) : 0.0f;
        if (t >= 0.0f) {
            cnt = 1.f;
            if (mode == 1) {
                const int gi = (int)(i % L.nx);
                const int gj = (int)((i / L.nx) % L.ny);
                const int a = (int)((i / ((long long)L.nx * L.ny)) % A.na);
                const long long b = i / ((long long)L.nx * L.ny * A.na);
                const long long off = b * L.sb + a * L.sa + gj * L.sy + gi * L.sx + 4;
                const float x = ld_logit(L.p, A.dtype, off);
                float g;
                lsum = bce_logits(x, t, A.obj_pw, g);
                const float n = fixed_n > 0.f ? fixed_n : A.acc[level * 16 + ACC_OBJN];
                L.dp[off] = g * (A.obj_w * L.balance / n);
            }
        }
    }
    lsum = et_wave_sum(lsum);
    cnt = et_wave_sum(cnt);
    if ((threadIdx.x & 63) == 0) {
        if (mode == 1) atomicAdd(A.acc + level * 16 + ACC_OBJ, lsum);
        else atomicAdd(A.acc + level * 16 + ACC_OBJN, cnt);
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
        lobj += (a[ACC_OBJ] / nobj) * bal[l];
    }
    lbox *= A.box_w; lobj *= A.obj_w; lcls *= A.cls_w;
    out[0] = lbox; out[1] = lobj; out[2] = lcls;
    out[3] = (lbox + lobj + lcls) * (float)A.B;
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
    A.tgt = d->targets; A.acc = d->acc_ws;
    if (A.dtype != ET_F32 && A.dtype != ET_BF16) return -2;
    (void)hipMemsetAsync(d->acc_ws, 0, sizeof(float) * 16 * LOSS_MAXL, s);
    float bal[4] = {0, 0, 0, 0};
    long long ncell[4] = {0, 0, 0, 0};
    for (int l = 0; l < d->nl; ++l) {
        const et_loss_level* e = &d->level[l];
        if (!e->p || !e->dp || !e->tobj_ws) return -1;
        LossLevel L;
        L.p = e->p; L.dp = e->dp; L.tobj = (unsigned long long*)e->tobj_ws;
        L.sb = e->sb; L.sa = e->sa; L.sy = e->sy; L.sx = e->sx; L.ny = e->ny; L.nx = e->nx;
        for (int a = 0; a < d->na; ++a) { L.anchors[a][0] = e->anchors[2 * a]; L.anchors[a][1] = e->anchors[2 * a + 1]; }
        L.balance = e->balance;
        bal[l] = e->balance;
        ncell[l] = (long long)d->B * d->na * e->ny * e->nx;
        (void)hipMemsetAsync(e->tobj_ws, 0, (size_t)ncell[l] * 8, s);
        const int nslot = 5 * d->na * d->NT;
        if (nslot > 0) {
            const dim3 grid((nslot + 255) / 256, LOSS_NPASS);
            hipLaunchKernelGGL(loss_count_kernel, grid, dim3(256), 0, s, A, L, l);
            hipLaunchKernelGGL(loss_pos_kernel, grid, dim3(256), 0, s, A, L, l);
        }
        const dim3 og(et_cdiv(ncell[l], 256));
        if (d->ignore_obj) hipLaunchKernelGGL(loss_obj_kernel, og, dim3(256), 0, s, A, L, l, 0, 0.f);
        hipLaunchKernelGGL(loss_obj_kernel, og, dim3(256), 0, s, A, L, l, 1, d->ignore_obj ? 0.f : (float)ncell[l]);
    }
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
    else if (dtype == ET_BF16) hipLaunchKernelGGL((scale_cast_kernel<uint16_t>), grid, dim3(256), 0, (hipStream_t)stream, src, (uint16_t*)dst, (long long)n, scale, dev_scale);
    else return -2;
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
    else return -2;
    ET_CHECK_LAUNCH();
    return 0;
}
