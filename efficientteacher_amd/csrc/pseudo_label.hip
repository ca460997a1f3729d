// Online pseudo-label creation on the device (reference
// FairPseudoLabel.create_pseudo_label_online_with_gt, utils/self_supervised_utils.py:194-245, with
// output_to_target_ssod utils/plots.py:485-491, online_label_transform :414-454, box_candidates
// :316-321, xywh2xyxy / xyxy2xywh utils/general.py:630 / 549).  The reference moves every detection
// to the host and loops in numpy; here one thread per (image, detection slot) does the same arithmetic
// with the same precisions: xyxy->xywh in fp32 (plots.py:490 runs on the fp32 NMS row), then fp64 for
// xywh->xyxy, the affine warp of the 4 corners by M, the clip, the candidate filter, xyxy->xywh,
// normalisation and the ud/lr flips.  Output stays padded (B*max_det rows + valid mask) so that the
// loss kernels can consume it without a host synchronisation.  Compiled with -ffp-contract=off.
#include "et_device.h"
#include "../../include/et_hip.h"

__global__ __launch_bounds__(256) void pseudo_label_kernel(const float* __restrict__ dets, const int* __restrict__ counts,
                                                           const double* __restrict__ M_s, int B, int max_det,
                                                           double width, double height, int clip01,
                                                           double* __restrict__ out, unsigned char* __restrict__ valid) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * max_det) return;
    const int img = i / max_det, k = i - img * max_det;
    double* o = out + (size_t)i * 9;
    bool ok = k < counts[img];
    // M_s[M_s[:, 0] == img][0]
    const double* row = nullptr;
    for (int r = 0; r < B; ++r)
        if (M_s[(size_t)r * 13] == (double)img) { row = M_s + (size_t)r * 13; break; }
    if (!row) ok = false;
    double res[9] = {(double)img, 0, 0, 0, 0, 0, 0, 0, 0};
    if (ok) {
        const float* d = dets + (size_t)i * 8;
        // plots.py:490  xyxy2xywh on the fp32 row
        const float cx = (d[0] + d[2]) / 2, cy = (d[1] + d[3]) / 2, w = d[2] - d[0], h = d[3] - d[1];
        // self_supervised_utils.py:209  xywh2xyxy in fp64
        const double X = cx, Y = cy, Wd = w, Hd = h;
        const double x1 = X - Wd / 2, y1 = Y - Hd / 2, x2 = X + Wd / 2, y2 = Y + Hd / 2;
        const double* M = row + 1;
        const double s = row[10];
        const int ud = (int)row[11], lr = (int)row[12];
        // corners x1y1, x2y2, x1y2, x2y1 warped by M (affine: third row ignored)   :433-436
        const double px[4] = {x1, x2, x1, x2}, py[4] = {y1, y2, y2, y1};
        double nx1 = 0, ny1 = 0, nx2 = 0, ny2 = 0;
        for (int c = 0; c < 4; ++c) {
            const double tx = px[c] * M[0] + py[c] * M[1] + M[2];
            const double ty = px[c] * M[3] + py[c] * M[4] + M[5];
            if (c == 0) { nx1 = nx2 = tx; ny1 = ny2 = ty; }
            else { nx1 = fmin(nx1, tx); nx2 = fmax(nx2, tx); ny1 = fmin(ny1, ty); ny2 = fmax(ny2, ty); }
        }
        nx1 = fmin(fmax(nx1, 0.0), width); nx2 = fmin(fmax(nx2, 0.0), width);     // :445-446
        ny1 = fmin(fmax(ny1, 0.0), height); ny2 = fmin(fmax(ny2, 0.0), height);
        // box_candidates(box1 = xyxy * s, box2 = new)   :316-321, :449
        const double w1 = x2 * s - x1 * s, h1 = y2 * s - y1 * s;
        const double w2 = nx2 - nx1, h2 = ny2 - ny1;
        const double eps = 1e-16;
        const double ar = fmax(w2 / (h2 + eps), h2 / (w2 + eps));
        ok = (w2 > 2) && (h2 > 2) && (w2 * h2 / (w1 * h1 + eps) > 0.1) && (ar < 20);
        if (ok) {
            double ox = (nx1 + nx2) / 2, oy = (ny1 + ny2) / 2;                       // xyxy2xywh :223
            const double ow = nx2 - nx1, oh = ny2 - ny1;
            ox /= width; oy /= height;                                             // :224-225
            double nw = ow / width, nh = oh / height;
            if (clip01) {       // LabelMatch only (utils/labelmatch.py:333): xywh clipped to [0, 1] before the flips
                ox = fmin(fmax(ox, 0.0), 1.0); oy = fmin(fmax(oy, 0.0), 1.0);
                nw = fmin(fmax(nw, 0.0), 1.0); nh = fmin(fmax(nh, 0.0), 1.0);
            }
            if (ud == 1) oy = 1 - oy;
            if (lr == 1) ox = 1 - ox;
            res[1] = d[5]; res[2] = ox; res[3] = oy; res[4] = nw; res[5] = nh;
            res[6] = d[4]; res[7] = d[6]; res[8] = d[7];
        }
    }
#pragma unroll
    for (int c = 0; c < 9; ++c) o[c] = res[c];
    valid[i] = ok ? 1 : 0;
}

extern "C" int et_pseudo_label_transform(const float* dets, const int* counts, const double* M_s, int B, int max_det,
                                         int width, int height, int clip01, double* targets9, uint8_t* valid, et_stream_t stream) {
    if (!dets || !counts || !M_s || !targets9 || !valid) return -1;
    if (B <= 0 || max_det <= 0) return -2;
    hipLaunchKernelGGL(pseudo_label_kernel, dim3((B * max_det + 255) / 256), dim3(256), 0, (hipStream_t)stream, dets, counts,
                       M_s, B, max_det, (double)width, (double)height, clip01, targets9, valid);
    ET_CHECK_LAUNCH();
    return 0;
}

// LabelMatch score bookkeeping (utils/labelmatch.py:279-287): the reference appends the confidence of EVERY detection the
// NMS returned to a per-class python list on the host (score_list_epoch), read once per epoch by update_epoch_cls_thr.
// Here the (class, confidence) pairs are appended to a device log through one atomic counter; the per-class lists are
// formed at the end of the epoch, where only the sorted values matter, so the append order is free.  log_count keeps
// counting past `cap` so that an overflow is visible to the host.
__global__ __launch_bounds__(256) void score_log_kernel(const float* __restrict__ dets, const int* __restrict__ counts, int B,
                                                        int max_det, float* __restrict__ conf_log, int* __restrict__ cls_log,
                                                        unsigned long long* __restrict__ log_count, long long cap) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool live = i < B * max_det && (i % max_det) < counts[i / max_det];
    const unsigned long long m = __ballot(live);
    if (!m) return;
    const int lane = threadIdx.x & 63;
    unsigned long long base = 0;
    const int leader = __ffsll((long long)m) - 1;
    if (lane == leader) base = atomicAdd(log_count, (unsigned long long)__popcll(m));
    const unsigned lo = __shfl((unsigned)(base & 0xffffffffull), leader), hi = __shfl((unsigned)(base >> 32), leader);
    base = ((unsigned long long)hi << 32) | lo;
    if (live) {
        const long long pos = (long long)base + __popcll(m & ((1ull << lane) - 1ull));
        if (pos < cap) {
            const float* d = dets + (size_t)i * 8;
            conf_log[pos] = d[4];
            cls_log[pos] = (int)d[5];
        }
    }
}

extern "C" int et_score_log_append(const float* dets, const int* counts, int B, int max_det, float* conf_log, int* cls_log,
                                   uint64_t* log_count, int64_t cap, et_stream_t stream) {
    if (!dets || !counts || !conf_log || !cls_log || !log_count) return -1;
    if (B <= 0 || max_det <= 0 || cap <= 0) return -2;
    hipLaunchKernelGGL(score_log_kernel, dim3((B * max_det + 255) / 256), dim3(256), 0, (hipStream_t)stream, dets, counts, B,
                       max_det, conf_log, cls_log, (unsigned long long*)log_count, (long long)cap);
    ET_CHECK_LAUNCH();
    return 0;
}
