// Pseudo-label filter: non_max_suppression_ssod (reference utils/general.py:887-992) and the
// torchvision.ops.nms it calls (utils/general.py:976), as four stream-ordered gfx950 kernels with no
// host round trip.  Compiled with -ffp-contract=off: every fp32 op is rounded exactly like the
// reference's separate torch ops so that the kept INDICES are bit-exact.
//
//   K1 nms_filter  : HBM-bound scan of pred (B, A, no).  One wave owns a tile of 64 anchors: the
//                    64*no floats are read fully coalesced into LDS (row stride `no` is odd for
//                    COCO => conflict-free), then lane r reduces row r: obj > thr, cls max,
//                    conf = max(cls*obj) with first-index argmax, xywh->xyxy.  Survivors are
//                    compacted inside the tile with a ballot (order preserving).
//   K2 nms_compact : per image, scan the tile counts and copy tile records into the order-preserving
//                    candidate list (== the reference's boolean-mask row order), emit 64-bit sort
//                    keys (~score_bits << 32 | candidate_index)  => ascending key order is
//                    "score descending, stable".
//   K3 nms_rank    : rank = #keys smaller than mine (keys are unique) -> order[rank] = candidate.
//   K4 nms_greedy  : one workgroup per image walks the sorted candidates in chunks of 256: test
//                    against the kept list, pairwise bitmask inside the chunk, serial resolution
//                    over set bits only; stops at max_det kept (the reference truncates after NMS).
#include "et_device.h"
#include "../../include/et_hip.h"

#define NMS_MAX_NO 96
#define NMS_TILE 64
#define NMS_REC 8      // x1 y1 x2 y2 conf cls obj cls_score
#define NMS_CHUNK 256
#define NMS_MAX_DET 1024

struct NmsWs {
    float* tile_rec;               // [B][T][64][8]
    int* tile_cnt;                 // [B][T]
    float* cand;                   // [B][A][8]
    unsigned long long* keys;      // [B][A]
    int* order;                    // [B][A]
    int* ncand;                    // [B]
};

static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

static size_t nms_carve(void* base, int B, int A, NmsWs* ws) {
    const size_t T = (A + NMS_TILE - 1) / NMS_TILE;
    size_t off = 0;
    char* p = (char*)base;
    auto take = [&](size_t bytes) { char* r = p ? p + off : nullptr; off += al256(bytes); return r; };
    char* a = take((size_t)B * T * NMS_TILE * NMS_REC * 4);
    char* b = take((size_t)B * T * 4);
    char* c = take((size_t)B * A * NMS_REC * 4);
    char* d = take((size_t)B * A * 8);
    char* e = take((size_t)B * A * 4);
    char* f = take((size_t)B * 4);
    if (ws) {
        ws->tile_rec = (float*)a; ws->tile_cnt = (int*)b; ws->cand = (float*)c;
        ws->keys = (unsigned long long*)d; ws->order = (int*)e; ws->ncand = (int*)f;
    }
    return off;
}

__global__ __launch_bounds__(256) void nms_filter_kernel(const float* __restrict__ pred, int A, int no,
                                                         float conf_thres, int T, NmsWs ws) {
    __shared__ __attribute__((aligned(16))) float lds[4][NMS_TILE * NMS_MAX_NO];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int img = blockIdx.y;
    const int tile = blockIdx.x * 4 + wave;
    if (tile >= T) return;                      // wave-uniform
    const int row0 = tile * NMS_TILE;
    const int nrows = min(NMS_TILE, A - row0);
    const size_t gbase = ((size_t)img * A + row0) * (size_t)no;
    const float* src = pred + gbase;
    const int nfl = nrows * no;
    float* L = lds[wave];
    if ((((uintptr_t)src) & 15) == 0) {
        const int nv = nfl >> 2;
        const float4* s4 = (const float4*)src;
        float4* l4 = (float4*)L;
        for (int i = lane; i < nv; i += 64) l4[i] = s4[i];
        for (int i = (nv << 2) + lane; i < nfl; i += 64) L[i] = src[i];
    } else {
        for (int i = lane; i < nfl; i += 64) L[i] = src[i];
    }
    // the tile is private to this wave: LDS ops of one wave complete in issue order, so a wave-level
    // scheduling barrier is all that is needed between the ds_writes above and the row reads below
    // (no block barrier: sibling waves may already have exited).
    __builtin_amdgcn_wave_barrier();
    bool pass = false;
    float rec[NMS_REC];
    if (lane < nrows) {
        const float* r = L + lane * no;
        const float obj = r[4];
        if (obj > conf_thres) {
            const int nc = no - 5;
            float cls_score = r[5];
            float conf = r[5] * obj;
            int j = 0;
            for (int c = 1; c < nc; ++c) {
                const float v = r[5 + c];
                cls_score = fmaxf(cls_score, v);
                const float pv = v * obj;
                if (pv > conf) { conf = pv; j = c; }
            }
            if (conf > conf_thres) {
                pass = true;
                const float hw = r[2] / 2, hh = r[3] / 2;
                rec[0] = r[0] - hw; rec[1] = r[1] - hh; rec[2] = r[0] + hw; rec[3] = r[1] + hh;
                rec[4] = conf; rec[5] = (float)j; rec[6] = obj; rec[7] = cls_score;
            }
        }
    }
    const unsigned long long m = __ballot(pass);
    const int pos = __popcll(m & ((1ull << lane) - 1ull));
    if (pass) {
        float4* dst = (float4*)(ws.tile_rec + (((size_t)img * T + tile) * NMS_TILE + pos) * NMS_REC);
        dst[0] = make_float4(rec[0], rec[1], rec[2], rec[3]);
        dst[1] = make_float4(rec[4], rec[5], rec[6], rec[7]);
    }
    if (lane == 0) ws.tile_cnt[(size_t)img * T + tile] = __popcll(m);
}

__global__ __launch_bounds__(256) void nms_compact_kernel(int A, int T, NmsWs ws) {
    __shared__ int offs[4096 + 1];
    __shared__ int wsum[4];
    const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int* cnt = ws.tile_cnt + (size_t)img * T;
    // exclusive scan of T (<= 4096) tile counts, 256 at a time
    int run = 0;
    for (int base = 0; base < T; base += 256) {
        const int t = base + tid;
        const int v = t < T ? cnt[t] : 0;
        int inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(inc, d);
            if (lane >= d) inc += o;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wave; ++w) woff += wsum[w];
        const int tot = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        if (t < T) offs[t] = run + woff + inc - v;
        run += tot;
        __syncthreads();
    }
    if (tid == 0) { offs[T] = run; ws.ncand[img] = run; }
    __syncthreads();
    for (int t = wave; t < T; t += 4) {
        const int c = cnt[t];
        if (lane < c) {
            const int ci = offs[t] + lane;
            const float4* s = (const float4*)(ws.tile_rec + (((size_t)img * T + t) * NMS_TILE + lane) * NMS_REC);
            float4* d = (float4*)(ws.cand + ((size_t)img * A + ci) * NMS_REC);
            const float4 a = s[0], b = s[1];
            d[0] = a; d[1] = b;
            // conf > 0 => float bits are monotone; invert for descending
            const unsigned long long k = ((unsigned long long)(~__float_as_uint(b.x)) << 32) | (unsigned)ci;
            ws.keys[(size_t)img * A + ci] = k;
        }
    }
}

__global__ __launch_bounds__(256) void nms_rank_kernel(int A, NmsWs ws) {
    __shared__ unsigned long long tile[256];
    const int img = blockIdx.y, tid = threadIdx.x;
    const int n = ws.ncand[img];
    if ((int)(blockIdx.x * 256) >= n) return;   // block-uniform
    const int i = blockIdx.x * 256 + tid;
    const unsigned long long* keys = ws.keys + (size_t)img * A;
    const unsigned long long mine = i < n ? keys[i] : ~0ull;
    int rank = 0;
    for (int base = 0; base < n; base += 256) {
        __syncthreads();
        tile[tid] = (base + tid) < n ? keys[base + tid] : ~0ull;
        __syncthreads();
        const int lim = min(256, n - base);
        for (int k = 0; k < lim; ++k) rank += tile[k] < mine ? 1 : 0;
    }
    if (i < n) ws.order[(size_t)img * A + rank] = i;
}

__device__ __forceinline__ bool nms_over(const float4 a, float aa, const float4 b, float ab, float thr) {
    const float xx1 = fmaxf(a.x, b.x), yy1 = fmaxf(a.y, b.y);
    const float xx2 = fminf(a.z, b.z), yy2 = fminf(a.w, b.w);
    const float w = fmaxf(0.0f, xx2 - xx1), h = fmaxf(0.0f, yy2 - yy1);
    const float inter = w * h;
    const float ovr = inter / (aa + ab - inter);
    return ovr > thr;
}

__global__ __launch_bounds__(256) void nms_greedy_kernel(int A, float iou_thres, float class_off, int max_det,
                                                         NmsWs ws, float* __restrict__ dets,
                                                         int* __restrict__ counts, long long* __restrict__ keep) {
    __shared__ float4 kbox[NMS_MAX_DET];
    __shared__ float karea[NMS_MAX_DET];
    __shared__ float4 cbox[NMS_CHUNK];
    __shared__ float carea[NMS_CHUNK];
    __shared__ int cidx[NMS_CHUNK];
    __shared__ unsigned long long rowmask[NMS_CHUNK][4];
    __shared__ unsigned long long alive[4];
    __shared__ int s_kept;
    const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = ws.ncand[img];
    const float* cand = ws.cand + (size_t)img * A * NMS_REC;
    const int* order = ws.order + (size_t)img * A;
    if (tid == 0) s_kept = 0;
    __syncthreads();
    for (int base = 0; base < n; base += NMS_CHUNK) {
        const int kept = s_kept;
        if (kept >= max_det) break;              // block-uniform
        const int c = base + tid;
        const bool valid = c < n;
        float4 bx = make_float4(0, 0, 0, 0);
        float area = 0.f;
        int ci = 0;
        if (valid) {
            ci = order[c];
            const float4 r0 = ((const float4*)(cand + (size_t)ci * NMS_REC))[0];
            const float4 r1 = ((const float4*)(cand + (size_t)ci * NMS_REC))[1];
            const float off = r1.y * class_off;   // utils/general.py:972  c = cls * max_wh
            bx = make_float4(r0.x + off, r0.y + off, r0.z + off, r0.w + off);
            area = (bx.z - bx.x) * (bx.w - bx.y);
        }
        bool ok = valid;
        for (int k = 0; k < kept && ok; ++k)
            if (nms_over(kbox[k], karea[k], bx, area, iou_thres)) ok = false;
        cbox[tid] = bx; carea[tid] = area; cidx[tid] = ci;
        const unsigned long long am = __ballot(ok);
        if (lane == 0) alive[wave] = am;
        __syncthreads();
        const int cnt = min(NMS_CHUNK, n - base);
        if (ok) {
            for (int w = 0; w < 4; ++w) {
                unsigned long long m = 0;
                if (w * 64 + 63 > tid) {
                    for (int jj = 0; jj < 64; ++jj) {
                        const int j = w * 64 + jj;
                        if (j > tid && j < cnt && nms_over(bx, area, cbox[j], carea[j], iou_thres)) m |= 1ull << jj;
                    }
                }
                rowmask[tid][w] = m;
            }
        }
        __syncthreads();
        if (tid == 0) {
            unsigned long long removed[4] = {0, 0, 0, 0};
            int k = kept;
            for (int w = 0; w < 4 && k < max_det; ++w) {
                unsigned long long a = alive[w];
                while (k < max_det) {
                    const unsigned long long m = a & ~removed[w];
                    if (!m) break;
                    const int b = __ffsll(m) - 1;
                    const int i = w * 64 + b;
                    a &= ~(1ull << b);
                    kbox[k] = cbox[i]; karea[k] = carea[i];
                    keep[(size_t)img * max_det + k] = cidx[i];
                    ++k;
                    for (int x = w; x < 4; ++x) removed[x] |= rowmask[i][x];
                }
            }
            s_kept = k;
        }
        __syncthreads();
    }
    const int kept = s_kept;
    if (tid == 0) counts[img] = kept;
    // gather the (un-offset) rows of the kept candidates
    for (int k = tid; k < max_det; k += 256) {
        float4 a = make_float4(0, 0, 0, 0), b = a;
        if (k < kept) {
            const long long ci = keep[(size_t)img * max_det + k];
            a = ((const float4*)(cand + (size_t)ci * NMS_REC))[0];
            b = ((const float4*)(cand + (size_t)ci * NMS_REC))[1];
        } else {
            keep[(size_t)img * max_det + k] = -1;
        }
        float4* d = (float4*)(dets + ((size_t)img * max_det + k) * NMS_REC);
        d[0] = a; d[1] = b;
    }
}

extern "C" int et_nms_ssod_workspace_bytes(int B, int A, size_t* bytes) {
    if (B <= 0 || A <= 0 || !bytes) return -1;
    *bytes = nms_carve(nullptr, B, A, nullptr);
    return 0;
}

extern "C" int et_nms_ssod(const float* pred, int B, int A, int no, float conf_thres, float iou_thres,
                           int agnostic, int max_det, float* dets, int* counts, int64_t* keep,
                           int* n_candidates, void* workspace, size_t ws_bytes, et_stream_t stream) {
    if (!pred || !dets || !counts || !keep || !workspace) return -1;
    if (B <= 0 || A <= 0 || no < 6 || no > NMS_MAX_NO) return -2;
    if (max_det <= 0 || max_det > NMS_MAX_DET) return -2;
    const int T = (A + NMS_TILE - 1) / NMS_TILE;
    if (T > 4096) return -2;
    NmsWs ws;
    if (nms_carve(workspace, B, A, &ws) > ws_bytes) return -3;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(nms_filter_kernel, dim3((T + 3) / 4, B), dim3(256), 0, s, pred, A, no, conf_thres, T, ws);
    hipLaunchKernelGGL(nms_compact_kernel, dim3(B), dim3(256), 0, s, A, T, ws);
    hipLaunchKernelGGL(nms_rank_kernel, dim3((A + 255) / 256, B), dim3(256), 0, s, A, ws);
    hipLaunchKernelGGL(nms_greedy_kernel, dim3(B), dim3(256), 0, s, A, iou_thres,
                       agnostic ? 0.0f : 7680.0f, max_det, ws, dets, counts, (long long*)keep);
    if (n_candidates)
        (void)hipMemcpyAsync(n_candidates, ws.ncand, (size_t)B * sizeof(int), hipMemcpyDeviceToDevice, s);
    ET_CHECK_LAUNCH();
    return 0;
}
