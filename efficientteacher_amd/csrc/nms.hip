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

// =====================================================================================================
// General non_max_suppression (reference utils/general.py:994-1100): the val.py path and
// the second NMS entry of the drop-in boundary (8b).  Differences from the SSOD filter: candidates need
// obj > thr AND max(cls) > thr (:1005); multi_label emits one candidate per (anchor, class) with
// cls*obj > thr in (anchor, class) order (:1052-1054, torch.nonzero order); an optional class filter
// (:1061); more than max_nms = 30000 candidates are cut to the 30000 best scores (:1071) -- here with the
// stable rule "score descending, then candidate order", which the unstable torch argsort leaves open.
// The cut is exact and needs no sort of the (up to A*nc ~ 2 M) candidates: a two-level histogram of the
// score bits (high 16, then low 16 inside the boundary bucket) finds the 32-bit threshold key and how many
// of the candidates that TIE with it may pass, in candidate order.  Then count -> scan -> emit writes the
// survivors order-preserving, and the SSOD path's rank + greedy kernels finish the job.
// =====================================================================================================
#define NMSG_BUCKETS 65536

struct NmsgSel {          // per image
    unsigned thr_key;     // candidates with score bits  > thr_key pass, == thr_key pass while quota lasts
    int quota;
    int need;             // level 0 -> 1: how many the boundary bucket must supply
    int done;             // n <= max_nms: everything passes
};

struct NmsgWs {
    unsigned* hist;       // [B][65536]
    NmsgSel* sel;         // [B]
    int* tile_in;         // [B][T] candidates above the threshold
    int* tile_tie;        // [B][T] candidates equal to the threshold
    int* off_in;          // [B][T] exclusive scans of the two
    int* off_tie;
    NmsWs w;              // cand / keys / order / ncand with row capacity `cap`
};

static size_t nmsg_carve(void* base, int B, int A, int cap, NmsgWs* ws) {
    const size_t T = (A + NMS_TILE - 1) / NMS_TILE;
    size_t off = 0;
    char* p = (char*)base;
    auto take = [&](size_t bytes) { char* r = p ? p + off : nullptr; off += al256(bytes); return r; };
    char* h = take((size_t)B * NMSG_BUCKETS * 4);
    char* se = take((size_t)B * sizeof(NmsgSel));
    char* t0 = take((size_t)B * T * 4);
    char* t1 = take((size_t)B * T * 4);
    char* o0 = take((size_t)B * T * 4);
    char* o1 = take((size_t)B * T * 4);
    char* c = take((size_t)B * cap * NMS_REC * 4);
    char* d = take((size_t)B * cap * 8);
    char* e = take((size_t)B * cap * 4);
    char* f = take((size_t)B * 4);
    if (ws) {
        ws->hist = (unsigned*)h; ws->sel = (NmsgSel*)se; ws->tile_in = (int*)t0; ws->tile_tie = (int*)t1;
        ws->off_in = (int*)o0; ws->off_tie = (int*)o1;
        ws->w.tile_rec = nullptr; ws->w.tile_cnt = nullptr; ws->w.cand = (float*)c;
        ws->w.keys = (unsigned long long*)d; ws->w.order = (int*)e; ws->w.ncand = (int*)f;
    }
    return off;
}

// one wave stages its tile of 64 anchor rows in LDS (same coalesced copy as nms_filter_kernel)
__device__ __forceinline__ int nmsg_load_tile(const float* __restrict__ pred, int img, int tile, int A, int no, float* L,
                                              int lane) {
    const int row0 = tile * NMS_TILE;
    const int nrows = min(NMS_TILE, A - row0);
    const float* src = pred + ((size_t)img * A + row0) * (size_t)no;
    const int nfl = nrows * no;
    if ((((uintptr_t)src) & 15) == 0) {
        const int nv = nfl >> 2;
        const float4* s4 = (const float4*)src;
        float4* l4 = (float4*)L;
        for (int i = lane; i < nv; i += 64) l4[i] = s4[i];
        for (int i = (nv << 2) + lane; i < nfl; i += 64) L[i] = src[i];
    } else {
        for (int i = lane; i < nfl; i += 64) L[i] = src[i];
    }
    __builtin_amdgcn_wave_barrier();
    return nrows;
}

struct NmsgRule { float thr; int multi; unsigned long long cm0, cm1; };   // class filter: bit c of cm0:cm1

__device__ __forceinline__ bool nmsg_class_ok(const NmsgRule& q, int c) {
    return c < 64 ? ((q.cm0 >> c) & 1ull) : ((q.cm1 >> (c - 64)) & 1ull);
}

// f(class, score) for every candidate of anchor row r, in the reference's order
template <typename F>
__device__ __forceinline__ void nmsg_for_each(const float* r, int no, const NmsgRule& q, F f) {
    const float obj = r[4];
    if (!(obj > q.thr)) return;
    const int nc = no - 5;
    float mx = r[5];
    for (int c = 1; c < nc; ++c) mx = fmaxf(mx, r[5 + c]);
    if (!(mx > q.thr)) return;                              // :1005
    if (q.multi) {
        for (int c = 0; c < nc; ++c) {
            const float pv = r[5 + c] * obj;                // :1046
            if (pv > q.thr && nmsg_class_ok(q, c)) f(c, pv);
        }
    } else {
        float conf = r[5] * obj;
        int j = 0;
        for (int c = 1; c < nc; ++c) {
            const float pv = r[5 + c] * obj;
            if (pv > conf) { conf = pv; j = c; }
        }
        if (conf > q.thr && nmsg_class_ok(q, j)) f(j, conf);
    }
}

// level 0: histogram of the high 16 score bits of every candidate; level 1: of the low 16 bits of the
// candidates inside the boundary bucket
__global__ __launch_bounds__(256) void nmsg_hist_kernel(const float* __restrict__ pred, int A, int no, NmsgRule q, int T,
                                                        int level, NmsgWs ws) {
    __shared__ __attribute__((aligned(16))) float lds[4][NMS_TILE * NMS_MAX_NO];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int img = blockIdx.y, tile = blockIdx.x * 4 + wave;
    if (tile >= T) return;
    const NmsgSel se = ws.sel[img];
    if (level == 1 && se.done) return;
    const int nrows = nmsg_load_tile(pred, img, tile, A, no, lds[wave], lane);
    if (lane >= nrows) return;
    unsigned* h = ws.hist + (size_t)img * NMSG_BUCKETS;
    const unsigned hi = se.thr_key >> 16;
    nmsg_for_each(lds[wave] + lane * no, no, q, [&](int, float pv) {
        const unsigned key = __float_as_uint(pv);
        if (level == 0) atomicAdd(&h[key >> 16], 1u);
        else if ((key >> 16) == hi) atomicAdd(&h[key & 0xffffu], 1u);
    });
}

// per image: walk the histogram from the top until `want` candidates are covered
__global__ __launch_bounds__(256) void nmsg_select_kernel(int level, int max_nms, NmsgWs ws) {
    __shared__ unsigned seg[256];
    __shared__ int s_bucket, s_above;
    const int img = blockIdx.x, tid = threadIdx.x;
    const unsigned* h = ws.hist + (size_t)img * NMSG_BUCKETS;
    NmsgSel se = ws.sel[img];
    if (level == 1 && se.done) return;                     // block-uniform
    unsigned mine = 0;
    for (int k = 0; k < 256; ++k) mine += h[tid * 256 + k];
    seg[tid] = mine;
    __syncthreads();
    if (tid == 0) {
        unsigned total = 0;
        for (int k = 0; k < 256; ++k) total += seg[k];
        const int want = level == 0 ? max_nms : se.need;
        if (level == 0 && (long long)total <= (long long)max_nms) {
            s_bucket = -1; s_above = (int)total;
        } else {
            unsigned above = 0;
            int sgi = 255;
            while (sgi > 0 && above + seg[sgi] < (unsigned)want) { above += seg[sgi]; --sgi; }
            int b = sgi * 256 + 255;
            while (b > sgi * 256 && above + h[b] < (unsigned)want) { above += h[b]; --b; }
            s_bucket = b; s_above = (int)above;            // buckets > b hold `above` (< want) candidates
        }
    }
    __syncthreads();
    if (tid == 0) {
        if (level == 0) {
            if (s_bucket < 0) { se.thr_key = 0; se.quota = 0; se.need = 0; se.done = 1; }
            else { se.thr_key = (unsigned)s_bucket << 16; se.need = max_nms - s_above; se.quota = 0; se.done = 0; }
        } else {
            se.thr_key = (se.thr_key & 0xffff0000u) | (unsigned)s_bucket;
            se.quota = se.need - s_above;                  // ties with the threshold key that still pass
        }
        ws.sel[img] = se;
    }
}

__global__ __launch_bounds__(256) void nmsg_count_kernel(const float* __restrict__ pred, int A, int no, NmsgRule q, int T,
                                                         NmsgWs ws) {
    __shared__ __attribute__((aligned(16))) float lds[4][NMS_TILE * NMS_MAX_NO];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int img = blockIdx.y, tile = blockIdx.x * 4 + wave;
    if (tile >= T) return;
    const NmsgSel se = ws.sel[img];
    const int nrows = nmsg_load_tile(pred, img, tile, A, no, lds[wave], lane);
    int nin = 0, ntie = 0;
    if (lane < nrows)
        nmsg_for_each(lds[wave] + lane * no, no, q, [&](int, float pv) {
            const unsigned key = __float_as_uint(pv);
            nin += key > se.thr_key ? 1 : 0;
            ntie += key == se.thr_key ? 1 : 0;
        });
    nin = et_wave_sum_i(nin);
    ntie = et_wave_sum_i(ntie);
    if (lane == 0) { ws.tile_in[(size_t)img * T + tile] = nin; ws.tile_tie[(size_t)img * T + tile] = ntie; }
}

// exclusive scans of the per-tile counts (T <= 4096), one workgroup per image
__global__ __launch_bounds__(256) void nmsg_scan_kernel(int T, NmsgWs ws) {
    __shared__ int wsum[2][4];
    const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int run0 = 0, run1 = 0;
    for (int base = 0; base < T; base += 256) {
        const int t = base + tid;
        const int v0 = t < T ? ws.tile_in[(size_t)img * T + t] : 0;
        const int v1 = t < T ? ws.tile_tie[(size_t)img * T + t] : 0;
        int i0 = v0, i1 = v1;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int a = __shfl_up(i0, d), b = __shfl_up(i1, d);
            if (lane >= d) { i0 += a; i1 += b; }
        }
        if (lane == 63) { wsum[0][wave] = i0; wsum[1][wave] = i1; }
        __syncthreads();
        int w0 = 0, w1 = 0;
        for (int w = 0; w < wave; ++w) { w0 += wsum[0][w]; w1 += wsum[1][w]; }
        const int t0 = wsum[0][0] + wsum[0][1] + wsum[0][2] + wsum[0][3];
        const int t1 = wsum[1][0] + wsum[1][1] + wsum[1][2] + wsum[1][3];
        if (t < T) {
            ws.off_in[(size_t)img * T + t] = run0 + w0 + i0 - v0;
            ws.off_tie[(size_t)img * T + t] = run1 + w1 + i1 - v1;
        }
        run0 += t0; run1 += t1;
        __syncthreads();
    }
    if (tid == 0) ws.w.ncand[img] = run0 + min(run1, ws.sel[img].quota);
}

__global__ __launch_bounds__(256) void nmsg_emit_kernel(const float* __restrict__ pred, int A, int no, NmsgRule q, int T,
                                                        int cap, NmsgWs ws) {
    __shared__ __attribute__((aligned(16))) float lds[4][NMS_TILE * NMS_MAX_NO];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int img = blockIdx.y, tile = blockIdx.x * 4 + wave;
    if (tile >= T) return;
    const NmsgSel se = ws.sel[img];
    const int nrows = nmsg_load_tile(pred, img, tile, A, no, lds[wave], lane);
    const float* r = lds[wave] + lane * no;
    int nin = 0, ntie = 0;
    if (lane < nrows)
        nmsg_for_each(r, no, q, [&](int, float pv) {
            const unsigned key = __float_as_uint(pv);
            nin += key > se.thr_key ? 1 : 0;
            ntie += key == se.thr_key ? 1 : 0;
        });
    // exclusive prefix over the lanes of this tile (candidate order = anchor, then class)
    int pin = nin, ptie = ntie;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int a = __shfl_up(pin, d), b = __shfl_up(ptie, d);
        if (lane >= d) { pin += a; ptie += b; }
    }
    int in_before = ws.off_in[(size_t)img * T + tile] + pin - nin;
    int tie_before = ws.off_tie[(size_t)img * T + tile] + ptie - ntie;
    if (lane >= nrows || nin + ntie == 0) return;
    const float hw = r[2] / 2, hh = r[3] / 2;
    const float4 box = make_float4(r[0] - hw, r[1] - hh, r[0] + hw, r[1] + hh);
    float* cand = ws.w.cand + (size_t)img * cap * NMS_REC;
    unsigned long long* keys = ws.w.keys + (size_t)img * cap;
    nmsg_for_each(r, no, q, [&](int c, float pv) {
        const unsigned key = __float_as_uint(pv);
        bool pass = key > se.thr_key;
        const bool tie = key == se.thr_key;
        if (tie) pass = tie_before < se.quota;
        if (pass) {
            const int ci = in_before + min(tie_before, se.quota);
            float4* d = (float4*)(cand + (size_t)ci * NMS_REC);
            d[0] = box;
            d[1] = make_float4(pv, (float)c, 0.f, 0.f);
            keys[ci] = ((unsigned long long)(~key) << 32) | (unsigned)ci;
        }
        if (tie) ++tie_before; else if (key > se.thr_key) ++in_before;
    });
}

extern "C" int et_nms_workspace_bytes(int B, int A, int no, int multi_label, int max_nms, size_t* bytes) {
    if (B <= 0 || A <= 0 || no < 6 || max_nms <= 0 || !bytes) return -1;
    const long long all = (long long)A * (multi_label ? (no - 5) : 1);
    const int cap = (int)(all < max_nms ? all : max_nms);
    *bytes = nmsg_carve(nullptr, B, A, cap, nullptr);
    return 0;
}

extern "C" int et_nms(const float* pred, int B, int A, int no, float conf_thres, float iou_thres, int agnostic,
                      int multi_label, uint64_t class_mask_lo, uint64_t class_mask_hi, int max_nms, float max_wh,
                      int max_det, float* dets, int* counts, int64_t* keep, int* n_candidates, void* workspace,
                      size_t ws_bytes, et_stream_t stream) {
    if (!pred || !dets || !counts || !keep || !workspace) return -1;
    if (B <= 0 || A <= 0 || no < 6 || no > NMS_MAX_NO || no - 5 > 128) return -2;
    if (max_det <= 0 || max_det > NMS_MAX_DET || max_nms <= 0) return -2;
    const int T = (A + NMS_TILE - 1) / NMS_TILE;
    if (T > 4096) return -2;
    const int multi = (multi_label && no - 5 > 1) ? 1 : 0;                      // :1017
    const long long all = (long long)A * (multi ? (no - 5) : 1);
    const int cap = (int)(all < max_nms ? all : max_nms);
    NmsgWs ws;
    if (nmsg_carve(workspace, B, A, cap, &ws) > ws_bytes) return -3;
    hipStream_t s = (hipStream_t)stream;
    NmsgRule q; q.thr = conf_thres; q.multi = multi; q.cm0 = class_mask_lo; q.cm1 = class_mask_hi;
    const dim3 tg((T + 3) / 4, B), tb(256);
    (void)hipMemsetAsync(ws.hist, 0, (size_t)B * NMSG_BUCKETS * 4, s);
    (void)hipMemsetAsync(ws.sel, 0, (size_t)B * sizeof(NmsgSel), s);
    hipLaunchKernelGGL(nmsg_hist_kernel, tg, tb, 0, s, pred, A, no, q, T, 0, ws);
    hipLaunchKernelGGL(nmsg_select_kernel, dim3(B), dim3(256), 0, s, 0, max_nms, ws);
    (void)hipMemsetAsync(ws.hist, 0, (size_t)B * NMSG_BUCKETS * 4, s);
    hipLaunchKernelGGL(nmsg_hist_kernel, tg, tb, 0, s, pred, A, no, q, T, 1, ws);
    hipLaunchKernelGGL(nmsg_select_kernel, dim3(B), dim3(256), 0, s, 1, max_nms, ws);
    hipLaunchKernelGGL(nmsg_count_kernel, tg, tb, 0, s, pred, A, no, q, T, ws);
    hipLaunchKernelGGL(nmsg_scan_kernel, dim3(B), dim3(256), 0, s, T, ws);
    hipLaunchKernelGGL(nmsg_emit_kernel, tg, tb, 0, s, pred, A, no, q, T, cap, ws);
    hipLaunchKernelGGL(nms_rank_kernel, dim3((cap + 255) / 256, B), dim3(256), 0, s, cap, ws.w);
    hipLaunchKernelGGL(nms_greedy_kernel, dim3(B), dim3(256), 0, s, cap, iou_thres, agnostic ? 0.0f : max_wh, max_det,
                       ws.w, dets, counts, (long long*)keep);
    if (n_candidates)
        (void)hipMemcpyAsync(n_candidates, ws.w.ncand, (size_t)B * sizeof(int), hipMemcpyDeviceToDevice, s);
    ET_CHECK_LAUNCH();
    return 0;
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
