// BatchNorm2d (train-mode batch statistics, eps 1e-3 / momentum 0.03 as set by
// utils/torch_utils.py:162-169) + SiLU (+ Bottleneck residual, models/backbone/common.py:544) of the
// reference's `Conv` block (common.py:471-481), forward and backward, on NHWC tensors.
//
// The conv epilogue already produced per-block partial sums of y and y^2 (et_conv2d_fwd
// stats_partial), so the forward is:  bn_finalize (tiny, per channel, fp64 combine) -> one HBM-bound
// elementwise pass  z = silu(y*scale + shift) (+ residual), 16 bytes per lane, every thread pinned to
// one channel vector so scale/shift live in registers.  Backward is the classic two passes:
// reduce (sum du, sum du*xhat) -> finalize -> apply.
#include <stdlib.h>
#include "et_device.h"
#include "../../include/et_hip.h"

enum { ACT_NONE = 0, ACT_SILU = 1, ACT_RELU = 2 };

typedef unsigned u4raw __attribute__((ext_vector_type(4)));
// Keeps a group of 16-byte loads where they were written: the use of all four registers in one (empty) asm statement
// makes the compiler issue the four loads back to back and wait once.  Without it the loads of the SECOND tensor are
// sunk to their first use, i.e. below the activation math of the first, and each is waited for on its own: three
// serialized HBM round trips per loop iteration (r04: bn_act_bwd_reduce at 4.15 TB/s of cold reads).
__device__ __forceinline__ void pin_loaded(u4raw& a, u4raw& b, u4raw& c, u4raw& d) { asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
__device__ __forceinline__ void pin_loaded(u4raw& a, u4raw& b) { asm volatile("" : "+v"(a), "+v"(b)); }

template <typename T> struct Vec16;
template <> struct Vec16<float> {
    static constexpr int N = 4;
    __device__ static __forceinline__ u4raw load_raw(const float* p) { return *(const u4raw*)p; }
    __device__ static __forceinline__ u4raw load_stream_raw(const float* p) { return *(const u4raw*)p; }
    __device__ static __forceinline__ void unpack(const u4raw t, float (&v)[4]) {
        v[0] = __uint_as_float(t.x); v[1] = __uint_as_float(t.y); v[2] = __uint_as_float(t.z); v[3] = __uint_as_float(t.w);
    }
    __device__ static __forceinline__ void load(const float* p, float (&v)[4]) {
        const float4 t = *(const float4*)p; v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    __device__ static __forceinline__ void load_stream(const float* p, float (&v)[4]) { load(p, v); }
    __device__ static __forceinline__ void store(float* p, const float (&v)[4]) {
        *(float4*)p = make_float4(v[0], v[1], v[2], v[3]);
    }
};
template <typename T> struct Vec16lp {          // the 16-bit storage formats (bf16, IEEE half): et_lp<T> converts
    static constexpr int N = 8;
    __device__ static __forceinline__ u4raw load_raw(const T* p) { return *(const u4raw*)p; }
    __device__ static __forceinline__ u4raw load_stream_raw(const T* p) { return __builtin_nontemporal_load((const u4raw*)p); }
    __device__ static __forceinline__ void unpack(const u4raw t, float (&v)[8]) {
        const unsigned w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[2 * i] = et_lp<T>::lo(w[i]); v[2 * i + 1] = et_lp<T>::hi(w[i]); }
    }
    __device__ static __forceinline__ void load(const T* p, float (&v)[8]) {
        const uint4 t = *(const uint4*)p;
        const unsigned w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[2 * i] = et_lp<T>::lo(w[i]); v[2 * i + 1] = et_lp<T>::hi(w[i]); }
    }
    // last use of the data for a long time (the apply pass of backward, y in the forward pass): a non-temporal
    // load keeps these streams from evicting what the neighbouring conv kernels re-read through L2 / MALL
    // (measured: BN micro-benchmark -2 %, whole step +0.7 %)
    __device__ static __forceinline__ void load_stream(const T* p, float (&v)[8]) {
        typedef unsigned u4v __attribute__((ext_vector_type(4)));
        const u4v t = __builtin_nontemporal_load((const u4v*)p);
        const unsigned w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[2 * i] = et_lp<T>::lo(w[i]); v[2 * i + 1] = et_lp<T>::hi(w[i]); }
    }
    __device__ static __forceinline__ void store(T* p, const float (&v)[8]) {
        *(uint4*)p = make_uint4(et_lp<T>::pack(v[0], v[1]), et_lp<T>::pack(v[2], v[3]), et_lp<T>::pack(v[4], v[5]), et_lp<T>::pack(v[6], v[7]));
    }
};
template <> struct Vec16<uint16_t> : Vec16lp<uint16_t> {};
template <> struct Vec16<et_f16> : Vec16lp<et_f16> {};

// ---- partial-sum reduction + per-channel finalize in ONE launch ---------------------------------------------
// (rows, 2, C) fp32 partial sums -> fp64 totals -> per-channel results.  grid (C/32, RB): each block reduces a
// slice of the rows for 32 channels (256 threads = 32 channels x 8 row groups) and adds into the fp64 totals
// with agent-scope atomics, so the reduction is spread over the chip instead of a handful of workgroups
// walking 10^4 rows.  The LAST block of a channel group to arrive (ticket counter per group) reads the totals
// back -- with an atomic exchange, which also leaves them zero for the next layer -- and runs the finalize
// math for its 32 channels: no separate memset / finalize launches (was 3 launches per BN layer per pass).
// `ws` = [2*C fp64 totals | C/32 int tickets], all ZERO on entry and zero again on return.
struct BnFwdFin {
    double count; const float* gamma; const float* beta; float eps, momentum;
    float* rmean; float* rvar; float* scale; float* shift; float* smean; float* sinvstd;
};
struct BnBwdFin {
    float count; const float* gamma; const float* invstd; float* dgamma; float* dbeta; float* k0; float* k1; float* k2;
    const float* mean;      // not null: the second total is sum(du * y) (conv-epilogue partials), converted here to sum(du * xhat)
};

// (sc, sh) = the folded affine of channel c from its totals; `write` = also store every per-channel output (one caller per channel)
__device__ __forceinline__ void bn_fwd_coeffs(const BnFwdFin& f, int c, double S, double Q, bool write, float& sc, float& sh) {
    const double mean = S / f.count;
    double var = Q / f.count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)f.eps));
    sc = f.gamma[c] * invstd;
    sh = f.beta[c] - (float)mean * sc;
    if (!write) return;
    f.scale[c] = sc;
    f.shift[c] = sh;
    if (f.smean) { f.smean[c] = (float)mean; f.sinvstd[c] = invstd; }
    if (f.rmean) {
        const double unb = f.count > 1.0 ? var * f.count / (f.count - 1.0) : var;
        f.rmean[c] = (1.0f - f.momentum) * f.rmean[c] + f.momentum * (float)mean;
        f.rvar[c] = (1.0f - f.momentum) * f.rvar[c] + f.momentum * (float)unb;
    }
}
__device__ __forceinline__ void bn_finalize_channel(const BnFwdFin& f, int c, double S, double Q) {
    float sc, sh;
    bn_fwd_coeffs(f, c, S, Q, true, sc, sh);
}
// dbeta += s1, dgamma += s2, coefficients of pass 2
__device__ __forceinline__ void bn_bwd_coeffs(const BnBwdFin& f, int c, double S, double Q, bool write, float& k0, float& k1, float& k2) {
    if (f.mean) Q = (double)f.invstd[c] * (Q - (double)f.mean[c] * S);
    if (write && f.dbeta) f.dbeta[c] += (float)S;
    if (write && f.dgamma) f.dgamma[c] += (float)Q;
    k0 = f.gamma[c] * f.invstd[c];
    k1 = (float)(S / f.count);
    k2 = (float)(Q / f.count);
}
__device__ __forceinline__ void bn_finalize_channel(const BnBwdFin& f, int c, double S, double Q) {
    float k0, k1, k2;
    bn_bwd_coeffs(f, c, S, Q, true, k0, k1, k2);
    f.k0[c] = k0; f.k1[c] = k1; f.k2[c] = k2;
}
// Sharded sums (et_hip.h, et_conv2d_fwd stats_ld > 0): [ET_BN_SHARDS][2][ld] fp32, folded in fp64 in shard order
struct BnShardFwd { const float* shards; int ld; BnFwdFin fin; };
struct BnShardBwd { const float* shards; int ld; BnBwdFin fin; };
constexpr int BN_SHARD_MAX_C = 1024;
__device__ __forceinline__ void bn_fold_shards(const float* shards, int ld, int c, double& S, double& Q) {
    S = 0.0; Q = 0.0;
#pragma unroll
    for (int s = 0; s < ET_BN_SHARDS; ++s) { S += (double)shards[(size_t)(2 * s) * ld + c]; Q += (double)shards[(size_t)(2 * s + 1) * ld + c]; }
}

template <typename FIN>
__global__ __launch_bounds__(256) void rows_reduce_finalize_kernel(const float* __restrict__ part, int rows, int C,
                                                                   int rows_per_block, double* __restrict__ ws, FIN fin) {
    __shared__ double red[2][8][32];
    __shared__ int s_last;
    const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    double s = 0.0, q = 0.0;
    if (c < C) {
        // four rows in flight per thread: with one, a block's 50-400 rows per row group were a serial chain of L2 latencies
        // (21 us for the 25600 partial rows of a 160x160 layer; the data is 13 MB)
        int r = r0 + rg;
        for (; r + 24 < r1; r += 32) {
            const float s0 = part[((size_t)r * 2 + 0) * C + c], q0 = part[((size_t)r * 2 + 1) * C + c];
            const float s1 = part[((size_t)(r + 8) * 2 + 0) * C + c], q1 = part[((size_t)(r + 8) * 2 + 1) * C + c];
            const float s2 = part[((size_t)(r + 16) * 2 + 0) * C + c], q2 = part[((size_t)(r + 16) * 2 + 1) * C + c];
            const float s3 = part[((size_t)(r + 24) * 2 + 0) * C + c], q3 = part[((size_t)(r + 24) * 2 + 1) * C + c];
            s += ((double)s0 + (double)s1) + ((double)s2 + (double)s3);
            q += ((double)q0 + (double)q1) + ((double)q2 + (double)q3);
        }
        for (; r < r1; r += 8) {
            s += (double)part[((size_t)r * 2 + 0) * C + c];
            q += (double)part[((size_t)r * 2 + 1) * C + c];
        }
    }
    red[0][rg][cl] = s; red[1][rg][cl] = q;
    __syncthreads();
    if (rg == 0 && c < C) {
        for (int k = 1; k < 8; ++k) { s += red[0][k][cl]; q += red[1][k][cl]; }
        atomicAdd(ws + c, s);
        atomicAdd(ws + C + c, q);
        __threadfence();                                   // my sums are visible before my ticket is
    }
    __syncthreads();
    int* tickets = (int*)(ws + 2 * C);
    if (threadIdx.x == 0) s_last = atomicAdd(&tickets[blockIdx.x], 1) == (int)gridDim.y - 1;
    __syncthreads();
    if (!s_last) return;                                   // block-uniform
    if (threadIdx.x == 0) tickets[blockIdx.x] = 0;         // zero again for the next layer
    if (rg == 0 && c < C) {
        __threadfence();
        // read-and-clear at agent scope: never served from this CU's / XCD's stale cache lines
        const double S = __longlong_as_double((long long)atomicExch((unsigned long long*)(ws + c), 0ull));
        const double Q = __longlong_as_double((long long)atomicExch((unsigned long long*)(ws + C + c), 0ull));
        bn_finalize_channel(fin, c, S, Q);
    }
}

// Few partial rows (the 40x40 and 20x20 maps: <= 1600 rows): one 1024-thread block per 16 channels walks ALL rows (64 row groups,
// four rows in flight per thread) and finalizes -- no fp64 atomics, no fences, no ticket.  The distributed form above spends most of
// its 10-14 us in four dependent L2 round trips (atomic, fence, ticket, exchange); on these sizes there is nothing to distribute.
template <typename FIN>
__global__ __launch_bounds__(1024) void rows_reduce_finalize_small_kernel(const float* __restrict__ part, int rows, int C, FIN fin) {
    __shared__ double red[2][64][16];
    const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    double s = 0.0, q = 0.0;
    if (c < C) {
        int r = rg;
        // sixteen rows (32 loads) in flight per thread, rows beyond the end predicated off: 1024 rows per trip of the block, so the
        // 200 / 800-row layers are ONE L2 round trip instead of one per four rows (this kernel is nothing but dependent latencies:
        // launch, loads, LDS, the fp64 finalize)
        for (; r < rows; r += 16 * 64) {
            float sv[16], qv[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int rr = min(r + 64 * k, rows - 1);           // UNCONDITIONAL loads (a select around a load compiles to a branch
                sv[k] = part[((size_t)rr * 2 + 0) * C + c];         // with its own wait: 32 serialized round trips, measured +2 ms per step)
                qv[k] = part[((size_t)rr * 2 + 1) * C + c];
            }
            double ts[4] = {0.0, 0.0, 0.0, 0.0}, tq[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const bool ok = r + 64 * k < rows;
                ts[k & 3] += ok ? (double)sv[k] : 0.0; tq[k & 3] += ok ? (double)qv[k] : 0.0;
            }
            s += (ts[0] + ts[1]) + (ts[2] + ts[3]);
            q += (tq[0] + tq[1]) + (tq[2] + tq[3]);
        }
    }
    red[0][rg][cl] = s; red[1][rg][cl] = q;
    __syncthreads();
    if (rg < 2 && c < C) {                                  // rg 0: the sums, rg 1: the second totals; 64 partials each
        double t = 0.0;
        for (int k = 0; k < 64; ++k) t += red[rg][k][cl];
        red[rg][0][cl] = t;
    }
    __syncthreads();
    if (rg == 0 && c < C) bn_finalize_channel(fin, c, red[0][0][cl], red[1][0][cl]);
}

template <typename FIN>
static void launch_rows_reduce_finalize(const float* part, int rows, int C, double* ws, const FIN& fin, hipStream_t s) {
    const char* const fe = getenv("ET_BN_FIN_SMALL");         // threshold in partial rows (0: always the distributed form); read per call: tests run both
    const int small_rows = fe ? atoi(fe) : 2048;
    if (rows <= small_rows) {
        hipLaunchKernelGGL((rows_reduce_finalize_small_kernel<FIN>), dim3((C + 15) / 16), dim3(1024), 0, s, part, rows, C, fin);
        return;
    }
    int rb = rows / 64;
    rb = rb < 1 ? 1 : (rb > 128 ? 128 : rb);
    const int per = (rows + rb - 1) / rb;
    hipLaunchKernelGGL((rows_reduce_finalize_kernel<FIN>), dim3((C + 31) / 32, (rows + per - 1) / per), dim3(256), 0, s, part,
                       rows, C, per, ws, fin);
}

// eval-mode affine from running statistics (teacher path): scale = g/sqrt(rv+eps), shift = b - rm*scale
__global__ __launch_bounds__(256) void bn_eval_affine_kernel(int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const float* __restrict__ rmean, const float* __restrict__ rvar,
                                                             float eps, float* __restrict__ scale, float* __restrict__ shift) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float sc = gamma[c] / sqrtf(rvar[c] + eps);
    scale[c] = sc;
    shift[c] = beta[c] - rmean[c] * sc;
}

// SiLU with the hardware exp / rcp (v_exp_f32, v_rcp_f32: ~1 ulp): these passes are HBM-bound only as
// long as the per-element VALU work stays small; an IEEE division costs ~10 extra instructions.
// The activation is a template parameter: a run-time switch inside the per-element loops compiles to scalar
// branches around every element and serialises the loads behind them (measured: the reduce pass ran at
// 2-3 TB/s with the switch, the apply pass at 5-6 TB/s).
template <int ACT> __device__ __forceinline__ float act_fwd(float u) {
    if constexpr (ACT == ACT_SILU) return u * __builtin_amdgcn_rcpf(1.0f + __expf(-u));
    else if constexpr (ACT == ACT_RELU) return fmaxf(u, 0.f);
    else return u;
}
template <int ACT> __device__ __forceinline__ float act_grad(float u) {
    if constexpr (ACT == ACT_SILU) { const float s = __builtin_amdgcn_rcpf(1.0f + __expf(-u)); return s * (1.0f + u * (1.0f - s)); }
    else if constexpr (ACT == ACT_RELU) return u > 0.f ? 1.f : 0.f;
    else return 1.f;
}
// host: instantiate `KERNEL<T, ACT>` for the run-time (dtype, act) pair
#define ET_ACT_LAUNCH(KERNEL, T, act, ...)                                                                    \
    do {                                                                                                      \
        if ((act) == ACT_SILU) hipLaunchKernelGGL((KERNEL<T, ACT_SILU>), __VA_ARGS__);                        \
        else if ((act) == ACT_RELU) hipLaunchKernelGGL((KERNEL<T, ACT_RELU>), __VA_ARGS__);                   \
        else hipLaunchKernelGGL((KERNEL<T, ACT_NONE>), __VA_ARGS__);                                          \
    } while (0)

// The pixels one thread visits: p0 = gt / CV, p0 + pstep, ... below P (pstep = threads of the grid / CV; the grid is sized so
// that CV divides it).  32-bit arithmetic: a 64-bit division is ~150 instructions per thread, a quarter of the work of a thread
// that handles eight vectors.  (r04: walking the tensor from its END -- to start on the bytes the previous pass over it touched
// last -- changes nothing, profiles/r04_bn_reduce_pin_and_reverse_walk_ab.txt: the memory-side cache does not keep streamed data.)
struct PixelWalk { long long p, step; int n; };
__device__ __forceinline__ PixelWalk pixel_walk(int P, int CV) {
    const unsigned gt = blockIdx.x * 256u + threadIdx.x;
    const unsigned p0 = gt / (unsigned)CV, ps = (gridDim.x * 256u) / (unsigned)CV;
    PixelWalk w;
    w.n = p0 < (unsigned)P ? (int)(((unsigned)P - 1u - p0) / ps) + 1 : 0;
    w.p = (long long)p0;
    w.step = (long long)ps;
    return w;
}

template <typename T, int ACT>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const T* __restrict__ y, int ldy, T* __restrict__ z, int ldz,
                                                         const T* __restrict__ res, int ldr, int P, int CV,
                                                         const float* __restrict__ scale, const float* __restrict__ shift, BnShardFwd q) {
    constexpr int N = Vec16<T>::N;
    const int cv = (int)((blockIdx.x * 256u + threadIdx.x) % (unsigned)CV);
    const PixelWalk w = pixel_walk(P, CV);
    long long p = w.p;
    float sc[N], sh[N];
    if (q.shards) {
        // sharded sums: this workgroup derives the affine of every channel itself (a workgroup's 256 consecutive thread ids reach every
        // channel vector, CV <= 256): one channel per thread, 16 coalesced floats each, fp64 -- 64 B of L2 reads per channel against
        // the >= 64 KB of HBM traffic of the workgroup's stream.  Workgroup 0 publishes the per-channel results.
        __shared__ float cst[2][BN_SHARD_MAX_C];
        const int C = CV * N;
        for (int c = threadIdx.x; c < C; c += 256) {
            double S, Q;
            bn_fold_shards(q.shards, q.ld, c, S, Q);
            bn_fwd_coeffs(q.fin, c, S, Q, blockIdx.x == 0, cst[0][c], cst[1][c]);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < N; ++i) { sc[i] = cst[0][cv * N + i]; sh[i] = cst[1][cv * N + i]; }
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) { sc[i] = scale[cv * N + i]; sh[i] = shift[cv * N + i]; }
    }
    for (int it = 0; it < w.n; ++it, p += w.step) {
        float v[N];
        u4raw ry = Vec16<T>::load_stream_raw(y + p * ldy + cv * N), rr = ry;
        if (res) { rr = Vec16<T>::load_raw(res + p * ldr + cv * N); pin_loaded(ry, rr); }   // both loads in flight before the math
        Vec16<T>::unpack(ry, v);
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = act_fwd<ACT>(v[i] * sc[i] + sh[i]);
        if (res) {
            float r[N];
            Vec16<T>::unpack(rr, r);
            // an ADD with its own rounding, as the reference's `x + self.cv2(self.cv1(x))` (common.py:544) -- with the residual in
            // registers before the activation the compiler contracts u * sigmoid(u) + r into one fma, which moved fp32 activations by
            // an ulp and, through ~100 train-mode BatchNorm layers at random init, some fp32-mode gradients by 1 % (r04,
            // tests/test_step_benchbatch.py: worst relative L2 vs the oracle 3.6e-3 -> 1.16e-2)
#pragma unroll
            for (int i = 0; i < N; ++i) v[i] = __fadd_rn(v[i], r[i]);
        }
        Vec16<T>::store(z + p * ldz + cv * N, v);
    }
}

// ---- backward -----------------------------------------------------------------------------------------
// pass 1: per-block partial sums of du and du*xhat, du = dz * act'(y*scale+shift), xhat = (y-mean)*invstd
template <typename T, int ACT>
__global__ __launch_bounds__(256) void bn_act_bwd_reduce_kernel(const T* __restrict__ dz, int lddz, const T* __restrict__ y, int ldy,
                                                                int P, int CV, const float* __restrict__ scale,
                                                                const float* __restrict__ shift, const float* __restrict__ mean,
                                                                const float* __restrict__ invstd, float* __restrict__ part, int part_ld) {
    constexpr int N = Vec16<T>::N;
    __shared__ float acc[2][2048];
    const int C = CV * N;
    for (int i = threadIdx.x; i < C; i += 256) { acc[0][i] = 0.f; acc[1][i] = 0.f; }
    __syncthreads();
    const int cv = (int)((blockIdx.x * 256u + threadIdx.x) % (unsigned)CV);
    const PixelWalk w = pixel_walk(P, CV);
    long long p = w.p;
    const long long pstep = w.step;
    int it = 0;
    // accumulate s1 = sum(du) and s2r = sum(du * y); sum(du * xhat) = invstd * (s2r - mean * s1) is formed once
    // per thread at the end, so mean / invstd stay out of the streaming loop (fewer live registers)
    float sc[N], sh[N], s1[N], s2[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        sc[i] = scale[cv * N + i]; sh[i] = shift[cv * N + i];
        s1[i] = 0.f; s2[i] = 0.f;
    }
    for (; it + 1 < w.n; it += 2, p += 2 * pstep) {  // two pixels in flight per thread
        float g[N], v[N], g2[N], v2[N];
        u4raw rg = Vec16<T>::load_raw(dz + p * lddz + cv * N), rv = Vec16<T>::load_raw(y + p * ldy + cv * N);
        u4raw rg2 = Vec16<T>::load_raw(dz + (p + pstep) * lddz + cv * N), rv2 = Vec16<T>::load_raw(y + (p + pstep) * ldy + cv * N);
        pin_loaded(rg, rv, rg2, rv2);
        Vec16<T>::unpack(rg, g); Vec16<T>::unpack(rv, v); Vec16<T>::unpack(rg2, g2); Vec16<T>::unpack(rv2, v2);
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const float du = g[i] * act_grad<ACT>(v[i] * sc[i] + sh[i]);
            const float du2 = g2[i] * act_grad<ACT>(v2[i] * sc[i] + sh[i]);
            s1[i] += du + du2;
            s2[i] += du * v[i] + du2 * v2[i];
        }
    }
    if (it < w.n) {
        float g[N], v[N];
        Vec16<T>::load(dz + p * lddz + cv * N, g);
        Vec16<T>::load(y + p * ldy + cv * N, v);
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const float du = g[i] * act_grad<ACT>(v[i] * sc[i] + sh[i]);
            s1[i] += du;
            s2[i] += du * v[i];
        }
    }
    // block reduction.  LDS layout acc[.][i*CV + cv]: the 64 lanes of a wave hit consecutive banks (the
    // channel-major layout cv*N+i is an 8-way bank conflict on every ds_add, which made this tail -- 8.4 M
    // conflicted LDS atomics per launch -- cost more than the streaming loop on the small maps).  When CV is
    // a power of two below 64 the lanes that share a channel group (lane % CV) are first summed with
    // xor-shuffles, so each wave issues one conflict-free atomic per address.
    const int lane = threadIdx.x & 63;
    const bool fold = CV < 64 && (CV & (CV - 1)) == 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const float mu = mean[cv * N + i], is = invstd[cv * N + i];
        float a = s1[i], b = is * (s2[i] - mu * s1[i]);
        if (fold) {
            for (int m = 32; m >= CV; m >>= 1) { a += __shfl_xor(a, m); b += __shfl_xor(b, m); }
        }
        if (!fold || lane < CV) {
            atomicAdd(&acc[0][i * CV + cv], a);
            atomicAdd(&acc[1][i * CV + cv], b);
        }
    }
    __syncthreads();
    if (part_ld) {       // sharded accumulator [ET_BN_SHARDS][2][part_ld], zero before the launch
        float* const d = part + (size_t)(blockIdx.x % ET_BN_SHARDS) * 2 * part_ld;
        for (int c = threadIdx.x; c < C; c += 256) {
            const int j = (c % N) * CV + c / N;
            unsafeAtomicAdd(d + c, acc[0][j]);
            unsafeAtomicAdd(d + part_ld + c, acc[1][j]);
        }
        return;
    }
    for (int c = threadIdx.x; c < C; c += 256) {
        const int j = (c % N) * CV + c / N;
        part[((size_t)blockIdx.x * 2 + 0) * C + c] = acc[0][j];
        part[((size_t)blockIdx.x * 2 + 1) * C + c] = acc[1][j];
    }
}

// pass 2: dy = k0 * (du - k1 - xhat*k2)
template <typename T, int ACT>
__global__ __launch_bounds__(256) void bn_act_bwd_apply_kernel(const T* __restrict__ dz, int lddz, const T* __restrict__ y, int ldy,
                                                               T* __restrict__ dy, int lddy, int P, int CV,
                                                               const float* __restrict__ scale, const float* __restrict__ shift,
                                                               const float* __restrict__ mean, const float* __restrict__ invstd,
                                                               const float* __restrict__ k0, const float* __restrict__ k1,
                                                               const float* __restrict__ k2, BnShardBwd q) {
    constexpr int N = Vec16<T>::N;
    const int cv = (int)((blockIdx.x * 256u + threadIdx.x) % (unsigned)CV);
    const PixelWalk w = pixel_walk(P, CV);
    long long p = w.p;
    // dy = k0*(du - k1 - xhat*k2), xhat = (y-mean)*invstd  ==  A*du + B*y + D with per-channel A, B, D
    float sc[N], sh[N], A[N], B[N], D[N];
    if (q.shards) {          // sharded sums: the coefficients are derived per workgroup (bn_act_fwd_kernel), workgroup 0 owns dgamma / dbeta
        __shared__ float cst[3][BN_SHARD_MAX_C];
        const int C = CV * N;
        for (int c = threadIdx.x; c < C; c += 256) {
            double S, Q;
            bn_fold_shards(q.shards, q.ld, c, S, Q);
            bn_bwd_coeffs(q.fin, c, S, Q, blockIdx.x == 0, cst[0][c], cst[1][c], cst[2][c]);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int c = cv * N + i;
            sc[i] = scale[c]; sh[i] = shift[c];
            const float a0 = cst[0][c], w = invstd[c] * cst[2][c];
            A[i] = a0; B[i] = -a0 * w; D[i] = a0 * (mean[c] * w - cst[1][c]);
        }
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int c = cv * N + i;
            sc[i] = scale[c]; sh[i] = shift[c];
            const float a0 = k0[c], w = invstd[c] * k2[c];
            A[i] = a0; B[i] = -a0 * w; D[i] = a0 * (mean[c] * w - k1[c]);
        }
    }
    for (int it = 0; it < w.n; ++it, p += w.step) {
        float g[N], v[N];
        Vec16<T>::load_stream(dz + p * lddz + cv * N, g);
        Vec16<T>::load_stream(y + p * ldy + cv * N, v);
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const float du = g[i] * act_grad<ACT>(v[i] * sc[i] + sh[i]);
            g[i] = A[i] * du + (B[i] * v[i] + D[i]);
        }
        Vec16<T>::store(dy + p * lddy + cv * N, g);
    }
}

// plain activation backward (netD ReLU between its two 1x1 convs, yolo_ssod.py:231-238): dy = dz * act'(y)
template <typename T, int ACT>
__global__ __launch_bounds__(256) void act_bwd_kernel(const T* __restrict__ dz, int lddz, const T* __restrict__ y, int ldy,
                                                      T* __restrict__ dy, int lddy, int P, int CV) {
    constexpr int N = Vec16<T>::N;
    // 32-bit index arithmetic (the grid is at most 2048 x 256 threads): a 64-bit division here is ~150 instructions per thread,
    // a quarter of the work of a thread that handles eight vectors
    const unsigned gt = blockIdx.x * 256u + threadIdx.x;
    const int cv = (int)(gt % (unsigned)CV);
    long long p = gt / (unsigned)CV;
    const long long pstep = (gridDim.x * 256u) / (unsigned)CV;
    for (; p < P; p += pstep) {
        float g[N], v[N];
        Vec16<T>::load(dz + p * lddz + cv * N, g);
        Vec16<T>::load(y + p * ldy + cv * N, v);
#pragma unroll
        for (int i = 0; i < N; ++i) g[i] *= act_grad<ACT>(v[i]);
        Vec16<T>::store(dy + p * lddy + cv * N, g);
    }
}

// ---- host ------------------------------------------------------------------------------------------------
// number of 256-thread blocks such that CV divides blocks*256 and the grid is ~4 waves of the chip
static int ew_blocks(long long P, int CV, int vpt_default = 8) {
    int a = CV, b = 256;
    while (b) { const int t = a % b; a = b; b = t; }
    const int unit = CV / a;                                   // blocks must be a multiple of this
    const long long need = (P * CV + 255) / 256;
    // ~8+ vectors per thread (amortises the per-thread channel constants and the partial-sum rows),
    // but never fewer than ~2 blocks per CU
    long long blocks = need / vpt_default;          // (vectors per thread was a knob until r02: 1 / 2 / 4 / 8 / 16 / 32 swept, 8 / 16 kept)
    if (blocks < 512) blocks = need < 512 ? need : 512;
    if (blocks > 2048) blocks = 2048;
    blocks = ((blocks + unit - 1) / unit) * unit;
    return (int)blocks;
}

extern "C" int et_bn_reduce_rows(int P, int C, int dtype) {
    const int vec = dtype == ET_F32 ? 4 : 8;
    return ew_blocks(P, C / vec, 16);
}

extern "C" int et_bn_finalize(const float* stats_partial, int rows, int C, double count, const float* gamma,
                              const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                              float* scale, float* shift, float* save_mean, float* save_invstd, double* ws,
                              et_stream_t stream) {
    if (!stats_partial || !gamma || !beta || !scale || !shift || !ws) return -1;
    if (rows <= 0 || C <= 0 || count <= 0) return -2;
    const BnFwdFin fin{count, gamma, beta, eps, momentum, running_mean, running_var, scale, shift, save_mean, save_invstd};
    launch_rows_reduce_finalize(stats_partial, rows, C, ws, fin, (hipStream_t)stream);
    ET_CHECK_LAUNCH();
    return 0;
}

extern "C" int et_bn_eval_affine(int C, const float* gamma, const float* beta, const float* running_mean,
                                 const float* running_var, float eps, float* scale, float* shift, et_stream_t stream) {
    if (!gamma || !beta || !running_mean || !running_var || !scale || !shift) return -1;
    hipLaunchKernelGGL(bn_eval_affine_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, C, gamma, beta,
                       running_mean, running_var, eps, scale, shift);
    ET_CHECK_LAUNCH();
    return 0;
}

extern "C" int et_bn_act_fwd(const void* y, int ldy, void* z, int ldz, const void* residual, int ldr, int dtype, int P,
                             int C, const float* scale, const float* shift, int act, et_stream_t stream) {
    if (!y || !z || !scale || !shift) return -1;
    const int vec = dtype == ET_F32 ? 4 : 8;
    if (P <= 0 || C <= 0 || C % vec || ldy % vec || ldz % vec || (residual && ldr % vec)) return -2;
    const int CV = C / vec;
    const dim3 grid(ew_blocks(P, CV));
    if (dtype == ET_F32)
        ET_ACT_LAUNCH(bn_act_fwd_kernel, float, act, grid, dim3(256), 0, (hipStream_t)stream, (const float*)y, ldy, (float*)z, ldz,
                      (const float*)residual, ldr, P, CV, scale, shift, BnShardFwd{});
    else if (dtype == ET_BF16)
        ET_ACT_LAUNCH(bn_act_fwd_kernel, uint16_t, act, grid, dim3(256), 0, (hipStream_t)stream, (const uint16_t*)y, ldy,
                      (uint16_t*)z, ldz, (const uint16_t*)residual, ldr, P, CV, scale, shift, BnShardFwd{});
    else if (dtype == ET_F16)
        ET_ACT_LAUNCH(bn_act_fwd_kernel, et_f16, act, grid, dim3(256), 0, (hipStream_t)stream, (const et_f16*)y, ldy,
                      (et_f16*)z, ldz, (const et_f16*)residual, ldr, P, CV, scale, shift, BnShardFwd{});
    else return -2;
    ET_CHECK_LAUNCH();
    return 0;
}

extern "C" int et_bn_act_bwd(const void* dz, int lddz, const void* y, int ldy, void* dy, int lddy, int dtype, int P, int C,
                             const float* gamma, const float* scale, const float* shift, const float* save_mean,
                             const float* save_invstd, int act, float* dgamma, float* dbeta, double* totals,
                             float* workspace, size_t ws_floats, et_stream_t stream) {
    // totals: 2*C fp64, zero on entry, zero again on return.  workspace (fp32 units): rows*2*C partial sums +
    // 3*C coefficients
    if (!dz || !y || !dy || !gamma || !scale || !shift || !save_mean || !save_invstd || !workspace || !totals) return -1;
    const int vec = dtype == ET_F32 ? 4 : 8;
    if (P <= 0 || C <= 0 || C > 2048 || C % vec || lddz % vec || ldy % vec || lddy % vec) return -2;
    const int CV = C / vec;
    // 16 vectors per thread: the per-block partial rows and the block reduction amortise better (measured)
    const int rows = ew_blocks(P, CV, 16);
    if (ws_floats < (size_t)rows * 2 * C + 3 * (size_t)C || (((uintptr_t)totals) & 7)) return -3;
    double* tot = totals;
    float* part = workspace;
    float* k0 = part + (size_t)rows * 2 * C;
    float* k1 = k0 + C;
    float* k2 = k1 + C;
    const dim3 grid(rows);
    if (dtype == ET_F32)
        ET_ACT_LAUNCH(bn_act_bwd_reduce_kernel, float, act, grid, dim3(256), 0, (hipStream_t)stream, (const float*)dz, lddz,
                      (const float*)y, ldy, P, CV, scale, shift, save_mean, save_invstd, part, 0);
    else if (dtype == ET_BF16)
        ET_ACT_LAUNCH(bn_act_bwd_reduce_kernel, uint16_t, act, grid, dim3(256), 0, (hipStream_t)stream, (const uint16_t*)dz, lddz,
                      (const uint16_t*)y, ldy, P, CV, scale, shift, save_mean, save_invstd, part, 0);
    else if (dtype == ET_F16)
        ET_ACT_LAUNCH(bn_act_bwd_reduce_kernel, et_f16, act, grid, dim3(256), 0, (hipStream_t)stream, (const et_f16*)dz, lddz,
                      (const et_f16*)y, ldy, P, CV, scale, shift, save_mean, save_invstd, part, 0);
    else return -2;
    const BnBwdFin fin{(float)P, gamma, save_invstd, dgamma, dbeta, k0, k1, k2, nullptr};
    launch_rows_reduce_finalize(part, rows, C, tot, fin, (hipStream_t)stream);
    if (dtype == ET_F32)
        ET_ACT_LAUNCH(bn_act_bwd_apply_kernel, float, act, grid, dim3(256), 0, (hipStream_t)stream, (const float*)dz, lddz,
                      (const float*)y, ldy, (float*)dy, lddy, P, CV, scale, shift, save_mean, save_invstd, k0, k1, k2, BnShardBwd{});
    else if (dtype == ET_BF16)
        ET_ACT_LAUNCH(bn_act_bwd_apply_kernel, uint16_t, act, grid, dim3(256), 0, (hipStream_t)stream, (const uint16_t*)dz, lddz,
                      (const uint16_t*)y, ldy, (uint16_t*)dy, lddy, P, CV, scale, shift, save_mean, save_invstd, k0, k1, k2, BnShardBwd{});
    else if (dtype == ET_F16)
        ET_ACT_LAUNCH(bn_act_bwd_apply_kernel, et_f16, act, grid, dim3(256), 0, (hipStream_t)stream, (const et_f16*)dz, lddz,
                      (const et_f16*)y, ldy, (et_f16*)dy, lddy, P, CV, scale, shift, save_mean, save_invstd, k0, k1, k2, BnShardBwd{});
    ET_CHECK_LAUNCH();
    return 0;
}

extern "C" int et_bn_act_bwd_from_partials(const void* dz, int lddz, const void* y, int ldy, void* dy, int lddy, int dtype, int P,
                                           int C, const float* gamma, const float* scale, const float* shift,
                                           const float* save_mean, const float* save_invstd, int act, float* dgamma,
                                           float* dbeta, double* totals, const float* partials, int partial_rows,
                                           float* workspace /* 3*C floats */, et_stream_t stream) {
    // the reduce pass has been done by the producer of dz (et_conv2d_dgrad_bn): partials (partial_rows, 2, C) hold per-tile
    // sums of du and du*y.  Finalize (one small launch) + the apply pass only.
    if (!dz || !y || !dy || !gamma || !scale || !shift || !save_mean || !save_invstd || !workspace || !totals || !partials) return -1;
    const int vec = dtype == ET_F32 ? 4 : 8;
    if (P <= 0 || C <= 0 || C > 2048 || C % vec || lddz % vec || ldy % vec || lddy % vec || partial_rows <= 0) return -2;
    if (((uintptr_t)totals) & 7) return -3;
    const int CV = C / vec;
    float* k0 = workspace;
    float* k1 = k0 + C;
    float* k2 = k1 + C;
    const BnBwdFin fin{(float)P, gamma, save_invstd, dgamma, dbeta, k0, k1, k2, save_mean};
    launch_rows_reduce_finalize(partials, partial_rows, C, totals, fin, (hipStream_t)stream);
    const dim3 grid(ew_blocks(P, CV, 16));
    if (dtype == ET_F32)
        ET_ACT_LAUNCH(bn_act_bwd_apply_kernel, float, act, grid, dim3(256), 0, (hipStream_t)stream, (const float*)dz, lddz,
                      (const float*)y, ldy, (float*)dy, lddy, P, CV, scale, shift, save_mean, save_invstd, k0, k1, k2, BnShardBwd{});
    else if (dtype == ET_BF16)
        ET_ACT_LAUNCH(bn_act_bwd_apply_kernel, uint16_t, act, grid, dim3(256), 0, (hipStream_t)stream, (const uint16_t*)dz, lddz,
                      (const uint16_t*)y, ldy, (uint16_t*)dy, lddy, P, CV, scale, shift, save_mean, save_invstd, k0, k1, k2, BnShardBwd{});
    else if (dtype == ET_F16)
        ET_ACT_LAUNCH(bn_act_bwd_apply_kernel, et_f16, act, grid, dim3(256), 0, (hipStream_t)stream, (const et_f16*)dz, lddz,
                      (const et_f16*)y, ldy, (et_f16*)dy, lddy, P, CV, scale, shift, save_mean, save_invstd, k0, k1, k2, BnShardBwd{});
    else return -2;
    ET_CHECK_LAUNCH();
    return 0;
}

extern "C" int et_bn_act_fwd_sharded(const void* y, int ldy, void* z, int ldz, const void* residual, int ldr, int dtype, int P, int C,
                                     const float* shards, int shard_ld, double count, const float* gamma, const float* beta, float eps,
                                     float momentum, float* running_mean, float* running_var, float* scale, float* shift,
                                     float* save_mean, float* save_invstd, int act, et_stream_t stream) {
    if (!y || !z || !shards || !gamma || !beta || !scale || !shift) return -1;
    const int vec = dtype == ET_F32 ? 4 : 8;
    if (P <= 0 || C <= 0 || C > BN_SHARD_MAX_C || C % vec || ldy % vec || ldz % vec || (residual && ldr % vec) || shard_ld < C || count <= 0) return -2;
    if ((running_mean == nullptr) != (running_var == nullptr) || (save_mean == nullptr) != (save_invstd == nullptr)) return -2;
    const int CV = C / vec;
    const dim3 grid(ew_blocks(P, CV));
    const BnShardFwd q{shards, shard_ld, BnFwdFin{count, gamma, beta, eps, momentum, running_mean, running_var, scale, shift, save_mean, save_invstd}};
    if (dtype == ET_F32)
        ET_ACT_LAUNCH(bn_act_fwd_kernel, float, act, grid, dim3(256), 0, (hipStream_t)stream, (const float*)y, ldy, (float*)z, ldz,
                      (const float*)residual, ldr, P, CV, nullptr, nullptr, q);
    else if (dtype == ET_BF16)
        ET_ACT_LAUNCH(bn_act_fwd_kernel, uint16_t, act, grid, dim3(256), 0, (hipStream_t)stream, (const uint16_t*)y, ldy,
                      (uint16_t*)z, ldz, (const uint16_t*)residual, ldr, P, CV, nullptr, nullptr, q);
    else if (dtype == ET_F16)
        ET_ACT_LAUNCH(bn_act_fwd_kernel, et_f16, act, grid, dim3(256), 0, (hipStream_t)stream, (const et_f16*)y, ldy,
                      (et_f16*)z, ldz, (const et_f16*)residual, ldr, P, CV, nullptr, nullptr, q);
    else return -2;
    ET_CHECK_LAUNCH();
    return 0;
}

extern "C" int et_bn_act_bwd_sharded(const void* dz, int lddz, const void* y, int ldy, void* dy, int lddy, int dtype, int P, int C,
                                     const float* gamma, const float* scale, const float* shift, const float* save_mean,
                                     const float* save_invstd, int act, float* dgamma, float* dbeta, float* shards, int shard_ld,
                                     int reduce, et_stream_t stream) {
    if (!dz || !y || !dy || !gamma || !scale || !shift || !save_mean || !save_invstd || !shards) return -1;
    const int vec = dtype == ET_F32 ? 4 : 8;
    if (P <= 0 || C <= 0 || C > BN_SHARD_MAX_C || C % vec || lddz % vec || ldy % vec || lddy % vec || shard_ld < C) return -2;
    const int CV = C / vec;
    const dim3 grid(ew_blocks(P, CV, 16));
    if (reduce) {
        if (dtype == ET_F32)
            ET_ACT_LAUNCH(bn_act_bwd_reduce_kernel, float, act, grid, dim3(256), 0, (hipStream_t)stream, (const float*)dz, lddz,
                          (const float*)y, ldy, P, CV, scale, shift, save_mean, save_invstd, shards, shard_ld);
        else if (dtype == ET_BF16)
            ET_ACT_LAUNCH(bn_act_bwd_reduce_kernel, uint16_t, act, grid, dim3(256), 0, (hipStream_t)stream, (const uint16_t*)dz, lddz,
                          (const uint16_t*)y, ldy, P, CV, scale, shift, save_mean, save_invstd, shards, shard_ld);
        else if (dtype == ET_F16)
            ET_ACT_LAUNCH(bn_act_bwd_reduce_kernel, et_f16, act, grid, dim3(256), 0, (hipStream_t)stream, (const et_f16*)dz, lddz,
                          (const et_f16*)y, ldy, P, CV, scale, shift, save_mean, save_invstd, shards, shard_ld);
        else return -2;
    }
    // the reduce pass leaves sums of du * xhat; a conv epilogue leaves sums of du * y (converted with the saved mean / invstd)
    const BnShardBwd q{shards, shard_ld, BnBwdFin{(float)P, gamma, save_invstd, dgamma, dbeta, nullptr, nullptr, nullptr, reduce ? nullptr : save_mean}};
    if (dtype == ET_F32)
        ET_ACT_LAUNCH(bn_act_bwd_apply_kernel, float, act, grid, dim3(256), 0, (hipStream_t)stream, (const float*)dz, lddz,
                      (const float*)y, ldy, (float*)dy, lddy, P, CV, scale, shift, save_mean, save_invstd, nullptr, nullptr, nullptr, q);
    else if (dtype == ET_BF16)
        ET_ACT_LAUNCH(bn_act_bwd_apply_kernel, uint16_t, act, grid, dim3(256), 0, (hipStream_t)stream, (const uint16_t*)dz, lddz,
                      (const uint16_t*)y, ldy, (uint16_t*)dy, lddy, P, CV, scale, shift, save_mean, save_invstd, nullptr, nullptr, nullptr, q);
    else if (dtype == ET_F16)
        ET_ACT_LAUNCH(bn_act_bwd_apply_kernel, et_f16, act, grid, dim3(256), 0, (hipStream_t)stream, (const et_f16*)dz, lddz,
                      (const et_f16*)y, ldy, (et_f16*)dy, lddy, P, CV, scale, shift, save_mean, save_invstd, nullptr, nullptr, nullptr, q);
    else return -2;
    ET_CHECK_LAUNCH();
    return 0;
}

extern "C" int et_act_bwd(const void* dz, int lddz, const void* y, int ldy, void* dy, int lddy, int dtype, int P, int C,
                          int act, et_stream_t stream) {
    if (!dz || !y || !dy) return -1;
    const int vec = dtype == ET_F32 ? 4 : 8;
    if (P <= 0 || C <= 0 || C % vec || lddz % vec || ldy % vec || lddy % vec) return -2;
    const int CV = C / vec;
    const dim3 grid(ew_blocks(P, CV));
    if (dtype == ET_F32)
        ET_ACT_LAUNCH(act_bwd_kernel, float, act, grid, dim3(256), 0, (hipStream_t)stream, (const float*)dz, lddz, (const float*)y, ldy,
                      (float*)dy, lddy, P, CV);
    else if (dtype == ET_BF16)
        ET_ACT_LAUNCH(act_bwd_kernel, uint16_t, act, grid, dim3(256), 0, (hipStream_t)stream, (const uint16_t*)dz, lddz,
                      (const uint16_t*)y, ldy, (uint16_t*)dy, lddy, P, CV);
    else if (dtype == ET_F16)
        ET_ACT_LAUNCH(act_bwd_kernel, et_f16, act, grid, dim3(256), 0, (hipStream_t)stream, (const et_f16*)dz, lddz,
                      (const et_f16*)y, ldy, (et_f16*)dy, lddy, P, CV);
    else return -2;
    ET_CHECK_LAUNCH();
    return 0;
}
