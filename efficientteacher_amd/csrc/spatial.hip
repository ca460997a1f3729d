// Spatial data-movement kernels of the YOLOv5 graph on NHWC tensors (all HBM-bound, 16 B per lane):
//   * input packing: (B,3,H,W) fp32 NCHW image -> (B,H,W,8) NHWC, channels zero-padded to one
//     16-byte MFMA K-vector (the stem conv, models/backbone/yolov5_backbone.py:56, then sees Cin=8);
//   * SPPF max-pool 5x5 s1 p2 (models/backbone/common.py:702-708), forward with argmax, backward as
//     a gather so that no atomics are needed; outputs land directly in channel slices of the SPPF
//     concat buffer;
//   * nearest 2x upsample (models/neck/yolov5_neck.py:60,64) written straight into the channel slice
//     of the following concat buffer, and its backward (sum of the 4 children).
#include "et_device.h"
#include "../../include/et_hip.h"

template <typename T>
__global__ __launch_bounds__(256) void pack_input_kernel(const float* __restrict__ x, T* __restrict__ y, int C, int HW, long long total) {
    const unsigned iu = blockIdx.x * 256u + threadIdx.x;                // pixel index over B*H*W (host: < 2^31; 32-bit divisions)
    if (iu >= total) return;
    const long long i = iu, b = iu / (unsigned)HW, hw = iu - (unsigned)b * (unsigned)HW;
    T v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = et_elem<T>::st(c < C ? x[(b * C + c) * HW + hw] : 0.f);
    T* d = y + i * 8;
#pragma unroll
    for (int c = 0; c < 8; ++c) d[c] = v[c];
}

// uint8 NCHW batch as the data loaders deliver it (utils/datasets.py:1164, datasets_ssod.py:591) -> NHWC8 of the compute
// dtype with the reference's `.float() / norm_scale` (trainer.py:411, ssod_trainer.py:694-696) folded in: 3 bytes read and
// 16 written per pixel instead of a float conversion pass, a division pass and the fp32 pack (4+4+4+4+12+16 bytes).
// v / scale is an IEEE division: bit-identical to torch's `x.float() / 255.0` on the CPU (the oracle); torch's GPU kernel
// multiplies by the rounded reciprocal, which differs by at most 1 ulp.
template <typename T>
__global__ __launch_bounds__(256) void pack_input_u8_kernel(const uint8_t* __restrict__ x, T* __restrict__ y, int C, int HW,
                                                            long long total, float scale) {
    const unsigned iu = blockIdx.x * 256u + threadIdx.x;                // pixel index over B*H*W (host: < 2^31; 32-bit divisions)
    if (iu >= total) return;
    const long long i = iu, b = iu / (unsigned)HW, hw = iu - (unsigned)b * (unsigned)HW;
    T v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = et_elem<T>::st(c < C ? (float)x[(b * C + c) * HW + hw] / scale : 0.f);
    T* d = y + i * 8;
#pragma unroll
    for (int c = 0; c < 8; ++c) d[c] = v[c];
}

// bf16 output, FOUR consecutive pixels per thread: one 16-byte (fp32 image) or 4-byte (uint8 image) load per channel and 64
// contiguous bytes stored, instead of a 4- / 1-byte load per channel and pixel (the fp32 pack ran at 3.7 TB/s, the uint8 one issues
// 64-byte wave loads).  Same arithmetic: fp32 value rounded to bf16; uint8 value / norm_scale (IEEE division), then rounded.
template <typename IN, typename T = uint16_t>       // T: the 16-bit output format (uint16_t = bf16, et_f16)
__global__ __launch_bounds__(256) void pack_input4_bf16_kernel(const IN* __restrict__ x, T* __restrict__ y, int C, int HW4,
                                                               long long total4, float scale) {
    const unsigned qu = blockIdx.x * 256u + threadIdx.x;               // quad index over B*HW/4 (host: < 2^31)
    if (qu >= total4) return;
    const unsigned b = qu / (unsigned)HW4, h4 = qu - b * (unsigned)HW4;
    float v[4][4];                                                      // [channel][pixel]; channels >= C stay zero (C <= 4 here)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[c][j] = 0.f;
        if (c < C) {
            const long long off = ((long long)b * C + c) * HW4 * 4 + (long long)h4 * 4;
            if constexpr (sizeof(IN) == 4) {
                const float4 t = *(const float4*)((const float*)x + off);
                v[c][0] = t.x; v[c][1] = t.y; v[c][2] = t.z; v[c][3] = t.w;
            } else {
                const unsigned w = *(const unsigned*)((const uint8_t*)x + off);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[c][j] = (float)((w >> (8 * j)) & 0xffu) / scale;
            }
        }
    }
    T* const d = y + ((long long)b * HW4 * 4 + (long long)h4 * 4) * 8;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        *(uint4*)(d + j * 8) = make_uint4(et_lp<T>::pack(v[0][j], v[1][j]), et_lp<T>::pack(v[2][j], v[3][j]), 0u, 0u);
}

template <typename T> struct PV;   // 16-byte vector <-> floats
template <> struct PV<float> {
    static constexpr int N = 4;
    __device__ static __forceinline__ void load(const float* p, float (&v)[4]) { const float4 t = *(const float4*)p; v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
    __device__ static __forceinline__ void store(float* p, const float (&v)[4]) { *(float4*)p = make_float4(v[0], v[1], v[2], v[3]); }
};
template <typename T> struct PVlp {       // the 16-bit storage formats
    static constexpr int N = 8;
    __device__ static __forceinline__ void load(const T* p, float (&v)[8]) {
        const uint4 t = *(const uint4*)p;
        const unsigned w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[2 * i] = et_lp<T>::lo(w[i]); v[2 * i + 1] = et_lp<T>::hi(w[i]); }
    }
    __device__ static __forceinline__ void store(T* p, const float (&v)[8]) {
        *(uint4*)p = make_uint4(et_lp<T>::pack(v[0], v[1]), et_lp<T>::pack(v[2], v[3]), et_lp<T>::pack(v[4], v[5]), et_lp<T>::pack(v[6], v[7]));
    }
};
template <> struct PV<uint16_t> : PVlp<uint16_t> {};
template <> struct PV<et_f16> : PVlp<et_f16> {};

// y = maxpool5x5(x); idx = window position (ky*5+kx) of the FIRST maximum in scan order
// (torch max_pool2d tie rule: strict '>' while scanning rows then columns).
template <typename T>
__global__ __launch_bounds__(256) void maxpool5_fwd_kernel(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy,
                                                           unsigned char* __restrict__ idx, int H, int W, int CV, long long total) {
    constexpr int N = PV<T>::N;
    const unsigned iu = blockIdx.x * 256u + threadIdx.x;   // over B*H*W*CV
    if (iu >= total) return;                                  // host: total < 2^31 (32-bit divisions: the 64-bit ones were ~600 instructions per thread)
    const unsigned pu = iu / (unsigned)CV, tu = pu / (unsigned)W;
    const int cv = (int)(iu - pu * (unsigned)CV);
    const long long p = pu;
    const int ox = (int)(pu - tu * (unsigned)W);
    const long long b = tu / (unsigned)H;
    const int oy = (int)(tu - (unsigned)b * (unsigned)H);
    float best[N];
    unsigned char bi[N];
#pragma unroll
    for (int k = 0; k < N; ++k) { best[k] = -INFINITY; bi[k] = 0; }
    for (int ky = 0; ky < 5; ++ky) {
        const int iy = oy + ky - 2;
        if ((unsigned)iy >= (unsigned)H) continue;
        for (int kx = 0; kx < 5; ++kx) {
            const int ix = ox + kx - 2;
            if ((unsigned)ix >= (unsigned)W) continue;
            float v[N];
            PV<T>::load(x + ((b * H + iy) * W + ix) * ldx + cv * N, v);
#pragma unroll
            for (int k = 0; k < N; ++k)
                if (v[k] > best[k]) { best[k] = v[k]; bi[k] = (unsigned char)(ky * 5 + kx); }
        }
    }
    PV<T>::store(y + p * ldy + cv * N, best);
    unsigned char* ip = idx + (p * CV + cv) * N;
#pragma unroll
    for (int k = 0; k < N; ++k) ip[k] = bi[k];
}

// dx = base + sum over the outputs whose argmax is this input position of dy
template <typename T>
__global__ __launch_bounds__(256) void maxpool5_bwd_kernel(const T* __restrict__ dy, int lddy, const unsigned char* __restrict__ idx,
                                                           const T* __restrict__ base, int ldb, T* __restrict__ dx, int lddx,
                                                           int H, int W, int CV, long long total) {
    constexpr int N = PV<T>::N;
    const unsigned iu = blockIdx.x * 256u + threadIdx.x;
    if (iu >= total) return;                                  // host: total < 2^31 (32-bit divisions: the 64-bit ones were ~600 instructions per thread)
    const unsigned pu = iu / (unsigned)CV, tu = pu / (unsigned)W;
    const int cv = (int)(iu - pu * (unsigned)CV);
    const long long p = pu;
    const int ix = (int)(pu - tu * (unsigned)W);
    const long long b = tu / (unsigned)H;
    const int iy = (int)(tu - (unsigned)b * (unsigned)H);
    float acc[N];
    if (base) PV<T>::load(base + p * ldb + cv * N, acc);
    else {
#pragma unroll
        for (int k = 0; k < N; ++k) acc[k] = 0.f;
    }
    for (int ky = 0; ky < 5; ++ky) {
        const int oy = iy - ky + 2;
        if ((unsigned)oy >= (unsigned)H) continue;
        for (int kx = 0; kx < 5; ++kx) {
            const int ox = ix - kx + 2;
            if ((unsigned)ox >= (unsigned)W) continue;
            const long long q = (b * H + oy) * W + ox;
            float g[N];
            PV<T>::load(dy + q * lddy + cv * N, g);
            const unsigned char* ip = idx + (q * CV + cv) * N;
            const unsigned char code = (unsigned char)(ky * 5 + kx);
#pragma unroll
            for (int k = 0; k < N; ++k)
                if (ip[k] == code) acc[k] += g[k];
        }
    }
    PV<T>::store(dx + p * lddx + cv * N, acc);
}

// y[b, 2h+dy, 2w+dx, :] = x[b, h, w, :]
template <typename T>
__global__ __launch_bounds__(256) void upsample2x_fwd_kernel(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy, int H, int W,
                                                             int CV, long long total) {
    const unsigned iu = blockIdx.x * 256u + threadIdx.x;             // over B*2H*2W*CV (host: < 2^31)
    if (iu >= total) return;
    const unsigned pu = iu / (unsigned)CV, tu = pu / (unsigned)(2 * W);
    const int cv = (int)(iu - pu * (unsigned)CV);
    const long long p = pu;
    const int ox = (int)(pu - tu * (unsigned)(2 * W));
    const long long b = tu / (unsigned)(2 * H);
    const int oy = (int)(tu - (unsigned)b * (unsigned)(2 * H));
    const uint4 v = *(const uint4*)((const char*)(x + ((b * H + (oy >> 1)) * W + (ox >> 1)) * ldx) + (size_t)cv * 16);
    *(uint4*)((char*)(y + p * ldy) + (size_t)cv * 16) = v;
}

// dx[b,h,w,:] (+)= sum of dy over the 4 children.  accumulate: dx already holds the gradient of another consumer of the upsampled
// tensor (autograd would add the two branches in a separate bf16 pass)
template <typename T>
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const T* __restrict__ dy, int lddy, T* __restrict__ dx, int lddx, int H, int W,
                                                             int CV, long long total, int accumulate) {
    constexpr int N = PV<T>::N;
    const unsigned iu = blockIdx.x * 256u + threadIdx.x;   // over B*H*W*CV
    if (iu >= total) return;                                  // host: total < 2^31 (32-bit divisions: the 64-bit ones were ~600 instructions per thread)
    const unsigned pu = iu / (unsigned)CV, tu = pu / (unsigned)W;
    const int cv = (int)(iu - pu * (unsigned)CV);
    const long long p = pu;
    const int w = (int)(pu - tu * (unsigned)W);
    const long long b = tu / (unsigned)H;
    const int h = (int)(tu - (unsigned)b * (unsigned)H);
    float acc[N];
    if (accumulate) PV<T>::load(dx + p * lddx + cv * N, acc);
    else {
#pragma unroll
        for (int k = 0; k < N; ++k) acc[k] = 0.f;
    }
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const long long q = (b * 2 * H + (2 * h + (d >> 1))) * 2 * W + (2 * w + (d & 1));
        float g[N];
        PV<T>::load(dy + q * lddy + cv * N, g);
#pragma unroll
        for (int k = 0; k < N; ++k) acc[k] += g[k];
    }
    PV<T>::store(dx + p * lddx + cv * N, acc);
}

extern "C" int et_pack_input(const float* x_nchw, void* y_nhwc8, int dtype, int B, int C, int H, int W, et_stream_t stream) {
    if (!x_nchw || !y_nhwc8) return -1;
    if (B <= 0 || C <= 0 || C > 8 || H <= 0 || W <= 0) return -2;
    const long long total = (long long)B * H * W;
    if (total >= (1ll << 31)) return -2;               // the kernels index with 32-bit arithmetic
    const dim3 grid(et_cdiv(total, 256));
    if (dtype == ET_F32) hipLaunchKernelGGL((pack_input_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, x_nchw, (float*)y_nhwc8, C, H * W, total);
    else if (dtype == ET_BF16) {
        if (C <= 4 && (H * W) % 4 == 0 && ((((uintptr_t)x_nchw) | ((uintptr_t)y_nhwc8)) & 15) == 0)
            hipLaunchKernelGGL((pack_input4_bf16_kernel<float>), dim3(et_cdiv(total / 4, 256)), dim3(256), 0, (hipStream_t)stream, x_nchw,
                               (uint16_t*)y_nhwc8, C, H * W / 4, total / 4, 1.0f);
        else
            hipLaunchKernelGGL((pack_input_kernel<uint16_t>), grid, dim3(256), 0, (hipStream_t)stream, x_nchw, (uint16_t*)y_nhwc8, C, H * W, total);
    } else if (dtype == ET_F16) {
        if (C <= 4 && (H * W) % 4 == 0 && ((((uintptr_t)x_nchw) | ((uintptr_t)y_nhwc8)) & 15) == 0)
            hipLaunchKernelGGL((pack_input4_bf16_kernel<float, et_f16>), dim3(et_cdiv(total / 4, 256)), dim3(256), 0, (hipStream_t)stream, x_nchw,
                               (et_f16*)y_nhwc8, C, H * W / 4, total / 4, 1.0f);
        else
            hipLaunchKernelGGL((pack_input_kernel<et_f16>), grid, dim3(256), 0, (hipStream_t)stream, x_nchw, (et_f16*)y_nhwc8, C, H * W, total);
    } else return -2;
    ET_CHECK_LAUNCH();
    return 0;
}

extern "C" int et_pack_input_u8(const uint8_t* x_nchw, void* y_nhwc8, int dtype, int B, int C, int H, int W, float norm_scale,
                                et_stream_t stream) {
    if (!x_nchw || !y_nhwc8) return -1;
    if (B <= 0 || C <= 0 || C > 8 || H <= 0 || W <= 0 || !(norm_scale > 0.f)) return -2;
    const long long total = (long long)B * H * W;
    if (total >= (1ll << 31)) return -2;               // the kernels index with 32-bit arithmetic
    const dim3 grid(et_cdiv(total, 256));
    if (dtype == ET_F32) hipLaunchKernelGGL((pack_input_u8_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, x_nchw, (float*)y_nhwc8, C, H * W, total, norm_scale);
    else if (dtype == ET_BF16) {
        if (C <= 4 && (H * W) % 4 == 0 && (((uintptr_t)x_nchw) & 3) == 0 && (((uintptr_t)y_nhwc8) & 15) == 0)
            hipLaunchKernelGGL((pack_input4_bf16_kernel<uint8_t>), dim3(et_cdiv(total / 4, 256)), dim3(256), 0, (hipStream_t)stream, x_nchw,
                               (uint16_t*)y_nhwc8, C, H * W / 4, total / 4, norm_scale);
        else
            hipLaunchKernelGGL((pack_input_u8_kernel<uint16_t>), grid, dim3(256), 0, (hipStream_t)stream, x_nchw, (uint16_t*)y_nhwc8, C, H * W, total, norm_scale);
    } else if (dtype == ET_F16) {
        if (C <= 4 && (H * W) % 4 == 0 && (((uintptr_t)x_nchw) & 3) == 0 && (((uintptr_t)y_nhwc8) & 15) == 0)
            hipLaunchKernelGGL((pack_input4_bf16_kernel<uint8_t, et_f16>), dim3(et_cdiv(total / 4, 256)), dim3(256), 0, (hipStream_t)stream, x_nchw,
                               (et_f16*)y_nhwc8, C, H * W / 4, total / 4, norm_scale);
        else
            hipLaunchKernelGGL((pack_input_u8_kernel<et_f16>), grid, dim3(256), 0, (hipStream_t)stream, x_nchw, (et_f16*)y_nhwc8, C, H * W, total, norm_scale);
    } else return -2;
    ET_CHECK_LAUNCH();
    return 0;
}

extern "C" int et_maxpool5_fwd(const void* x, int ldx, void* y, int ldy, uint8_t* argmax, int dtype, int B, int H, int W, int C,
                               et_stream_t stream) {
    if (!x || !y || !argmax) return -1;
    const int vec = dtype == ET_F32 ? 4 : 8;
    if (B <= 0 || C % vec || ldx % vec || ldy % vec) return -2;
    const int CV = C / vec;
    const long long total = (long long)B * H * W * CV;
    if (total >= (1ll << 31)) return -2;               // the kernels index with 32-bit arithmetic
    if (dtype == ET_F32) { const dim3 g(et_cdiv(total, 256)); hipLaunchKernelGGL((maxpool5_fwd_kernel<float>), g, dim3(256), 0, (hipStream_t)stream, (const float*)x, ldx, (float*)y, ldy, argmax, H, W, CV, total); }
    else if (dtype == ET_BF16) { const dim3 g(et_cdiv(total, 256)); hipLaunchKernelGGL((maxpool5_fwd_kernel<uint16_t>), g, dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, ldx, (uint16_t*)y, ldy, argmax, H, W, CV, total); }
    else if (dtype == ET_F16) { const dim3 g(et_cdiv(total, 256)); hipLaunchKernelGGL((maxpool5_fwd_kernel<et_f16>), g, dim3(256), 0, (hipStream_t)stream, (const et_f16*)x, ldx, (et_f16*)y, ldy, argmax, H, W, CV, total); }
    else return -2;
    ET_CHECK_LAUNCH();
    return 0;
}

extern "C" int et_maxpool5_bwd(const void* dy, int lddy, const uint8_t* argmax, const void* base, int ldb, void* dx, int lddx,
                               int dtype, int B, int H, int W, int C, et_stream_t stream) {
    if (!dy || !dx || !argmax) return -1;
    const int vec = dtype == ET_F32 ? 4 : 8;
    if (B <= 0 || C % vec || lddy % vec || lddx % vec || (base && ldb % vec)) return -2;
    const int CV = C / vec;
    const long long total = (long long)B * H * W * CV;
    if (total >= (1ll << 31)) return -2;               // the kernels index with 32-bit arithmetic
    if (dtype == ET_F32) { const dim3 g(et_cdiv(total, 256)); hipLaunchKernelGGL((maxpool5_bwd_kernel<float>), g, dim3(256), 0, (hipStream_t)stream, (const float*)dy, lddy, argmax, (const float*)base, ldb, (float*)dx, lddx, H, W, CV, total); }
    else if (dtype == ET_BF16) { const dim3 g(et_cdiv(total, 256)); hipLaunchKernelGGL((maxpool5_bwd_kernel<uint16_t>), g, dim3(256), 0, (hipStream_t)stream, (const uint16_t*)dy, lddy, argmax, (const uint16_t*)base, ldb, (uint16_t*)dx, lddx, H, W, CV, total); }
    else if (dtype == ET_F16) { const dim3 g(et_cdiv(total, 256)); hipLaunchKernelGGL((maxpool5_bwd_kernel<et_f16>), g, dim3(256), 0, (hipStream_t)stream, (const et_f16*)dy, lddy, argmax, (const et_f16*)base, ldb, (et_f16*)dx, lddx, H, W, CV, total); }
    else return -2;
    ET_CHECK_LAUNCH();
    return 0;
}

extern "C" int et_upsample2x_fwd(const void* x, int ldx, void* y, int ldy, int dtype, int B, int H, int W, int C, et_stream_t stream) {
    if (!x || !y) return -1;
    const int vec = dtype == ET_F32 ? 4 : 8;
    if (B <= 0 || C % vec || ldx % vec || ldy % vec) return -2;
    const int CV = C / vec;
    const long long total = (long long)B * 4 * H * W * CV;
    if (total >= (1ll << 31)) return -2;               // the kernels index with 32-bit arithmetic
    if (dtype == ET_F32) { const dim3 g(et_cdiv(total, 256)); hipLaunchKernelGGL((upsample2x_fwd_kernel<float>), g, dim3(256), 0, (hipStream_t)stream, (const float*)x, ldx, (float*)y, ldy, H, W, CV, total); }
    else if (dtype == ET_BF16) { const dim3 g(et_cdiv(total, 256)); hipLaunchKernelGGL((upsample2x_fwd_kernel<uint16_t>), g, dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, ldx, (uint16_t*)y, ldy, H, W, CV, total); }
    else if (dtype == ET_F16) { const dim3 g(et_cdiv(total, 256)); hipLaunchKernelGGL((upsample2x_fwd_kernel<et_f16>), g, dim3(256), 0, (hipStream_t)stream, (const et_f16*)x, ldx, (et_f16*)y, ldy, H, W, CV, total); }
    else return -2;
    ET_CHECK_LAUNCH();
    return 0;
}

extern "C" int et_upsample2x_bwd(const void* dy, int lddy, void* dx, int lddx, int dtype, int B, int H, int W, int C, int accumulate,
                                 et_stream_t stream) {
    if (!dy || !dx) return -1;
    const int vec = dtype == ET_F32 ? 4 : 8;
    if (B <= 0 || C % vec || lddy % vec || lddx % vec) return -2;
    const int CV = C / vec;
    const long long total = (long long)B * H * W * CV;
    if (total >= (1ll << 31)) return -2;               // the kernels index with 32-bit arithmetic
    if (dtype == ET_F32) { const dim3 g(et_cdiv(total, 256)); hipLaunchKernelGGL((upsample2x_bwd_kernel<float>), g, dim3(256), 0, (hipStream_t)stream, (const float*)dy, lddy, (float*)dx, lddx, H, W, CV, total, accumulate); }
    else if (dtype == ET_BF16) { const dim3 g(et_cdiv(total, 256)); hipLaunchKernelGGL((upsample2x_bwd_kernel<uint16_t>), g, dim3(256), 0, (hipStream_t)stream, (const uint16_t*)dy, lddy, (uint16_t*)dx, lddx, H, W, CV, total, accumulate); }
    else if (dtype == ET_F16) { const dim3 g(et_cdiv(total, 256)); hipLaunchKernelGGL((upsample2x_bwd_kernel<et_f16>), g, dim3(256), 0, (hipStream_t)stream, (const et_f16*)dy, lddy, (et_f16*)dx, lddx, H, W, CV, total, accumulate); }
    else return -2;
    ET_CHECK_LAUNCH();
    return 0;
}
