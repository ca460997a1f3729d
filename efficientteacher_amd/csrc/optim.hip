// Per-step state updates over FLAT fp32 arenas (all parameters / buffers of a model live in one
// contiguous allocation, see efficientteacher_amd/flat_state.py): one HBM-bound launch replaces the
// reference's python loop over 619 state tensors.
//   EMA  : utils/torch_utils.py:330-338 (ModelEMA), :366-375 (SemiSupModelEMA), :406-416 (CosineEMA)
//          v *= d ; v += (1-d) * m      (three separately rounded fp32 ops, as the reference)
//   SGD  : torch.optim.SGD(momentum, nesterov=True) as built at trainer/trainer.py:215-223, plus the
//          GradScaler-style 1/scale (trainer/trainer.py:399-400) and an optional bf16 shadow copy of
//          the updated weights for the MFMA kernels, all in the same pass.
// Compiled with -ffp-contract=off.
#include "et_device.h"
#include <math.h>
#include "../../include/et_hip.h"

// 16-bit shadow copy of an updated weight in the compute format (ET_BF16 / ET_F16)
__device__ __forceinline__ uint16_t lp_of(float v, int shadow_dtype) { return shadow_dtype == ET_F16 ? et_f2h(v) : et_f2bf(v); }
// Loss-scaler state in DEVICE memory (fp16 mode; torch.cuda.amp.GradScaler as the reference drives it, trainer/trainer.py:248,348,
// 399-400): {scale, 1 / scale, found_inf, growth_tracker}.  The optimizer kernels read it: found_inf != 0 -> the step is skipped
// (GradScaler.step), else the gradients are multiplied by 1 / scale (GradScaler.unscale_) -- with no host round trip, where
// torch's scaler.step() synchronises on found_inf.item().
#define ET_SCALER_SKIP(scaler) ((scaler) != nullptr && (scaler)[2] != 0.0f)
#define ET_SCALER_INV(scaler) ((scaler) != nullptr ? (scaler)[1] : 1.0f)

__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ v, const float* __restrict__ m, long long n,
                                                  float d, float omd) {
    const long long i4 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i4 + 3 < n) {
        float4 a = *(float4*)(v + i4);
        const float4 b = *(const float4*)(m + i4);
        a.x = a.x * d; a.x = a.x + omd * b.x;
        a.y = a.y * d; a.y = a.y + omd * b.y;
        a.z = a.z * d; a.z = a.z + omd * b.z;
        a.w = a.w * d; a.w = a.w + omd * b.w;
        *(float4*)(v + i4) = a;
    } else {
        for (long long i = i4; i < n; ++i) { float a = v[i] * d; v[i] = a + omd * m[i]; }
    }
}

extern "C" int et_ema_update(float* ema, const float* model, int64_t n, float d, float one_minus_d,
                             et_stream_t stream) {
    if (!ema || !model) return -1;
    if (n < 0 || (((uintptr_t)ema | (uintptr_t)model) & 15)) return -2;
    if (n == 0) return 0;
    hipLaunchKernelGGL(ema_kernel, dim3(et_cdiv(n, 1024)), dim3(256), 0, (hipStream_t)stream, ema, model,
                       (long long)n, d, one_minus_d);
    ET_CHECK_LAUNCH();
    return 0;
}

// Hyper-parameters in DEVICE memory (et_*_dev entry points): a captured HIP graph of the training step must not bake
// this step's lr / momentum / EMA decay into its kernel arguments -- warm-up, the lr schedule and ModelEMA's decay ramp
// change them from step to step.  The host writes a few floats before each replay; the arithmetic is unchanged.
__global__ __launch_bounds__(256) void ema_dev_kernel(float* __restrict__ v, const float* __restrict__ m, long long n,
                                                      const float* __restrict__ d2) {
    const float d = d2[0], omd = d2[1];
    const long long i4 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i4 + 3 < n) {
        float4 a = *(float4*)(v + i4);
        const float4 b = *(const float4*)(m + i4);
        a.x = a.x * d; a.x = a.x + omd * b.x;
        a.y = a.y * d; a.y = a.y + omd * b.y;
        a.z = a.z * d; a.z = a.z + omd * b.z;
        a.w = a.w * d; a.w = a.w + omd * b.w;
        *(float4*)(v + i4) = a;
    } else {
        for (long long i = i4; i < n; ++i) { float a = v[i] * d; v[i] = a + omd * m[i]; }
    }
}

extern "C" int et_ema_update_dev(float* ema, const float* model, int64_t n, const float* d_and_one_minus_d, et_stream_t stream) {
    if (!ema || !model || !d_and_one_minus_d) return -1;
    if (n < 0 || (((uintptr_t)ema | (uintptr_t)model) & 15)) return -2;
    if (n == 0) return 0;
    hipLaunchKernelGGL(ema_dev_kernel, dim3(et_cdiv(n, 1024)), dim3(256), 0, (hipStream_t)stream, ema, model, (long long)n,
                       d_and_one_minus_d);
    ET_CHECK_LAUNCH();
    return 0;
}

__global__ __launch_bounds__(256) void sgd_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf,
                                                      uint16_t* __restrict__ shadow, int shadow_dtype, long long n,
                                                      const float* __restrict__ hp, int first, const float* __restrict__ scaler) {
    if (ET_SCALER_SKIP(scaler)) return;
    const float lr = hp[0], mu = hp[1], wd = hp[2], inv_scale = hp[3] * ET_SCALER_INV(scaler);
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float gi = g[i] * inv_scale;
    const float pi = p[i];
    if (wd != 0.0f) gi = gi + wd * pi;
    float b = first ? gi : (buf[i] * mu + gi);
    buf[i] = b;
    gi = gi + mu * b;
    const float o = pi - lr * gi;
    p[i] = o;
    if (shadow) shadow[i] = lp_of(o, shadow_dtype);
}

extern "C" int et_sgd_nesterov_dev(float* p, const float* grad, float* momentum_buf, void* lp_shadow, int shadow_dtype, int64_t n,
                                   const float* hp /* device: lr, momentum, weight_decay, inv_scale */, int first_step,
                                   const float* scaler, et_stream_t stream) {
    if (!p || !grad || !momentum_buf || !hp) return -1;
    if (n < 0 || (lp_shadow && shadow_dtype != ET_BF16 && shadow_dtype != ET_F16)) return -2;
    if (n == 0) return 0;
    hipLaunchKernelGGL(sgd_dev_kernel, dim3(et_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, p, grad, momentum_buf,
                       (uint16_t*)lp_shadow, shadow_dtype, (long long)n, hp, first_step, scaler);
    ET_CHECK_LAUNCH();
    return 0;
}

__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                  float* __restrict__ buf, uint16_t* __restrict__ shadow, int shadow_dtype,
                                                  long long n, float lr, float mu, float wd, int first,
                                                  float inv_scale_host, const float* __restrict__ scaler) {
    if (ET_SCALER_SKIP(scaler)) return;
    const float inv_scale = inv_scale_host * ET_SCALER_INV(scaler);
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float gi = g[i] * inv_scale;
    const float pi = p[i];
    if (wd != 0.0f) gi = gi + wd * pi;
    float b = first ? gi : (buf[i] * mu + gi);
    buf[i] = b;
    gi = gi + mu * b;
    const float o = pi - lr * gi;
    p[i] = o;
    if (shadow) shadow[i] = lp_of(o, shadow_dtype);
}

extern "C" int et_sgd_nesterov(float* p, const float* grad, float* momentum_buf, void* lp_shadow, int shadow_dtype, int64_t n,
                               float lr, float momentum, float weight_decay, int first_step, float inv_scale,
                               const float* scaler, et_stream_t stream) {
    if (!p || !grad || !momentum_buf) return -1;
    if (n < 0 || (lp_shadow && shadow_dtype != ET_BF16 && shadow_dtype != ET_F16)) return -2;
    if (n == 0) return 0;
    hipLaunchKernelGGL(sgd_kernel, dim3(et_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, p, grad, momentum_buf,
                       (uint16_t*)lp_shadow, shadow_dtype, (long long)n, lr, momentum, weight_decay, first_step, inv_scale, scaler);
    ET_CHECK_LAUNCH();
    return 0;
}

// AdamW (torch.optim.AdamW as built at trainer/trainer.py:212 when cfg.adam: betas = (hyp.momentum, 0.999), eps 1e-8, decoupled
// weight decay), same operation order as torch's single-tensor implementation:
//   p *= 1 - lr * wd ; m = lerp(m, g, 1 - b1) ; v = v * b2 + (1 - b2) g^2 ; p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
// bc1 = 1 - b1^t, bc2 = 1 - b2^t are computed on the host in double and passed as floats (step_size, 1/sqrt(bc2)).
template <bool DEV>
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, uint16_t* __restrict__ shadow, int shadow_dtype, long long n,
                                                    float lr_wd, float b1, float b2, float step_size, float inv_sqrt_bc2, float eps,
                                                    float inv_scale_host, const float* __restrict__ scaler, float lr,
                                                    const double* __restrict__ tick) {
    if (ET_SCALER_SKIP(scaler)) return;
    if constexpr (DEV) {                                   // bias corrections of the device-resident step count (et_adamw_tick)
        step_size = (float)((double)lr / tick[1]);
        inv_sqrt_bc2 = (float)(1.0 / sqrt(tick[2]));
    }
    const float inv_scale = inv_scale_host * ET_SCALER_INV(scaler);
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i] * inv_scale;
    float pi = p[i];
    pi = pi * (1.0f - lr_wd);
    float mi = m[i];
    mi = mi + (gi - mi) * (1.0f - b1);
    float vi = v[i] * b2;
    vi = vi + (1.0f - b2) * gi * gi;
    const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
    pi = pi - step_size * (mi / denom);
    p[i] = pi; m[i] = mi; v[i] = vi;
    if (shadow) shadow[i] = lp_of(pi, shadow_dtype);
}

extern "C" int et_adamw(float* p, const float* grad, float* exp_avg, float* exp_avg_sq, void* lp_shadow, int shadow_dtype, int64_t n,
                        float lr, float beta1, float beta2, float eps, float weight_decay, int step, float inv_scale,
                        const float* scaler, et_stream_t stream) {
    if (!p || !grad || !exp_avg || !exp_avg_sq) return -1;
    if (n < 0 || step < 1 || (lp_shadow && shadow_dtype != ET_BF16 && shadow_dtype != ET_F16)) return -2;
    if (n == 0) return 0;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(adamw_kernel<false>, dim3(et_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, p, grad, exp_avg, exp_avg_sq,
                       (uint16_t*)lp_shadow, shadow_dtype, (long long)n, lr * weight_decay, beta1, beta2, (float)((double)lr / bc1),
                       (float)(1.0 / sqrt(bc2)), eps, inv_scale, scaler, lr, (const double*)nullptr);
    ET_CHECK_LAUNCH();
    return 0;
}

// AdamW's step count under the loss scaler: torch's GradScaler.step does not call optimizer.step() on an overflow, so a
// skipped step must not advance the bias corrections -- and whether a step is skipped is only known on the device (found_inf).  The
// count lives in device memory: tick = {t, 1 - beta1^t, 1 - beta2^t} (doubles); one single-thread launch per optimizer step advances it
// unless found_inf is set, and the update kernels of that step form lr / bc1 and 1 / sqrt(bc2) from it.
__global__ void adamw_tick_kernel(double* __restrict__ tick, double b1, double b2, const float* __restrict__ scaler) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (ET_SCALER_SKIP(scaler)) return;
    const double t = tick[0] + 1.0;
    tick[0] = t; tick[1] = 1.0 - pow(b1, t); tick[2] = 1.0 - pow(b2, t);
}
extern "C" int et_adamw_tick(double* tick, float beta1, float beta2, const float* scaler, et_stream_t stream) {
    if (!tick) return -1;
    if (!(beta1 >= 0.0f && beta1 < 1.0f) || !(beta2 >= 0.0f && beta2 < 1.0f)) return -2;
    hipLaunchKernelGGL(adamw_tick_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, tick, (double)beta1, (double)beta2, scaler);
    ET_CHECK_LAUNCH();
    return 0;
}
extern "C" int et_adamw_dev(float* p, const float* grad, float* exp_avg, float* exp_avg_sq, void* lp_shadow, int shadow_dtype, int64_t n,
                            float lr, float beta1, float beta2, float eps, float weight_decay, const double* tick, float inv_scale,
                            const float* scaler, et_stream_t stream) {
    if (!p || !grad || !exp_avg || !exp_avg_sq || !tick) return -1;
    if (n < 0 || (lp_shadow && shadow_dtype != ET_BF16 && shadow_dtype != ET_F16)) return -2;
    if (n == 0) return 0;
    hipLaunchKernelGGL(adamw_kernel<true>, dim3(et_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, p, grad, exp_avg, exp_avg_sq,
                       (uint16_t*)lp_shadow, shadow_dtype, (long long)n, lr * weight_decay, beta1, beta2, 0.0f, 0.0f, eps, inv_scale,
                       scaler, lr, tick);
    ET_CHECK_LAUNCH();
    return 0;
}

// fp32 -> 16-bit compute format cast of a flat arena (initial shadow copy of the weights)
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ s, uint16_t* __restrict__ d, long long n, int dtype) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) d[i] = lp_of(s[i], dtype);
}
// eight elements per thread (two 16-byte loads, one 16-byte store): the teacher's bf16 weight shadow is refreshed from the EMA
// master every step, and the element-per-thread kernel above took 175 us for its 276 MB
template <typename T>
__global__ __launch_bounds__(256) void cast_bf16_vec8_kernel(const float* __restrict__ s, T* __restrict__ d, long long n8) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const float4 a = *(const float4*)(s + i * 8), b = *(const float4*)(s + i * 8 + 4);
    *(uint4*)(d + i * 8) = make_uint4(et_lp<T>::pack(a.x, a.y), et_lp<T>::pack(a.z, a.w), et_lp<T>::pack(b.x, b.y), et_lp<T>::pack(b.z, b.w));
}
extern "C" int et_cast_f32_to_lp(const float* src, void* dst, int dtype, int64_t n, et_stream_t stream) {
    if (!src || !dst) return -1;
    if (dtype != ET_BF16 && dtype != ET_F16) return -2;
    if (n <= 0) return n == 0 ? 0 : -2;
    long long done = 0;
    if (((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0 && n >= 8) {
        const long long n8 = n / 8;
        if (dtype == ET_BF16) hipLaunchKernelGGL((cast_bf16_vec8_kernel<uint16_t>), dim3(et_cdiv(n8, 256)), dim3(256), 0, (hipStream_t)stream, src, (uint16_t*)dst, n8);
        else hipLaunchKernelGGL((cast_bf16_vec8_kernel<et_f16>), dim3(et_cdiv(n8, 256)), dim3(256), 0, (hipStream_t)stream, src, (et_f16*)dst, n8);
        done = n8 * 8;
    }
    if (done < n)
        hipLaunchKernelGGL(cast_bf16_kernel, dim3(et_cdiv(n - done, 256)), dim3(256), 0, (hipStream_t)stream, src + done,
                           (uint16_t*)dst + done, (long long)(n - done), dtype);
    ET_CHECK_LAUNCH();
    return 0;
}

// ---- loss scaler (fp16 mode) ------------------------------------------------------------------------------------
// found_inf over the scaled fp32 gradient arena: torch's _amp_foreach_non_finite_check_and_unscale_ minus the unscale (the optimizer
// kernels multiply by 1 / scale as they read the gradient).  One 16-byte load per thread and trip; a wave that saw a non-finite
// value stores 1.0f -- a plain store of the same value from many waves, no atomic needed.
__global__ __launch_bounds__(256) void scaler_check_kernel(const float* __restrict__ g, long long n, float* __restrict__ scaler) {
    const long long stride = (long long)gridDim.x * 256 * 4;
    bool bad = false;
    for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 3 < n) {
            const float4 v = *(const float4*)(g + i);
            bad |= !(fabsf(v.x) <= 3.402823466e38f) | !(fabsf(v.y) <= 3.402823466e38f) | !(fabsf(v.z) <= 3.402823466e38f) | !(fabsf(v.w) <= 3.402823466e38f);
        } else {
            for (long long j = i; j < n; ++j) bad |= !(fabsf(g[j]) <= 3.402823466e38f);
        }
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) scaler[2] = 1.0f;
}
extern "C" int et_scaler_check(const float* grads, int64_t n, float* scaler, et_stream_t stream) {
    if (!grads || !scaler) return -1;
    if (n < 0 || (((uintptr_t)grads) & 15)) return -2;
    if (n == 0) return 0;
    long long blocks = et_cdiv(n, 256 * 4 * 8);            // ~8 trips per thread
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(scaler_check_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, grads, (long long)n, scaler);
    ET_CHECK_LAUNCH();
    return 0;
}
// GradScaler.update (torch's _amp_update_scale_): after a skipped step the scale shrinks, after `growth_interval` clean steps in a row
// it grows; found_inf is cleared for the next step.
__global__ void scaler_update_kernel(float* __restrict__ scaler, float growth, float backoff, int interval) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float scale = scaler[0], tracker = scaler[3];
    if (scaler[2] != 0.0f) { scale = scale * backoff; tracker = 0.0f; }
    else {
        tracker = tracker + 1.0f;
        if (tracker >= (float)interval) {
            const float grown = scale * growth;
            if (fabsf(grown) <= 3.402823466e38f) scale = grown;         // torch: never grow into inf
            tracker = 0.0f;
        }
    }
    scaler[0] = scale; scaler[1] = 1.0f / scale; scaler[2] = 0.0f; scaler[3] = tracker;
}
extern "C" int et_scaler_update(float* scaler, float growth_factor, float backoff_factor, int growth_interval, et_stream_t stream) {
    if (!scaler) return -1;
    if (!(growth_factor >= 1.0f) || !(backoff_factor > 0.0f && backoff_factor <= 1.0f) || growth_interval < 1) return -2;
    hipLaunchKernelGGL(scaler_update_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, scaler, growth_factor, backoff_factor, growth_interval);
    ET_CHECK_LAUNCH();
    return 0;
}
