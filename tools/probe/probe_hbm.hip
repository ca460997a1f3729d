// HBM streaming probe: how fast can this part READ (and copy) a COLD buffer -- one far larger than the 256 MB Infinity Cache -- with
// the access forms the step's kernels use?  (the BatchNorm reduce pass reads dz + y at 3.7 TB/s in the step, the apply / forward passes
// move 5.7-5.8 TB/s of read + write)
//   mode 0: global_load_dwordx4, UNROLL loads in flight per thread, grid-stride over 16-byte vectors, sum -> one float per block
//   mode 1: the same with non-temporal loads
//   mode 2: LDS-DMA (global_load_lds_dwordx4) into a ring, nothing consumes it (pure fetch rate)
//   mode 3: copy (read + write), 16 bytes per lane
//   mode 4: two input streams (a[i], b[i]) like the reduce pass, plain loads
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef unsigned u4v __attribute__((ext_vector_type(4)));

template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void read_kernel(const u4v* __restrict__ a, long long n, float* out) {
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long step = (long long)gridDim.x * 256;
    unsigned acc = 0;
    for (; i + (UNROLL - 1) * step < n; i += UNROLL * step) {
        u4v v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(a + i + u * step) : a[i + u * step];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) out[blockIdx.x] = 1.f;
}

__global__ __launch_bounds__(256) void read2_kernel(const u4v* __restrict__ a, const u4v* __restrict__ b, long long n, float* out) {
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long step = (long long)gridDim.x * 256;
    unsigned acc = 0;
    for (; i + step < n; i += 2 * step) {
        const u4v v0 = a[i], w0 = b[i], v1 = a[i + step], w1 = b[i + step];
        acc += (v0.x ^ w0.y) + (v1.z ^ w1.w) + v0.w + w1.x;
    }
    if (acc == 0x12345678u) out[blockIdx.x] = 1.f;
}

__global__ __launch_bounds__(256) void lds_dma_kernel(const u4v* __restrict__ a, long long n, float* out) {
    __shared__ __attribute__((aligned(16))) u4v ring[8 * 256];          // 8 slots x 4 KB
    const int wave = threadIdx.x >> 6;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long step = (long long)gridDim.x * 256;
    int slot = 0;
    for (; i < n; i += step) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a + i),
                                         (__attribute__((address_space(3))) void*)(ring + slot * 256 + wave * 64), 16, 0, 0);
        slot = (slot + 1) & 7;
        if (slot == 0) __builtin_amdgcn_s_waitcnt(0x0F70 | 4);          // vmcnt(4): at most four pieces in flight behind this point
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    if (ring[threadIdx.x].x == 0x12345678u) out[blockIdx.x] = 1.f;
}

__global__ __launch_bounds__(256) void copy_kernel(const u4v* __restrict__ a, u4v* __restrict__ b, long long n) {
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long step = (long long)gridDim.x * 256;
    for (; i + step < n; i += 2 * step) {
        const u4v v0 = a[i], v1 = a[i + step];
        b[i] = v0; b[i + step] = v1;
    }
}

extern "C" int probe_hbm(int mode, int unroll, const void* a, const void* b, void* c, long long n_vec, int blocks, float* out, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    const u4v* pa = (const u4v*)a;
    if (mode == 0 || mode == 1) {
#define RK(U) do { if (mode == 0) hipLaunchKernelGGL((read_kernel<U, false>), dim3(blocks), dim3(256), 0, s, pa, n_vec, out); \
                   else hipLaunchKernelGGL((read_kernel<U, true>), dim3(blocks), dim3(256), 0, s, pa, n_vec, out); } while (0)
        if (unroll == 1) RK(1); else if (unroll == 2) RK(2); else if (unroll == 4) RK(4); else RK(8);
#undef RK
    } else if (mode == 2) hipLaunchKernelGGL(lds_dma_kernel, dim3(blocks), dim3(256), 0, s, pa, n_vec, out);
    else if (mode == 3) hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, s, pa, (u4v*)c, n_vec);
    else hipLaunchKernelGGL(read2_kernel, dim3(blocks), dim3(256), 0, s, pa, (const u4v*)b, n_vec, out);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
