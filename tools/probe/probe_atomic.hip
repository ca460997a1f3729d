// Probe: throughput of fp32 atomic adds at agent scope vs workgroup scope (executed in the XCD-local L2), in the
// access pattern of the wgrad epilogue: `nblk` workgroups of 256 threads, each adds a 128 x 128 fp32 tile
// (64 atomics per lane, a wave instruction covers 2 rows x 32 consecutive floats) into one of `ntiles`
// tiles; tile = f(blockIdx).  Also reports the XCC id every workgroup actually ran on.
#include <hip/hip_runtime.h>
#include <stdint.h>

template <int SCOPE>
__global__ __launch_bounds__(256) void atomic_tile_kernel(float* __restrict__ dw, int ntiles, int mode, int* __restrict__ xcc_out) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bid = blockIdx.x;
    unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11));   // HW_REG_XCC_ID, bits [3:0]
    if (tid == 0) xcc_out[bid] = (int)xcc;
    // mode 0: tile = bid % ntiles (splits of a tile spread over XCDs); mode 1: tile by XCC id (all adders of a
    // tile share an XCD: tile = xcc + 8 * ((bid / 8) % (ntiles / 8)))
    int tile = mode == 0 ? bid % ntiles : (int)xcc + 8 * ((bid >> 3) % (ntiles >> 3));
    float* base = dw + (size_t)tile * 128 * 128;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) {
                const int col = wn * 64 + tn * 32 + l31;
                float* p = base + row * 128 + col;
                if (SCOPE == 0) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
}

extern "C" int probe_atomic(float* dw, int ntiles, int nblk, int scope, int mode, int* xcc_out, void* stream) {
    if (scope == 0) hipLaunchKernelGGL(atomic_tile_kernel<0>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, dw, ntiles, mode, xcc_out);
    else hipLaunchKernelGGL(atomic_tile_kernel<1>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, dw, ntiles, mode, xcc_out);
    return (int)hipGetLastError();
}
