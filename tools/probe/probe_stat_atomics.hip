// Probe for "BatchNorm totals straight from the conv epilogue": `nblk` workgroups (one per CU and round, like the 256x256 conv
// tiles) spin for `spin` clock ticks (the tile's main loop), then their first `C` threads add two doubles each to the SAME
// 2*C totals (agent-scope atomics, what rows_reduce_finalize does today with far fewer adders).  mode 0: no atomics;
// mode 1: the atomics; mode 2: atomics + ticket, the last workgroup reads the totals back (the fused finalize).
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(512) void stat_atomics_kernel(double* __restrict__ totals, int* __restrict__ ticket, float* __restrict__ out,
                                                            int C, long long spin, int mode) {
    const long long t0 = __builtin_readcyclecounter();
    float acc = (float)threadIdx.x;
    while (__builtin_readcyclecounter() - t0 < spin) acc = acc * 1.0001f + 0.5f;
    if (mode == 0) { if (acc == 123.f) out[0] = acc; return; }
    const int c = threadIdx.x;
    if (c < C) {
        atomicAdd(totals + c, (double)acc * 1e-9);
        atomicAdd(totals + C + c, (double)acc * 2e-9);
    }
    if (mode == 1) return;
    __threadfence();
    __shared__ int last;
    __syncthreads();
    if (threadIdx.x == 0) last = atomicAdd(ticket, 1) == (int)gridDim.x - 1;
    __syncthreads();
    if (!last) return;
    if (threadIdx.x == 0) *ticket = 0;
    if (c < C) {
        const double S = __longlong_as_double((long long)atomicExch((unsigned long long*)(totals + c), 0ull));
        const double Q = __longlong_as_double((long long)atomicExch((unsigned long long*)(totals + C + c), 0ull));
        out[c] = (float)(S / (Q + 1.0));
    }
}

extern "C" int probe_stat_atomics(double* totals, int* ticket, float* out, int C, int nblk, long long spin, int mode, void* stream) {
    hipLaunchKernelGGL(stat_atomics_kernel, dim3(nblk), dim3(512), 0, (hipStream_t)stream, totals, ticket, out, C, spin, mode);
    return (int)hipGetLastError();
}
