// Hardware probe (tooling, not product): lane mapping of ds_read_b64_tr_b16 on gfx950.
#include <hip/hip_runtime.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void probe_kernel(const int* addr_bytes, unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 64) lds[i] = (unsigned short)i;   // value = element index
    __syncthreads();
    const int a = addr_bytes[threadIdx.x];
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)((char*)lds + a));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)v[j];
}
extern "C" int probe_tr(const int* addr_bytes, unsigned short* out, void* stream) {
    hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, addr_bytes, out);
    return (int)hipGetLastError();
}
