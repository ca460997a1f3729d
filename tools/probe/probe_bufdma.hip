// Hardware probe (tooling, not product): buffer_load_dwordx4 ... lds (LDS-DMA through a buffer descriptor) on gfx950 --
// (1) do out-of-range lanes write ZEROS into LDS or leave it alone?  (2) is soffset part of the range check?
#include <hip/hip_runtime.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void probe_kernel(const unsigned* src, unsigned num_records, const int* voff, int soff, unsigned* out) {
    __shared__ __attribute__((aligned(16))) u32x4 lds[64];
    lds[threadIdx.x] = u32x4{0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu};
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, num_records, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, voff[threadIdx.x], soff, 0, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = lds[threadIdx.x][j];
}
extern "C" int probe_bufdma(const unsigned* src, unsigned num_records, const int* voff, int soff, unsigned* out, void* stream) {
    hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, src, num_records, voff, soff, out);
    return (int)hipGetLastError();
}
