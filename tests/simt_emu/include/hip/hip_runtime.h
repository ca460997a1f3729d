// SIMT emulator -- TEST INFRASTRUCTURE ONLY (never shipped, never loaded by the product).
//
// A stand-in <hip/hip_runtime.h> that lets the *unmodified* kernel sources under
// efficientteacher_amd/csrc/ be compiled for the host (clang++ -x c++) and executed on a CPU,
// one workgroup at a time, every work-item a fiber.  Purpose: this container has no GPU and GPU
// minutes are rationed, so indexing / ordering / fragment-layout bugs are flushed out here before
// a kernel ever reaches an MI355X.  tests/ compares the emulated kernels with oracle/ on CPU
// (`-m "not gpu"`), and the same tests run the real libet_hip.so on the GPU (`-m gpu`).
//
// Modelled: 64-lane wavefronts, __syncthreads, wave cross-lane ops (all live lanes of the wave
// must execute them together), LDS as block-shared statics, atomics, and the MFMA fragment
// layouts of /opt/skills/guides/cdna_hip_programming.md section 3 (C/D: col = lane&31,
// row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) for 32x32; col = lane&15, row = 4*(lane>>4)+reg for
// 16x16).  NOT modelled: timing, memory coalescing, bank conflicts, data races between waves.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__ __restrict

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_emu { unsigned x, y, z; };

typedef void* hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1 };
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
struct hipDeviceProp_t { int multiProcessorCount; };
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { p->multiProcessorCount = 256; return hipSuccess; }

struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
struct ushort4 { unsigned short x, y, z, w; };
static inline float2 make_float2(float x, float y) { return {x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return {x, y}; }
static inline int4 make_int4(int x, int y, int z, int w) { return {x, y, z, w}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return {x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return {x, y, z, w}; }

namespace emu {

extern "C" void emu_switch(void** save_sp, void* load_sp);

enum State { RUN, WAIT_BLOCK, WAIT_WAVE, DONE };

struct Fiber {
    void* sp;
    State st;
    uint3_emu tid;
    int lin, lane, wave;
};

struct Wave {
    int alive, arrived;
    alignas(16) unsigned char buf[64][64];   // per-lane scratch for cross-lane ops
    alignas(16) unsigned char buf2[64][64];
};

// LDS-DMA (global_load_lds) in flight: hardware lands the bytes some time between the issue and the s_waitcnt
// vmcnt(N) that retires it.  ET_EMU_DMA=late defers every landing to that wait (the LATEST legal moment: a
// reader that did not wait + barrier sees stale LDS), the default lands at issue (the EARLIEST legal moment:
// a writer that did not let every reader finish clobbers live data).  Tests run both.
struct PendingDma { void* dst; unsigned size; unsigned char data[16]; };

struct Block {
    std::vector<Fiber> fibers;
    std::vector<Wave> waves;
    std::vector<std::vector<PendingDma>> dmaq;   // per fiber, oldest first
    int alive, arrived_block;
    uint3_emu bid;
    dim3 bdim, gdim;
    void* sched_sp;
    Fiber* cur;
    std::function<void()> body;
};

Block& blk();
static inline Fiber& cur() { return *blk().cur; }
void yield_to_sched();
void run_grid(dim3 grid, dim3 block, std::function<void()> body);

static inline void block_sync() {
    Block& b = blk();
    b.cur->st = WAIT_BLOCK;
    if (++b.arrived_block == b.alive) {
        for (auto& f : b.fibers) if (f.st == WAIT_BLOCK) f.st = RUN;
        b.arrived_block = 0;
        return;
    }
    yield_to_sched();
}
static inline void wave_sync() {
    Block& b = blk();
    Wave& w = b.waves[b.cur->wave];
    b.cur->st = WAIT_WAVE;
    if (++w.arrived == w.alive) {
        int base = b.cur->wave * 64;
        for (int i = base; i < base + 64 && i < (int)b.fibers.size(); ++i)
            if (b.fibers[i].st == WAIT_WAVE) b.fibers[i].st = RUN;
        w.arrived = 0;
        return;
    }
    yield_to_sched();
}
static inline Wave& mywave() { Block& b = blk(); return b.waves[b.cur->wave]; }

template <typename T> static inline T xlane(T v, int src) {
    static_assert(sizeof(T) <= 64, "xlane payload");
    Wave& w = mywave();
    int lane = cur().lane;
    memcpy(w.buf[lane], &v, sizeof(T));
    wave_sync();
    T r;
    memcpy(&r, w.buf[src & 63], sizeof(T));
    wave_sync();
    return r;
}

}  // namespace emu

#define threadIdx (emu::cur().tid)
#define blockIdx (emu::blk().bid)
#define blockDim (emu::blk().bdim)
#define gridDim (emu::blk().gdim)
#define warpSize 64

#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) \
    emu::run_grid((grid), (block), [=]() { kern(__VA_ARGS__); })

namespace emu {
bool dma_late();
static inline void dma_retire(int keep) {      // land the oldest entries until at most `keep` remain in flight
    Block& b = blk();
    auto& q = b.dmaq[b.cur->lin];
    if ((int)q.size() <= keep) return;
    const size_t n = q.size() - (size_t)keep;
    for (size_t i = 0; i < n; ++i) memcpy(q[i].dst, q[i].data, q[i].size);
    q.erase(q.begin(), q.begin() + n);
}
}  // namespace emu
// __syncthreads() = fence + s_barrier: with an LDS-DMA in flight the fence is an s_waitcnt vmcnt(0)
// (cdna_hip_programming.md, "Pipelining across barriers"); the bare s_barrier builtin waits for nothing
static inline void __syncthreads() { emu::dma_retire(0); emu::block_sync(); }
static inline void __builtin_amdgcn_s_barrier_emu() { emu::block_sync(); }
#define __builtin_amdgcn_s_barrier __builtin_amdgcn_s_barrier_emu
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#ifndef __clang__
#define __builtin_nontemporal_load(p) (*(p))
#endif
#define __builtin_amdgcn_wave_barrier() emu::wave_sync()
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_sched_group_barrier(a, b, c) ((void)0)

template <typename T> static inline T __shfl(T v, int src, int width = 64) {
    int lane = emu::cur().lane;
    int s = (lane & ~(width - 1)) | (src & (width - 1));
    return emu::xlane(v, s);
}
template <typename T> static inline T __shfl_xor(T v, int mask, int width = 64) {
    int lane = emu::cur().lane;
    int s = lane ^ mask;
    if ((s & ~(width - 1)) != (lane & ~(width - 1))) s = lane;
    return emu::xlane(v, s);
}
template <typename T> static inline T __shfl_down(T v, unsigned d, int width = 64) {
    int lane = emu::cur().lane;
    int s = lane + (int)d;
    if ((s & ~(width - 1)) != (lane & ~(width - 1))) s = lane;
    return emu::xlane(v, s);
}
template <typename T> static inline T __shfl_up(T v, unsigned d, int width = 64) {
    int lane = emu::cur().lane;
    int s = lane - (int)d;
    if (s < 0 || (s & ~(width - 1)) != (lane & ~(width - 1))) s = lane;
    return emu::xlane(v, s);
}
static inline unsigned long long __ballot(int pred) {
    emu::Wave& w = emu::mywave();
    int lane = emu::cur().lane;
    w.buf[lane][0] = pred ? 1 : 0;
    emu::wave_sync();
    unsigned long long m = 0;
    emu::Block& b = emu::blk();
    int base = b.cur->wave * 64;
    for (int i = 0; i < 64 && base + i < (int)b.fibers.size(); ++i)
        if (b.fibers[base + i].st != emu::DONE && w.buf[i][0]) m |= 1ull << i;
    emu::wave_sync();
    return m;
}
static inline int __any(int p) { return __ballot(p) != 0; }
static inline int __all(int p) { return __ballot(!p) == 0; }
template <typename T> static inline T __builtin_amdgcn_readfirstlane_emu(T v) {
    emu::Block& b = emu::blk();
    int base = b.cur->wave * 64, first = 0;
    for (int i = 0; i < 64 && base + i < (int)b.fibers.size(); ++i)
        if (b.fibers[base + i].st != emu::DONE) { first = i; break; }
    return emu::xlane(v, first);
}
#define __builtin_amdgcn_readfirstlane __builtin_amdgcn_readfirstlane_emu
// v_readlane_b32: the value of `v` in lane `src_lane` (wave-uniform index)
template <typename T> static inline T __builtin_amdgcn_readlane_emu(T v, int src_lane) { return emu::xlane(v, src_lane & 63); }
#define __builtin_amdgcn_readlane __builtin_amdgcn_readlane_emu
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline int __ffsll(unsigned long long x) { return __builtin_ffsll((long long)x); }
static inline int __clzll(unsigned long long x) { return x ? __builtin_clzll(x) : 64; }
static inline int __clz(unsigned x) { return x ? __builtin_clz(x) : 32; }

template <typename T> static inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
template <typename T> static inline T unsafeAtomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { auto o = *p; *p = o + v; return o; }
template <typename T> static inline T atomicMax(T* p, T v) { T o = *p; *p = std::max(o, v); return o; }
template <typename T> static inline T atomicMin(T* p, T v) { T o = *p; *p = std::min(o, v); return o; }
template <typename T> static inline T atomicOr(T* p, T v) { T o = *p; *p = o | v; return o; }
template <typename T> static inline T atomicExch(T* p, T v) { T o = *p; *p = v; return o; }
template <typename T> static inline T atomicCAS(T* p, T c, T v) { T o = *p; if (o == c) *p = v; return o; }
static inline void __threadfence() {}
static inline void __threadfence_block() {}

#define __expf(x) expf(x)
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
#define __logf(x) logf(x)
static inline float __fdividef(float a, float b) { return a / b; }
static inline float __frcp_rn(float a) { return 1.0f / a; }
static inline float rsqrtf(float a) { return 1.0f / sqrtf(a); }
static inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
static inline unsigned __float_as_uint(float f) { unsigned i; memcpy(&i, &f, 4); return i; }
static inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
static inline float __uint_as_float(unsigned i) { float f; memcpy(&f, &i, 4); return f; }
// an addition with its own rounding (never contracted into a multiply-add)
static inline float __fadd_rn(float a, float b) { volatile float s = a + b; return s; }
static inline double __longlong_as_double(long long i) { double f; memcpy(&f, &i, 8); return f; }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
using std::max;
using std::min;

// LDS-DMA: every lane copies `size` bytes from its own global address to  M0_base + offset + lane*size,
// where M0_base is the LDS pointer of the first live lane (the compiler readfirstlane's it).
static inline void emu_global_load_lds(const void* g, void* l, unsigned size, int offset) {
    void* base = __builtin_amdgcn_readfirstlane_emu(l);
    char* dst = (char*)base + offset + (size_t)emu::cur().lane * size;
    if (!emu::dma_late() || size > 16) { memcpy(dst, g, size); return; }
    emu::PendingDma d; d.dst = dst; d.size = size; memcpy(d.data, g, size);
    emu::Block& b = emu::blk();
    b.dmaq[b.cur->lin].push_back(d);
}
#define __builtin_amdgcn_global_load_lds(g, l, size, off, aux) emu_global_load_lds((const void*)(g), (void*)(l), (size), (off))
// the same through a buffer descriptor (buffer_load_dwordx4 ... offen lds), semantics PROBED on hardware (tools/probe/probe_bufdma.py):
// byte offset = voffset + soffset + inst offset (unsigned); a dword whose end lies beyond num_records reads as ZERO -- and the zero IS
// written to LDS; a "negative" voffset is a huge unsigned one, i.e. out of range
struct emu_buffer_rsrc { const char* base; unsigned num_records; };
typedef emu_buffer_rsrc __amdgpu_buffer_rsrc_t;
static inline emu_buffer_rsrc __builtin_amdgcn_make_buffer_rsrc(void* p, short stride, unsigned num, unsigned flags) {
    (void)stride; (void)flags;
    return emu_buffer_rsrc{(const char*)p, num};
}
static inline void emu_buffer_load_lds(emu_buffer_rsrc r, void* l, unsigned size, int voff, int soff, int ioff) {
    unsigned char tmp[16];
    const unsigned long long off = (unsigned long long)(unsigned)voff + (unsigned long long)(unsigned)soff + (unsigned long long)(unsigned)ioff;
    for (unsigned d = 0; d < size / 4; ++d) {
        if (off + 4ull * (d + 1) <= (unsigned long long)r.num_records) memcpy(tmp + 4 * d, r.base + off + 4 * d, 4);
        else memset(tmp + 4 * d, 0, 4);
    }
    emu_global_load_lds(tmp, l, size, 0);
}
#define __builtin_amdgcn_raw_ptr_buffer_load_lds(r, l, size, voff, soff, ioff, aux) emu_buffer_load_lds((r), (void*)(l), (size), (voff), (soff), (ioff))
// s_waitcnt immediate (gfx9): vmcnt = bits [3:0] | bits [15:14] << 4; only the LDS-DMA queue is modelled
#define __builtin_amdgcn_s_waitcnt(x) emu::dma_retire((int)(((x) & 0xF) | ((((x) >> 14) & 3) << 4)))

// ds_read_b64_tr_b16 (gfx950), semantics PROBED on hardware (tools/probe/probe_tr.py, profiles/r01_probe_tr.json):
// every lane loads the 8 bytes (4 x 16-bit) at its own LDS address; inside each 16-lane group the
// 16 x 4 elements are redistributed as  out[lane c][j] = in[lane 4*j + c/4][c % 4]   (a 4x16 -> 16x4 transpose
// when lane t holds row t/4, columns 4*(t%4)..+3 of a row-major 4x16 block).
typedef short emu_s16x4 __attribute__((ext_vector_type(4)));
static inline emu_s16x4 emu_ds_read_tr16_b64(const void* p) {
    emu::Wave& w = emu::mywave();
    const int lane = emu::cur().lane;
    memcpy(w.buf[lane], p, 8);
    emu::wave_sync();
    emu_s16x4 r;
    const int gb = lane & ~15, c = lane & 15;
    for (int j = 0; j < 4; ++j) {
        short v;
        memcpy(&v, w.buf[gb + 4 * j + (c >> 2)] + 2 * (c & 3), 2);
        r[j] = v;
    }
    emu::wave_sync();
    return r;
}
#define __builtin_amdgcn_ds_read_tr16_b64_v4i16(p) emu_ds_read_tr16_b64((const void*)(p))

// ----------------------------------------------------------------------------------- MFMA
typedef __attribute__((ext_vector_type(16))) float emu_f32x16;
typedef __attribute__((ext_vector_type(4))) float emu_f32x4;

static inline float emu_bf16_to_f32(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
static inline float emu_f16_to_f32(unsigned short h) { _Float16 x; memcpy(&x, &h, 2); return (float)x; }

// D(32x32) += A(32xK) * B(Kx32); lane l supplies A[l&31][g*KL + i], B[g*KL + i][l&31], g = l>>5.
template <int KL, typename AB, typename CVT>
static inline emu_f32x16 emu_mfma32(AB a, AB b, emu_f32x16 c, CVT cvt) {
    emu::Wave& w = emu::mywave();
    int lane = emu::cur().lane;
    memcpy(w.buf[lane], &a, sizeof(AB));
    memcpy(w.buf2[lane], &b, sizeof(AB));
    emu::wave_sync();
    const int col = lane & 31;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        float acc = c[r];
        for (int g = 0; g < 2; ++g) {
            AB av, bv;
            memcpy(&av, w.buf[row + 32 * g], sizeof(AB));
            memcpy(&bv, w.buf2[col + 32 * g], sizeof(AB));
            for (int i = 0; i < KL; ++i) acc = fmaf(cvt(av, i), cvt(bv, i), acc);
        }
        c[r] = acc;
    }
    emu::wave_sync();
    return c;
}
// D(16x16) += A(16xK) * B(Kx16); lane l supplies A[l&15][g*KL+i], B[g*KL+i][l&15], g = l>>4.
template <int KL, typename AB, typename CVT>
static inline emu_f32x4 emu_mfma16(AB a, AB b, emu_f32x4 c, CVT cvt) {
    emu::Wave& w = emu::mywave();
    int lane = emu::cur().lane;
    memcpy(w.buf[lane], &a, sizeof(AB));
    memcpy(w.buf2[lane], &b, sizeof(AB));
    emu::wave_sync();
    const int col = lane & 15;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * (lane >> 4) + r;
        float acc = c[r];
        for (int g = 0; g < 4; ++g) {
            AB av, bv;
            memcpy(&av, w.buf[row + 16 * g], sizeof(AB));
            memcpy(&bv, w.buf2[col + 16 * g], sizeof(AB));
            for (int i = 0; i < KL; ++i) acc = fmaf(cvt(av, i), cvt(bv, i), acc);
        }
        c[r] = acc;
    }
    emu::wave_sync();
    return c;
}
struct emu_cvt_bf16 { template <typename V> float operator()(const V& v, int i) const { unsigned short h; memcpy(&h, (const char*)&v + 2 * i, 2); return emu_bf16_to_f32(h); } };
struct emu_cvt_f16 { template <typename V> float operator()(const V& v, int i) const { unsigned short h; memcpy(&h, (const char*)&v + 2 * i, 2); return emu_f16_to_f32(h); } };
struct emu_cvt_f32 { float operator()(const float& v, int) const { return v; } };

#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) emu_mfma32<8>((a), (b), (c), emu_cvt_bf16())
#define __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z) emu_mfma32<8>((a), (b), (c), emu_cvt_f16())
#define __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, x, y, z) emu_mfma32<1>((float)(a), (float)(b), (c), emu_cvt_f32())
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, x, y, z) emu_mfma16<8>((a), (b), (c), emu_cvt_bf16())
#define __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, x, y, z) emu_mfma16<1>((float)(a), (float)(b), (c), emu_cvt_f32())
