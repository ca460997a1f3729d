// Shared device helpers for the gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ET_WAVE 64

#define ET_CHECK_LAUNCH()                                    \
    do {                                                     \
        hipError_t e__ = hipGetLastError();                  \
        if (e__ != hipSuccess) return -100 - (int)e__;       \
    } while (0)

static inline int et_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// ---- bf16 <-> f32 (round-to-nearest-even), storage type is uint16_t -------------------------
__device__ __forceinline__ float et_bf2f(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
#if defined(__HIP_DEVICE_COMPILE__)
// gfx950 converts in hardware: v_cvt_pk_bf16_f32, round-to-nearest-even, two values per instruction (the bit-twiddling form
// below is 7-8 VALU instructions per VALUE -- it was a third of the conv epilogue's and of the BN passes' instruction count)
typedef __bf16 et_bf16x2_t __attribute__((ext_vector_type(2)));
typedef float et_f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t et_pack_bf2(float lo, float hi) {
    const et_f32x2_t f = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, et_bf16x2_t));
}
__device__ __forceinline__ uint16_t et_f2bf(float f) { return (uint16_t)(et_pack_bf2(f, 0.f) & 0xffffu); }
#else
// host-side (CPU emulator) form of the same rounding
__device__ __forceinline__ uint16_t et_f2bf(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // quiet NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__device__ __forceinline__ uint32_t et_pack_bf2(float lo, float hi) {
    return (uint32_t)et_f2bf(lo) | ((uint32_t)et_f2bf(hi) << 16);
}
#endif

// register budget of a kernel as "n waves per SIMD" (device pass only: the host pass and the CPU emulator see a plain function)
#if defined(__HIP_DEVICE_COMPILE__)
#define ET_WAVES_PER_EU(n) __attribute__((amdgpu_waves_per_eu(n, n)))
#else
#define ET_WAVES_PER_EU(n)
#endif

// a wave-uniform value the optimiser must treat as unknown from here on (keeps address arithmetic that depends on it where it is
// written instead of hoisted and kept alive across phases)
__device__ __forceinline__ int et_opaque_uniform(int v) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+s"(v));
#endif
    return v;
}

// ---- fp16 (IEEE half), storage type et_f16 ---------------------------------------------------------------------------
// The reference's own reduced precision (torch.cuda.amp autocast + GradScaler, trainer/trainer.py:248,348,399-400): same MFMA rate as
// bf16 (v_mfma_f32_32x32x16_f16), 11-bit significand, 5-bit exponent -- hence the loss scaling.  A distinct C++ type so that the
// kernels can be instantiated per format: uint16_t keeps meaning bf16 everywhere.
struct et_f16 { uint16_t v; };
typedef _Float16 et_half2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float et_h2f(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ uint16_t et_f2h(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }      // round-to-nearest-even
__device__ __forceinline__ uint32_t et_pack_h2(float lo, float hi) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    const f32x2_ f = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, et_half2_t));        // v_cvt_pk_f16_f32 (RNE) on gfx950
#else
    return (uint32_t)et_f2h(lo) | ((uint32_t)et_f2h(hi) << 16);
#endif
}

// 16-bit storage formats, two values per 32-bit word: what the vector kernels need to be format-generic
//   lo / hi(w)   the two values of a word as fp32          pack(lo, hi)   two fp32 -> one word (RNE)
template <typename T> struct et_lp;
template <> struct et_lp<uint16_t> {
    __device__ static __forceinline__ float lo(uint32_t w) { return __uint_as_float(w << 16); }
    __device__ static __forceinline__ float hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
    __device__ static __forceinline__ uint32_t pack(float a, float b) { return et_pack_bf2(a, b); }
};
template <> struct et_lp<et_f16> {
    __device__ static __forceinline__ float lo(uint32_t w) { return et_h2f((uint16_t)(w & 0xffffu)); }
    __device__ static __forceinline__ float hi(uint32_t w) { return et_h2f((uint16_t)(w >> 16)); }
    __device__ static __forceinline__ uint32_t pack(float a, float b) { return et_pack_h2(a, b); }
};
template <typename T> struct et_is_lp { static constexpr bool value = false; };
template <> struct et_is_lp<uint16_t> { static constexpr bool value = true; };
template <> struct et_is_lp<et_f16> { static constexpr bool value = true; };

// element-type traits: T = float (parity mode), uint16_t holding bf16 (performance mode) or et_f16 (the reference's fp16 recipe)
template <typename T> struct et_elem;
template <> struct et_elem<float> {
    static constexpr int VEC = 4;  // elements per 16-byte vector
    __device__ static __forceinline__ float ld(float v) { return v; }
    __device__ static __forceinline__ float st(float v) { return v; }
};
template <> struct et_elem<uint16_t> {
    static constexpr int VEC = 8;
    __device__ static __forceinline__ float ld(uint16_t v) { return et_bf2f(v); }
    __device__ static __forceinline__ uint16_t st(float v) { return et_f2bf(v); }
};
template <> struct et_elem<et_f16> {
    static constexpr int VEC = 8;
    __device__ static __forceinline__ float ld(et_f16 v) { return et_h2f(v.v); }
    __device__ static __forceinline__ et_f16 st(float v) { et_f16 r; r.v = et_f2h(v); return r; }
};

// ---- wave / block reductions ------------------------------------------------------------------
__device__ __forceinline__ float et_wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
__device__ __forceinline__ double et_wave_sum_d(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
__device__ __forceinline__ int et_wave_sum_i(int v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

__device__ __forceinline__ float et_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// ---- LDS-DMA (gfx950 global_load_lds_dwordx4): 16 bytes per lane straight from global memory into LDS,
// no VGPR staging and no ds_write.  The LDS destination is WAVE-UNIFORM base + lane*16: pass the same
// `lds_wave_base` in every lane of the wave; the per-lane part is the global address.
__device__ __forceinline__ void et_glds16(const void* g, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
// the same with the non-temporal cache policy (aux = 2): for a stream that ONE CU reads once (MI355X_MICROARCH.md "nt-weights")
__device__ __forceinline__ void et_glds16_nt(const void* g, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 2);
}
// LDS-DMA through a BUFFER descriptor (buffer_load_dwordx4 ... offen lds): address = base + soffset (SGPR, wave-uniform) + voffset
// (VGPR, per lane, unsigned bytes); a lane whose voffset + soffset + 16 exceeds num_records is OUT OF RANGE and the hardware writes
// ZEROS into its LDS slot (probed on gfx950, tools/probe/probe_bufdma.py, profiles/r06_probe_bufdma.txt: soffset IS part of the range
// check, per dword; a negative voffset is out of range).  Against the flat form (et_glds16) a padding / tail lane needs no zero
// page and no 64-bit select: an all-ones voffset; rows beyond a tensor's end fall out of range by themselves; and the wave-uniform
// part of the address (tap, channel chunk, half-tile row block) travels in an SGPR instead of per-lane VALU.
typedef __amdgpu_buffer_rsrc_t et_rsrc;
__device__ __forceinline__ et_rsrc et_make_rsrc(const void* base, unsigned num_bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, num_bytes, 0x00020000);
}
__device__ __forceinline__ void et_bufdma16(et_rsrc r, unsigned voff, unsigned soff, void* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, (int)voff, (int)soff, 0, 0);
}
// the same with the non-temporal cache policy (aux = 2), as et_glds16_nt
__device__ __forceinline__ void et_bufdma16_nt(et_rsrc r, unsigned voff, unsigned soff, void* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, (int)voff, (int)soff, 0, 2);
}
// s_waitcnt vmcnt(0): all of this wave's LDS-DMA writes have landed (expcnt / lgkmcnt left at max)
__device__ __forceinline__ void et_wait_vmem() { __builtin_amdgcn_s_waitcnt(0x0F70); }
// The wait in front of a bare s_barrier that hands an LDS slot BACK to the LDS-DMA (the single-barrier rings: the slot one step read is
// rewritten by the pieces issued right behind the NEXT step's barrier): vmcnt <= N AND lgkmcnt(0).  Program order is not enough there:
// s_barrier orders the ISSUE of the ds_reads in front of it, not their completion, and the scheduler may sink a step's last MFMAs --
// with the lgkmcnt wait the compiler attaches to them -- below the barrier.  A faster wave then passes the barrier and its piece can
// land in the slot while this wave's reads of it are still queued in the LDS.  It happened in the buffer-descriptor form of the row-shift
// kernels (pieces issued two instructions behind the barrier, three workgroups per CU, weight rows hot in the vector L1): one wave's
// tile computed on stale / foreign weight rows in ~15 % of the launches on 160-pixel-wide maps; 0 of 540 with this wait
// (profiles/r06_lds_ring_war_race.txt).  The flat-address forms, the K-chunk ring and the 1x1 stream kernel have the same source shape;
// their compiled code happened to keep the wait in front of the barrier (tests/test_lds_ring_barriers.py reads the disassembly of the
// built library for exactly this), and they now state it.  Step-neutral (same file).
template <int N> __device__ __forceinline__ void et_wait_vmem_le_lds_read_done() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    __builtin_amdgcn_s_waitcnt(0x0070 | (N & 0xF) | ((N >> 4) << 14));
}
__device__ __forceinline__ void et_wait_vmem_lds_read_done() { __builtin_amdgcn_s_waitcnt(0x0070); }
