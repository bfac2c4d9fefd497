// Host/device portability shim.
//
// Kernel bodies in this directory are written as per-thread "phase" functions
// (everything between two workgroup barriers) over explicit per-thread state.
// Under hipcc they are inlined into __global__ kernels; under a plain host
// compiler (-DMGX_HOST_EMU, tests/emu) the same functions are driven by a loop
// over thread ids, which lets the index arithmetic be checked against the oracle
// on a machine without a GPU.  The emulation is test infrastructure only: the
// shipped library contains no host implementation of any kernel.
#pragma once

#include <cstdint>
#include <cmath>

#if defined(__HIPCC__) && !defined(MGX_HOST_EMU)
#include <hip/hip_runtime.h>
#define MGX_HD __host__ __device__ __forceinline__
#define MGX_D __device__ __forceinline__
#define MGX_UNROLL _Pragma("unroll")
#else
#define MGX_HD inline
#define MGX_D inline
#define MGX_UNROLL
#ifndef __HIPCC__
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct double2 { double x, y; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
#endif
#endif

// A point the compiler's scheduler may not move instructions across (device code only): keeps a phase's
// global loads from being issued all at once when the registers to hold them do not exist.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MGX_HOST_EMU)
#define MGX_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define MGX_SCHED_FENCE() ((void)0)
#endif

// A value the compiler must take as new at this point (device code only): whatever is derived from it -- a load's
// address -- cannot be computed, or the load issued, any earlier.
static MGX_HD int mgx_opaque(int v) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MGX_HOST_EMU)
    asm volatile("" : "+v"(v));
#endif
    return v;
}

// ... and a value that must EXIST at this point: keeps the arithmetic that produces it from sinking below later
// loads (whose results would then all be alive at once).
static MGX_HD void mgx_pin(float2& v) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MGX_HOST_EMU)
    asm volatile("" : "+v"(v.x), "+v"(v.y));
#endif
}

// float32 complex helpers: their adds may fuse with the multiply that feeds them (fft2.h: the one place
// of a library built with -ffp-contract=off where contraction is let in)
#if defined(__clang__)
#pragma clang fp contract(fast)
#endif
namespace mgx {

MGX_HD float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
MGX_HD float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
MGX_HD float2 cmul(float2 a, float2 b) {
    return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
}
// a * conj(b)
MGX_HD float2 cmulc(float2 a, float2 b) {
    return make_float2(fmaf(a.x, b.x, a.y * b.y), fmaf(a.y, b.x, -a.x * b.y));
}
MGX_HD float2 cconj(float2 a) { return make_float2(a.x, -a.y); }
// multiply by +i / -i
MGX_HD float2 cmul_i(float2 a) { return make_float2(-a.y, a.x); }
MGX_HD float2 cmul_mi(float2 a) { return make_float2(a.y, -a.x); }

}  // namespace mgx
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
namespace mgx {


// ---------------------------------------------------------------------------
// Uniform-base global memory views.
//
// On gfx950 a view is a buffer resource (4 SGPRs): every access is
//     buffer_load/store  vdata, voffset(VGPR, bytes), rsrc, soffset(SGPR, bytes)
// so a thread keeps ONE 32-bit lane offset and the per-access displacement is a scalar add
// -- no 64-bit vector address arithmetic and nothing for the compiler to hoist into VGPRs.
// Offsets are unsigned 32-bit byte counts (a 15-minute 96 kHz stereo float track is 691 MB).
// The hardware range check IS relied upon at the ends of a track: an access is out of range when
// lane offset + scalar displacement + size exceeds the view (the sum does not wrap), loads then
// return zero and stores are dropped (tools/micro/buffer_range.hip).  A lane offset that has wrapped
// below zero is therefore always out of range, whatever the displacement: code that may start
// before the view adds the displacement into the lane offset instead (`ld_f2_or_zero`).
// Under the host emulation a view is a pointer plus its size, with the same rule.
// ---------------------------------------------------------------------------
#if defined(__HIPCC__) && !defined(MGX_HOST_EMU)
struct MemView {
    __amdgpu_buffer_rsrc_t r;
};
__device__ __forceinline__ MemView mem_view(const void* p, long long bytes) {
    const long long cap = 0xfffffff0ll;
    MemView v;
    v.r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)(unsigned)(bytes < cap ? bytes : cap),
                                            0x00020000);
    return v;
}
// AUX: cache-policy bits of the buffer instruction (gfx950: 1 = sc0, 2 = nt, 16 = sc1)
template <int AUX = 0>
__device__ __forceinline__ float2 ld_f2(MemView m, unsigned voff, unsigned soff) {
    typedef unsigned u2_t __attribute__((ext_vector_type(2)));
    const u2_t t = __builtin_amdgcn_raw_buffer_load_b64(m.r, voff, soff, AUX);
    const unsigned a = t.x, b = t.y;
    return make_float2(__uint_as_float(a), __uint_as_float(b));
}
// Load whose lane offset may lie outside the view (before its start the 32-bit offset has wrapped to
// ~4 G, past its end it exceeds the size): the buffer range check returns zeros for such lanes, which
// is exactly the zero padding a convolution wants at the ends of a track.  The whole offset sits in
// the VGPR so that the check sees all of it.
template <int AUX = 0>
__device__ __forceinline__ float2 ld_f2_or_zero(MemView m, unsigned voff) {
    typedef unsigned u2_t __attribute__((ext_vector_type(2)));
    const u2_t t = __builtin_amdgcn_raw_buffer_load_b64(m.r, voff, 0, AUX);
    const unsigned a = t.x, b = t.y;
    return make_float2(__uint_as_float(a), __uint_as_float(b));
}
// Stores and 4-byte loads of the same kind: lanes whose offset lies outside the view are dropped /
// read zero by the buffer range check, so kernels need no separate code path for the ends of a track.
template <int AUX = 0>
__device__ __forceinline__ void st_f2_in_range(MemView m, unsigned voff, float2 v) {
    typedef unsigned u2_t __attribute__((ext_vector_type(2)));
    u2_t t;
    t.x = __float_as_uint(v.x);
    t.y = __float_as_uint(v.y);
    __builtin_amdgcn_raw_buffer_store_b64(t, m.r, voff, 0, AUX);
}
template <int AUX = 0>
__device__ __forceinline__ void st_f1_in_range(MemView m, unsigned voff, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), m.r, voff, 0, AUX);
}
template <int AUX = 0>
__device__ __forceinline__ float ld_f1_or_zero(MemView m, unsigned voff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(m.r, voff, 0, AUX));
}
template <int AUX = 0>
__device__ __forceinline__ void st_f2(MemView m, unsigned voff, unsigned soff, float2 v) {
    typedef unsigned u2_t __attribute__((ext_vector_type(2)));
    u2_t t;
    t.x = __float_as_uint(v.x);
    t.y = __float_as_uint(v.y);
    __builtin_amdgcn_raw_buffer_store_b64(t, m.r, voff, soff, AUX);
}
template <int AUX = 0>
__device__ __forceinline__ float ld_f1(MemView m, unsigned voff, unsigned soff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(m.r, voff, soff, AUX));
}
template <int AUX = 0>
__device__ __forceinline__ void st_f1(MemView m, unsigned voff, unsigned soff, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), m.r, voff, soff, AUX);
}
#else
// Host emulation with the semantics of the hardware range check (tools/micro/buffer_range.hip): an
// access is out of range when lane offset + scalar displacement + size exceeds the view, WITHOUT
// 32-bit wrap-around of the sum; loads then return zero and stores are dropped.
struct MemView {
    char* p;
    unsigned long long bytes;
};
inline MemView mem_view(const void* p, long long bytes) {
    return MemView{const_cast<char*>(static_cast<const char*>(p)), (unsigned long long)(bytes > 0 ? bytes : 0)};
}
inline bool mem_in_range(MemView m, unsigned voff, unsigned soff, unsigned size) {
    return (unsigned long long)voff + soff + size <= m.bytes;
}
template <int AUX = 0>
inline float2 ld_f2(MemView m, unsigned voff, unsigned soff) {
    if (!mem_in_range(m, voff, soff, 8)) return float2{0.f, 0.f};
    return *reinterpret_cast<const float2*>(m.p + (size_t)voff + (size_t)soff);
}
template <int AUX = 0>
inline float2 ld_f2_or_zero(MemView m, unsigned voff) { return ld_f2<AUX>(m, voff, 0); }
template <int AUX = 0>
inline void st_f2(MemView m, unsigned voff, unsigned soff, float2 v) {
    if (mem_in_range(m, voff, soff, 8)) *reinterpret_cast<float2*>(m.p + (size_t)voff + (size_t)soff) = v;
}
template <int AUX = 0>
inline void st_f2_in_range(MemView m, unsigned voff, float2 v) { st_f2<AUX>(m, voff, 0, v); }
template <int AUX = 0>
inline float ld_f1(MemView m, unsigned voff, unsigned soff) {
    return mem_in_range(m, voff, soff, 4) ? *reinterpret_cast<const float*>(m.p + (size_t)voff + (size_t)soff) : 0.f;
}
template <int AUX = 0>
inline float ld_f1_or_zero(MemView m, unsigned voff) { return ld_f1<AUX>(m, voff, 0); }
template <int AUX = 0>
inline void st_f1(MemView m, unsigned voff, unsigned soff, float v) {
    if (mem_in_range(m, voff, soff, 4)) *reinterpret_cast<float*>(m.p + (size_t)voff + (size_t)soff) = v;
}
template <int AUX = 0>
inline void st_f1_in_range(MemView m, unsigned voff, float v) { st_f1<AUX>(m, voff, 0, v); }
#endif

// Streaming stores: data this kernel will not read again goes out non-temporal so that it does not
// evict what is still to be re-read from the L2 (the convolution's input frames, the limiter's).
MGX_HD void st_stream(float4* p, float4 v) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MGX_HOST_EMU) && !defined(MGX_NO_STREAM_STORES)
    typedef float v4_t __attribute__((ext_vector_type(4)));
    v4_t t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<v4_t*>(p));
#else
    *p = v;
#endif
}
MGX_HD void st_stream(float2* p, float2 v) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MGX_HOST_EMU) && !defined(MGX_NO_STREAM_STORES)
    typedef float v2_t __attribute__((ext_vector_type(2)));
    v2_t t = {v.x, v.y};
    __builtin_nontemporal_store(t, reinterpret_cast<v2_t*>(p));
#else
    *p = v;
#endif
}

// sin(pi x), cos(pi x)
MGX_HD void sincos_pi(float x, float& sn, float& cs) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MGX_HOST_EMU)
    sincospif(x, &sn, &cs);
#else
    sn = (float)sin(3.14159265358979323846 * (double)x);
    cs = (float)cos(3.14159265358979323846 * (double)x);
#endif
}

// sqrt to 1 ulp (v_sqrt_f32) for magnitudes; the library sqrtf adds a denormal-safe refinement
// sequence that costs ~15 instructions
MGX_HD float fast_sqrt(float x) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MGX_HOST_EMU)
    return __builtin_amdgcn_sqrtf(x);
#else
    return sqrtf(x);
#endif
}

// 1/x to 1 ulp (v_rcp_f32); the division operator expands to a ~10-instruction IEEE sequence
MGX_HD float fast_rcp(float x) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MGX_HOST_EMU)
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.0f / x;
#endif
}

// maximum of two NON-NEGATIVE floats (no NaN, no -0): their order is the order of their bit patterns, so one integer
// maximum does it.  fmaxf on a value that comes from memory costs a second instruction on this target (the compiler
// quiets a possible signalling NaN first: v_max_f32 v, v, v), and the limiter's window maxima are nothing but that.
MGX_HD float pmax(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MGX_HOST_EMU)
    const int x = __float_as_int(a), y = __float_as_int(b);
    return __int_as_float(x > y ? x : y);
#else
    return a > b ? a : b;
#endif
}

constexpr int ilog2(int v) { return v <= 1 ? 0 : 1 + ilog2(v >> 1); }
constexpr int bitrev(int v, int bits) {
    int r = 0;
    for (int i = 0; i < bits; ++i) r |= ((v >> i) & 1) << (bits - 1 - i);
    return r;
}

}  // namespace mgx
