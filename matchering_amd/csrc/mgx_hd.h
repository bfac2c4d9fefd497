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

constexpr int ilog2(int v) { return v <= 1 ? 0 : 1 + ilog2(v >> 1); }
constexpr int bitrev(int v, int bits) {
    int r = 0;
    for (int i = 0; i < bits; ++i) r |= ((v >> i) & 1) << (bits - 1 - i);
    return r;
}

}  // namespace mgx
