// The two radix-8 middle passes of a four-pass Fft2 plan as ONE phase: the exchange between them goes through the
// wave's lanes (v_permlane32_swap / v_permlane16_swap, gfx950) instead of through the LDS.
//
// Why: the 16384-point kernels spend as many cycles of the CU's one LDS as of a SIMD's VALU (tools/lds_cycles.py,
// profiles/r06_a_*): a store costs 6 LDS cycles per 8 bytes and lane (the VGPR -> LDS path), and a transform makes three
// round trips per direction.  Between pass 1 (sub-transforms of 1024 points, stride 128) and pass 2 (sub-transforms of
// 128 points, stride 16) a wave's 1024 points move only WITHIN GROUPS OF FOUR LANES 16 apart:
//
//   pass 1, lane l = n' + 16 k'' (n' < 16), butterfly c in {0, 1}:  n = l + 64 c, outputs X_q at point n + 128 q;
//   pass 2, block b (= q), column n':  inputs at points 128 b + n' + 16 j, j = 0..7  <=>  n = n' + 16 j
//                                       <=>  lane group k'' = j % 4, butterfly c = j / 4, register q = b.
//
// Give lane (n', k) the two blocks b = k + 4 c', c' in {0, 1}: what it needs from lane group k'' is that group's
// register (c, q = k + 4 c') -- for every pair (c, c') a 4 x 4 transpose between the lane-group index and the register
// index q % 4, i.e. two v_permlane32_swap + two v_permlane16_swap per four registers: 32 swaps per thread replace 16
// ds_write_b64 + 16 ds_read_b64 (>= 130 LDS cycles per wave, 2 k per CU) and the wait between them.
//
// The arithmetic is exactly fft2.h's fwd_mid_pass<1> + fwd_mid_pass<2> (the same butterflies and twiddles in the same
// order), so results are bit-identical to the two-phase form; the inverse is the mirror.  Device only.
#pragma once

#include "fft2.h"

#if defined(__HIPCC__) && !defined(MGX_HOST_EMU)

#if defined(__clang__)
#pragma clang fp contract(fast)
#endif

namespace mgx {

template <int LOG2N>
struct Fft2Lanes {
    using F = Fft2<LOG2N>;
    static constexpr bool AVAILABLE = F::P == 4 && F::R(1) == 8 && F::R(2) == 8 && F::S(2) == 16 && F::CNT(1) == 2 &&
                                      F::CNT(2) == 2 && F::WAVE_LOCAL;

    // rows of 16 lanes <-> four registers
    static __device__ __forceinline__ void swap32(float& a, float& b) {       // a's rows 2, 3 <-> b's rows 0, 1
        typedef unsigned u2_t __attribute__((ext_vector_type(2)));
        const u2_t r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
        a = __uint_as_float(r.x);
        b = __uint_as_float(r.y);
    }
    static __device__ __forceinline__ void swap16(float& a, float& b) {       // a's rows 1, 3 <-> b's rows 0, 2
        typedef unsigned u2_t __attribute__((ext_vector_type(2)));
        const u2_t r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
        a = __uint_as_float(r.x);
        b = __uint_as_float(r.y);
    }
    // register r of row k <- register k of row r
    static __device__ __forceinline__ void transpose4(float& r0, float& r1, float& r2, float& r3) {
        swap32(r0, r2);
        swap32(r1, r3);
        swap16(r0, r1);
        swap16(r2, r3);
    }
    static __device__ __forceinline__ void exchange(float2 (&a)[2][8]) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                transpose4(a[c][4 * h].x, a[c][4 * h + 1].x, a[c][4 * h + 2].x, a[c][4 * h + 3].x);
                transpose4(a[c][4 * h].y, a[c][4 * h + 1].y, a[c][4 * h + 2].y, a[c][4 * h + 3].y);
            }
        }
    }
    // pass-2 butterfly of lane (n' = tid & 15, k = (tid >> 4) & 3), c': block k + 4 c' of the wave's eight
    static __device__ __forceinline__ int butterfly2(int tid, int cc) {
        return (tid >> 6) * 128 + 16 * (((tid >> 4) & 3) + 4 * cc) + (tid & 15);
    }

    static __device__ __forceinline__ void fwd_mid_fused(int tid, float2* lds, const float2* table) {
        static_assert(AVAILABLE, "a four-pass plan with two radix-8 middle passes");
        constexpr int s1 = F::S(1), s2 = F::S(2);
        float2 a[2][8];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int u = F::template mid_butterfly<1>(tid, c);
            const int n = u % s1;
            const float2* p = lds + F::template base<1>(u);
            float2 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = p[F::template off<1>(j)];
            dft_regs<8, false>(v);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                float2 x = v[bitrev(q, 3)];
                if (q != 0) x = cmul(x, table[(q - 1) * s1 + n]);
                a[c][q] = x;
            }
        }
        exchange(a);
        const float2* t2 = table + F::MID_TABLE1;
        float2 w[7];
#pragma unroll
        for (int q = 1; q < 8; ++q) w[q - 1] = t2[(q - 1) * s2 + (tid & 15)];
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            float2* p = lds + F::template base<2>(butterfly2(tid, cc));
            float2 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = a[j >> 2][4 * cc + (j & 3)];
            dft_regs<8, false>(v);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                float2 x = v[bitrev(q, 3)];
                if (q != 0) x = cmul(x, w[q - 1]);
                p[F::template off<2>(q)] = x;
            }
        }
    }

    static __device__ __forceinline__ void inv_mid_fused(int tid, float2* lds, const float2* table) {
        static_assert(AVAILABLE, "a four-pass plan with two radix-8 middle passes");
        constexpr int s1 = F::S(1), s2 = F::S(2);
        float2 a[2][8];
        {
            const float2* t2 = table + F::MID_TABLE1;
            float2 w[7];
#pragma unroll
            for (int q = 1; q < 8; ++q) w[q - 1] = t2[(q - 1) * s2 + (tid & 15)];
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                const float2* p = lds + F::template base<2>(butterfly2(tid, cc));
                float2 v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    float2 x = p[F::template off<2>(q)];
                    if (q != 0) x = cmulc(x, w[q - 1]);
                    v[bitrev(q, 3)] = x;
                }
                dft_regs<8, true>(v);
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j >> 2][4 * cc + (j & 3)] = v[j];
            }
        }
        exchange(a);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int u = F::template mid_butterfly<1>(tid, c);
            const int n = u % s1;
            float2* p = lds + F::template base<1>(u);
            float2 v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                float2 x = a[c][q];
                if (q != 0) x = cmulc(x, table[(q - 1) * s1 + n]);
                v[bitrev(q, 3)] = x;
            }
            dft_regs<8, true>(v);
#pragma unroll
            for (int j = 0; j < 8; ++j) p[F::template off<1>(j)] = v[j];
        }
    }
};

}  // namespace mgx

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

#endif  // hipcc
