// Radix-8 butterflies of Fft2's middle passes on the matrix pipe (gfx950, v_mfma_f32_16x16x4_f32) -- an EXPERIMENT of
// round 6, kept beside its micro-benchmark as the record; the product does not use it.
//
// The question: gfx950 has an exact float32 MFMA (f32 in, f32 accumulate, bitwise an fmaf chain) at twice the rate of
// scalar f32 VALU instructions.  A radix-8 butterfly as a dense 16 x 16 real matrix costs 4.6 times the flops of the
// butterfly network -- 512 matrix-pipe cycles per 64 butterflies where the network costs ~85 VALU instructions -- so it
// could only pay BESIDE the VALU butterflies: a wave hands a share of a pass's butterflies to the matrix pipe and keeps
// the rest, if the two streams ran concurrently.
// The answer (tools/micro/mfma_pass.hip, profiles/r06_a_*): they do not.  A SIMD that executes a
// v_mfma_f32_16x16x4_f32 issues no VALU instruction meanwhile; eight VALU waves beside eight matrix waves take the sum
// of their times.  The code below is correct (1.6e-7 against the VALU passes) and slower in every mix.
//
// One tile = 16 butterflies (the 16 columns of the MFMA).  Lane l = 16 g + n (g = 0..3) of the wave:
//   * loads points g and g + 4 of butterfly n (two ds_read_b64): x.re / x.im of point g + 4h are the B operands of
//     MFMA 2h / 2h + 1 (B[k = g][col = n]);
//   * holds the matrix in four registers: A operand of MFMA m = 2h + p is W[row = l & 15][col = (point g + 4h, part p)],
//     row r <-> output (q = 2 (r >> 2) + ((r >> 1) & 1), part r & 1);
//   * receives D[row = 4 g + v][col = n], v = 0..3: the complex outputs q = 2g and 2g + 1 of butterfly n, which it
//     twiddles and stores (two ds_write_b64).
// Forward (decimation in frequency, fft2.h): X_q = sum_j x_j w8^(jq), stored as X_q w_M^(qn) at point q.
// Inverse (its mirror): y_j = sum_q (x_q conj(w_M^(qn))) w8^(-jq), stored at point j.  Unnormalised, as fft2.h.
//
// The layouts, ownership (a wave's butterflies cover the points it owns: Fft2::WAVE_LOCAL) and twiddle tables are
// fft2.h's; a pass may be split between fwd_mid_pass-style VALU butterflies and these tiles in any proportion.
#pragma once

#include "fft2.h"          // (matchering_amd/csrc, on the include path of the micro-benchmark)

#if defined(__HIPCC__) && !defined(MGX_HOST_EMU)

#if defined(__clang__)
#pragma clang fp contract(fast)
#endif

namespace mgx {

typedef float mfma_f4 __attribute__((ext_vector_type(4)));

template <int LOG2N>
struct Fft2Mfma {
    using F = Fft2<LOG2N>;

    // the wave's copy of the 16 x 16 real DFT-8 matrix: a[m] = A operand of MFMA m
    struct Matrix {
        float a[4];
    };
    template <bool INV>
    static __device__ __forceinline__ void load_matrix(int lane, Matrix& w) {
        const float h = 0.70710678118654752440f;
        const int r = lane & 15, g = (lane >> 4) & 3;          // (lane may be a thread id)
        const int q = 2 * (r >> 2) + ((r >> 1) & 1), part_out = r & 1;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int j = g + 4 * (m >> 1), part_in = m & 1;
            const int k = (j * q) & 7;                       // angle 2 pi k / 8
            // cos, sin of 2 pi k / 8 (exact table values)
            const float c = (k == 0) ? 1.f : (k == 4) ? -1.f : (k == 2 || k == 6) ? 0.f : (k == 1 || k == 7) ? h : -h;
            const float s = (k == 2) ? 1.f : (k == 6) ? -1.f : (k == 0 || k == 4) ? 0.f : (k == 1 || k == 3) ? h : -h;
            // forward: (xr + i xi)(c - i s) = (xr c + xi s) + i (xi c - xr s); inverse: s -> -s
            const float sg = INV ? -s : s;
            float v;
            if (part_out == 0) v = part_in == 0 ? c : sg;
            else v = part_in == 0 ? -sg : c;
            w.a[m] = v;
        }
    }

    // ---- one pass's tiles [T0, T0 + NT) of the wave (tile t = the wave's butterflies 16 t .. 16 t + 15) -------------
    // Split into load / compute / store so that a caller can lay another stream of work between them.
    template <int NT>
    struct Tiles {
        float2 x[NT][2];       // loaded points g, g + 4 (inverse: already twiddled)
        mfma_f4 d[NT];
    };

    template <int PASS>
    static __device__ __forceinline__ int tile_butterfly(int tid, int t) {      // butterfly of this lane's column in tile t
        return (tid >> 6) * (F::CNT(PASS) * 64) + 16 * t + (tid & 15);
    }
    // padded distance of point e (run time) from point 0 of a butterfly: off<PASS>(e) is linear in e for middle passes
    template <int PASS>
    static __device__ __forceinline__ int off_rt(int e) {
        static_assert(F::S(PASS) % F::RL == 0, "linear padded stride");
        return e * F::template off<PASS>(1);
    }

    // A tile further on lies a constant number of LDS elements further on (16 | S(PASS) in a middle pass), and so do
    // its twiddles: every address of a pass is one per-lane value plus literals.
    template <int PASS>
    static constexpr int tile_step() {
        constexpr int i = F::S(PASS) > 16 ? 16 : F::M(PASS);      // the next 16 columns of a block, or the next block
        return F::PADDED ? i + ((i >> F::LRL) << 1) : i;
    }
    template <int PASS>
    static constexpr int tile_twiddle_step() { return F::S(PASS) > 16 ? 16 : 0; }     // n advances by 16, or stays
    // table index of w^(q n) for this lane's column, q run time; q = 0 has no row: entry [0][0] = w^0 = 1 serves
    template <int PASS>
    static __device__ __forceinline__ int twiddle_index(int q, int n) {
        return q == 0 ? 0 : (q - 1) * F::S(PASS) + n;
    }

    template <int PASS, bool INV, int T0, int NT>
    static __device__ __forceinline__ void load(int tid, const float2* lds, const float2* table, Tiles<NT>& ts) {
        static_assert(F::S(PASS) % 16 == 0 && (F::S(PASS) == 16 || F::S(PASS) % 64 == 0), "tile geometry");
        constexpr int s = F::S(PASS);
        const int g = (tid >> 4) & 3;
        const int u0 = tile_butterfly<PASS>(tid, 0);
        const float2* p0 = lds + F::template base<PASS>(u0) + off_rt<PASS>(g);
        const int n0 = u0 % s;
        const float2* tw0 = table + twiddle_index<PASS>(g, n0);
        const float2* tw1 = table + twiddle_index<PASS>(g + 4, n0);
        float2 w0, w1;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float2* p = p0 + (T0 + t) * tile_step<PASS>();
            ts.x[t][0] = p[0];
            ts.x[t][1] = p[F::template off<PASS>(4)];
            if (INV) {
                // inputs q = g and g + 4, each times conj(w^(q n))
                if (t == 0 || tile_twiddle_step<PASS>() != 0) {
                    w0 = tw0[g == 0 ? 0 : (T0 + t) * tile_twiddle_step<PASS>()];
                    w1 = tw1[(T0 + t) * tile_twiddle_step<PASS>()];
                }
                ts.x[t][0] = cmulc(ts.x[t][0], w0);
                ts.x[t][1] = cmulc(ts.x[t][1], w1);
            }
        }
    }
    template <int NT>
    static __device__ __forceinline__ void multiply(const Matrix& w, Tiles<NT>& ts) {
        // MFMA m of every tile before MFMA m + 1 of any: consecutive instructions never depend on each other
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const mfma_f4 zero = {0.f, 0.f, 0.f, 0.f};
            ts.d[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.a[0], ts.x[t][0].x, zero, 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) ts.d[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.a[1], ts.x[t][0].y, ts.d[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; ++t) ts.d[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.a[2], ts.x[t][1].x, ts.d[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; ++t) ts.d[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.a[3], ts.x[t][1].y, ts.d[t], 0, 0, 0);
    }
    template <int PASS, bool INV, int T0, int NT>
    static __device__ __forceinline__ void store(int tid, float2* lds, const float2* table, const Tiles<NT>& ts) {
        constexpr int s = F::S(PASS);
        const int g = (tid >> 4) & 3;
        const int u0 = tile_butterfly<PASS>(tid, 0);
        float2* p0 = lds + F::template base<PASS>(u0) + off_rt<PASS>(2 * g);
        const int n0 = u0 % s;
        const float2* tw0 = table + twiddle_index<PASS>(2 * g, n0);
        const float2* tw1 = table + twiddle_index<PASS>(2 * g + 1, n0);
        float2 w0, w1;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float2* p = p0 + (T0 + t) * tile_step<PASS>();
            float2 y0 = make_float2(ts.d[t][0], ts.d[t][1]), y1 = make_float2(ts.d[t][2], ts.d[t][3]);
            if (!INV) {
                // outputs q = 2g and 2g + 1, each times w^(q n)
                if (t == 0 || tile_twiddle_step<PASS>() != 0) {
                    w0 = tw0[g == 0 ? 0 : (T0 + t) * tile_twiddle_step<PASS>()];
                    w1 = tw1[(T0 + t) * tile_twiddle_step<PASS>()];
                }
                y0 = cmul(y0, w0);
                y1 = cmul(y1, w1);
            }
            p[0] = y0;
            p[F::template off<PASS>(1)] = y1;
        }
    }

    // the whole of a pass's tiles [T0, T0 + NT) in one go
    template <int PASS, bool INV, int T0, int NT>
    static __device__ __forceinline__ void pass_tiles(int tid, float2* lds, const float2* table, const Matrix& w) {
        Tiles<NT> ts;
        load<PASS, INV, T0, NT>(tid, lds, table, ts);
        multiply<NT>(w, ts);
        store<PASS, INV, T0, NT>(tid, lds, table, ts);
    }

    // ---- the VALU butterflies c in [C0, C1) of a middle pass (fft2.h's fwd_mid_pass / inv_mid_pass, restricted) ------
    template <int PASS, bool INV, int C0, int C1>
    static __device__ __forceinline__ void pass_valu(int tid, float2* lds, const float2* table) {
        constexpr int r = F::R(PASS), bits = F::lr(PASS), s = F::S(PASS);
        constexpr bool same = s <= 64;
        float2 w[r - 1];
#pragma unroll
        for (int c = C0; c < C1; ++c) {
            const int u = F::template mid_butterfly<PASS>(tid, c);
            if (c == C0 || !same) {
                const int n = u % s;
#pragma unroll
                for (int q = 1; q < r; ++q) w[q - 1] = table[(q - 1) * s + n];
            }
            float2* p = lds + F::template base<PASS>(u);
            float2 v[r];
            if (!INV) {
#pragma unroll
                for (int j = 0; j < r; ++j) v[j] = p[F::template off<PASS>(j)];
                dft_regs<r, false>(v);
#pragma unroll
                for (int q = 0; q < r; ++q) {
                    float2 x = v[bitrev(q, bits)];
                    if (q != 0) x = cmul(x, w[q - 1]);
                    p[F::template off<PASS>(q)] = x;
                }
            } else {
#pragma unroll
                for (int q = 0; q < r; ++q) {
                    float2 x = p[F::template off<PASS>(q)];
                    if (q != 0) x = cmulc(x, w[q - 1]);
                    v[bitrev(q, bits)] = x;
                }
                dft_regs<r, true>(v);
#pragma unroll
                for (int j = 0; j < r; ++j) p[F::template off<PASS>(j)] = v[j];
            }
        }
    }

    // ---- a middle pass shared between the pipes: the wave's first 64 VC butterflies per lane on the VALU, the tiles
    // behind them on the matrix pipe.  VC = 0 .. CNT(PASS); tiles 4 VC .. 4 CNT - 1.
    template <int PASS, bool INV, int VC>
    static __device__ __forceinline__ void pass_shared(int tid, float2* lds, const float2* table, const Matrix& w) {
        static_assert(F::R(PASS) == 8, "radix-8 middle passes only");
        constexpr int CNT = F::CNT(PASS);
        constexpr int NT = 4 * (CNT - VC);
        if constexpr (NT == 0) {
            pass_valu<PASS, INV, 0, CNT>(tid, lds, table);
        } else if constexpr (VC == 0) {
            // (four tiles at a time: eight would hold 64 registers of operands and results)
#pragma unroll
            for (int h = 0; h < CNT; ++h) {
                if (h == 0) pass_tiles<PASS, INV, 0, 4>(tid, lds, table, w);
                else pass_tiles<PASS, INV, 4, 4>(tid, lds, table, w);
            }
            static_assert(CNT <= 2, "two groups of four tiles");
        } else {
            // loads of the tiles first (the compiler cannot tell that the two streams touch different points, so it
            // keeps LDS reads behind earlier LDS writes: everything the matrix stream reads is read before the VALU
            // stream writes), then both streams' arithmetic -- free to interleave --, then the tiles' stores
            Tiles<NT> ts;
            load<PASS, INV, 4 * VC, NT>(tid, lds, table, ts);
            multiply<NT>(w, ts);
            pass_valu<PASS, INV, 0, VC>(tid, lds, table);
            store<PASS, INV, 4 * VC, NT>(tid, lds, table, ts);
        }
    }
};

}  // namespace mgx

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

#endif  // hipcc
