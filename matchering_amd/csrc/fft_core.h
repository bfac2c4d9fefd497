// LDS-resident complex FFT building blocks for gfx950 (wave64), float32.
//
// A transform of N = 2^LOG2N points lives in one workgroup's LDS as N float2
// (N <= 16384 -> 128 KiB of the CU's 160 KiB).  It is computed in P passes of
// register radix-R butterflies (R in {8,16,32}); between passes the data makes one
// round trip through LDS.  The forward transform is decimation-in-frequency
// (natural order in, digit-reversed order out), the inverse is the exact mirror
// (decimation-in-time, digit-reversed in, natural out), so nothing is ever
// re-ordered: spectra are consumed in "position order" and every table that is
// multiplied against a spectrum (FIR spectra) is produced by the same forward
// transform and therefore already sits in position order.
//
// Forward pass p works on sub-transforms of length M_p = N / (R_0 ... R_{p-1})
// with stride S_p = M_p / R_p:
//     x_j = s[b*M_p + n + j*S_p],  X_q = sum_j x_j w_R^(jq),
//     s[b*M_p + n + q*S_p] = X_q * w_{M_p}^(q*n)
// After the last pass position q_0*S_0 + q_1*S_1 + ... holds X[q_0 + R_0*q_1 + ...].
//
// Everything here is a per-thread function over explicit state (mgx_hd.h) so the
// same code runs under the host emulation used by the CPU tests.
#pragma once

#include "mgx_hd.h"

namespace mgx {

// cos/sin of 2*pi*k/32, k = 0..8 (first octant+), everything else by symmetry
MGX_HD float2 unit32(int k) {
    // returns (cos(2 pi k/32), sin(2 pi k/32)) for k in [0, 32)
    constexpr double C[9] = {1.0,
                             0.98078528040323044913,
                             0.92387953251128675613,
                             0.83146961230254523708,
                             0.70710678118654752440,
                             0.55557023301960222474,
                             0.38268343236508977173,
                             0.19509032201612826785,
                             0.0};
    k &= 31;
    const int quad = k >> 3, r = k & 7;
    // angle = quad*90deg + r*11.25deg
    const double c = C[r], s = C[8 - r];
    switch (quad) {
        case 0: return make_float2((float)c, (float)s);
        case 1: return make_float2((float)-s, (float)c);
        case 2: return make_float2((float)-c, (float)-s);
        default: return make_float2((float)s, (float)-c);
    }
}

// t * exp(-/+ 2 pi i k / m), sign - for forward (INV = false); k, m are compile-time
// constants after unrolling so the special cases fold away.
template <bool INV>
MGX_HD float2 rot(float2 t, int k, int m) {
    if (k == 0) return t;
    if (4 * k == m) return INV ? cmul_i(t) : cmul_mi(t);
    const float2 u = unit32(k * (32 / m));
    return INV ? cmul(t, u) : cmulc(t, u);
}

// In-register radix-R DFT.  Forward: natural in, X[q] left at v[bitrev(q)].
// Inverse: expects Y[q] at v[bitrev(q)], leaves x[j] at v[j].  Unnormalised.
template <int R, bool INV>
MGX_HD void dft_regs(float2 (&v)[R]) {
    if (!INV) {
        MGX_UNROLL
        for (int span = R / 2; span >= 1; span >>= 1) {
            MGX_UNROLL
            for (int g = 0; g < R; g += 2 * span) {
                MGX_UNROLL
                for (int k = 0; k < span; ++k) {
                    const float2 a = v[g + k], b = v[g + k + span];
                    v[g + k] = cadd(a, b);
                    v[g + k + span] = rot<false>(csub(a, b), k, 2 * span);
                }
            }
        }
    } else {
        MGX_UNROLL
        for (int span = 1; span <= R / 2; span <<= 1) {
            MGX_UNROLL
            for (int g = 0; g < R; g += 2 * span) {
                MGX_UNROLL
                for (int k = 0; k < span; ++k) {
                    const float2 a = v[g + k];
                    const float2 t = rot<true>(v[g + k + span], k, 2 * span);
                    v[g + k] = cadd(a, t);
                    v[g + k + span] = csub(a, t);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Plans
// ---------------------------------------------------------------------------
template <int LOG2N>
struct FftPlan;
#define MGX_PLAN2(L, A, B_)                                               \
    template <>                                                           \
    struct FftPlan<L> {                                                   \
        static constexpr int P = 2;                                       \
        static constexpr int LR[3] = {A, B_, 0};                          \
    };
#define MGX_PLAN3(L, A, B_, C_)                                           \
    template <>                                                           \
    struct FftPlan<L> {                                                   \
        static constexpr int P = 3;                                       \
        static constexpr int LR[3] = {A, B_, C_};                         \
    };
MGX_PLAN2(6, 3, 3)
MGX_PLAN2(7, 4, 3)
MGX_PLAN2(8, 4, 4)
MGX_PLAN2(9, 5, 4)
MGX_PLAN2(10, 5, 5)
MGX_PLAN3(11, 4, 4, 3)
MGX_PLAN3(12, 4, 4, 4)
MGX_PLAN3(13, 5, 4, 4)
MGX_PLAN3(14, 5, 5, 4)
#undef MGX_PLAN2
#undef MGX_PLAN3

template <int LOG2N>
struct Fft {
    using Plan = FftPlan<LOG2N>;
    static constexpr int N = 1 << LOG2N;
    static constexpr int P = Plan::P;
    static constexpr int T = (N / 32) < 64 ? 64 : (N / 32);   // threads per workgroup
    static constexpr int LAST = P - 1;

    static constexpr int lr(int p) { return Plan::LR[p]; }
    static constexpr int R(int p) { return 1 << Plan::LR[p]; }
    static constexpr int logM(int p) {           // log2 of sub-transform length entering pass p
        int l = LOG2N;
        for (int i = 0; i < p; ++i) l -= Plan::LR[i];
        return l;
    }
    static constexpr int M(int p) { return 1 << logM(p); }
    static constexpr int S(int p) { return M(p) >> Plan::LR[p]; }
    static constexpr int NB(int p) { return N >> Plan::LR[p]; }          // butterflies in pass p
    static constexpr int CNT(int p) { return NB(p) / T > 0 ? NB(p) / T : 1; }   // per thread

    // LDS layout: one float2 of padding after every 16 keeps the strided passes
    // (stride 16/256 elements between lanes) off a single bank.
    static constexpr int PAD_SHIFT = 4;
    static constexpr int LDS_ELEMS = N + (N >> PAD_SHIFT);
    static MGX_HD int pad(int i) { return i + (i >> PAD_SHIFT); }

    // position of butterfly u of pass p, element index e (input j or output q)
    template <int PASS>
    static MGX_HD int pos(int u, int e) {
        constexpr int s = S(PASS), m = M(PASS);
        const int blk = u / s, n = u % s;
        return blk * m + n + e * s;
    }

    // twiddle exponent (in units of 2*pi/N) for output q of butterfly u in pass p
    template <int PASS>
    static MGX_HD int tw_index(int u, int q) {
        constexpr int s = S(PASS), m = M(PASS);
        return ((u % s) * q) * (N / m);       // < N because n*q < m
    }

    // ---- forward pass from registers holding natural-order inputs ------------
    // v[j] in, twiddled outputs written to LDS.  tw = table of exp(-2 pi i k/N).
    template <int PASS>
    static MGX_HD void fwd_store(float2 (&v)[R(PASS)], int u, float2* lds, const float2* tw) {
        constexpr int r = R(PASS), bits = lr(PASS);
        dft_regs<r, false>(v);
        MGX_UNROLL
        for (int q = 0; q < r; ++q) {
            float2 x = v[bitrev(q, bits)];
            if (PASS != LAST && q != 0) x = cmul(x, tw[tw_index<PASS>(u, q)]);
            lds[pad(pos<PASS>(u, q))] = x;
        }
    }
    template <int PASS>
    static MGX_HD void load_natural(float2 (&v)[R(PASS)], int u, const float2* lds) {
        MGX_UNROLL
        for (int j = 0; j < R(PASS); ++j) v[j] = lds[pad(pos<PASS>(u, j))];
    }
    // ---- inverse pass: load position order, un-twiddle, inverse butterfly ----
    template <int PASS>
    static MGX_HD void inv_load(float2 (&v)[R(PASS)], int u, const float2* lds, const float2* tw) {
        constexpr int r = R(PASS), bits = lr(PASS);
        MGX_UNROLL
        for (int q = 0; q < r; ++q) {
            float2 x = lds[pad(pos<PASS>(u, q))];
            if (PASS != LAST && q != 0) x = cmulc(x, tw[tw_index<PASS>(u, q)]);
            v[bitrev(q, bits)] = x;
        }
        dft_regs<r, true>(v);
    }
    template <int PASS>
    static MGX_HD void store_natural(const float2 (&v)[R(PASS)], int u, float2* lds) {
        MGX_UNROLL
        for (int j = 0; j < R(PASS); ++j) lds[pad(pos<PASS>(u, j))] = v[j];
    }

    // ---- whole middle passes (LDS -> LDS), one call per pass, barrier outside --
    template <int PASS>
    static MGX_HD void fwd_pass_lds(int tid, float2* lds, const float2* tw) {
        MGX_UNROLL
        for (int i = 0; i < CNT(PASS); ++i) {
            const int u = tid + i * T;
            if (u < NB(PASS)) {
                float2 v[R(PASS)];
                load_natural<PASS>(v, u, lds);
                fwd_store<PASS>(v, u, lds, tw);
            }
        }
    }
    template <int PASS>
    static MGX_HD void inv_pass_lds(int tid, float2* lds, const float2* tw) {
        MGX_UNROLL
        for (int i = 0; i < CNT(PASS); ++i) {
            const int u = tid + i * T;
            if (u < NB(PASS)) {
                float2 v[R(PASS)];
                inv_load<PASS>(v, u, lds, tw);
                store_natural<PASS>(v, u, lds);
            }
        }
    }

    // ---- frequency index <-> position ------------------------------------------
    // position holding X[k]
    static MGX_HD int position_of(int k) {
        int posn = 0;
        MGX_UNROLL
        for (int p = 0; p < P; ++p) {
            const int q = k & (R(p) - 1);
            k >>= lr(p);
            posn += q * S(p);
        }
        return posn;
    }
    // frequency index stored at a position
    static MGX_HD int frequency_at(int posn) {
        int k = 0, shift = 0;
        MGX_UNROLL
        for (int p = 0; p < P; ++p) {
            const int q = (posn / S(p)) & (R(p) - 1);
            k |= q << shift;
            shift += lr(p);
        }
        return k;
    }
};

}  // namespace mgx
