// LDS-resident complex FFT building blocks for gfx950 (wave64), float32 -- second generation.
//
// A transform of N = 2^LOG2N points lives in one workgroup's LDS as N float2 plus padding.
// It is computed in P <= 4 passes of register radix-R butterflies, R in {8,16,32}; between
// passes the data makes one round trip through LDS.  The forward transform is
// decimation-in-frequency (natural order in, "position order" out), the inverse is its exact
// mirror (decimation-in-time, position order in, natural order out), so nothing is ever
// re-ordered: whatever is multiplied against a spectrum is tabulated in position order.
//
// Forward pass p works on sub-transforms of length M_p = N / (R_0 ... R_{p-1}) with stride
// S_p = M_p / R_p.  Butterfly u = (blk, n) = (u / S_p, u % S_p):
//     x_j = s[blk*M_p + n + j*S_p],   X_q = sum_j x_j w_R^(jq),
//     s[blk*M_p + n + q*S_p] = X_q * w_{M_p}^(q*n)
// After the last pass position q_0*S_0 + q_1*S_1 + ... holds X[q_0 + R_0*q_1 + R_0*R_1*q_2].
//
// Design rules (tools/lds_conflicts.py checks the LDS ones against the gfx950 bank model):
//  * The last pass has S = 1: a thread owns ONE contiguous row of RL float2 (RL = 32, or 16 for
//    N = 4096) which it moves with 16-byte LDS accesses; T = N/RL threads (>= 64).  Thread t owns
//    butterflies u = t + c*T, c < CNT_p, in pass 0; in the middle passes a WAVE owns a contiguous run
//    of butterflies (mid_butterfly) -- the points its rows cover -- so that only pass 0 exchanges data
//    between waves.  Consecutive lanes touch consecutive float2 either way (ds_read_b64 / ds_write_b64).
//  * Two float2 of padding follow every row: pad(i) = i + 2*(i / RL).  A row is then 272 (144)
//    bytes, ds_read_b128 / ds_write_b128 of rows are conflict-free and so are all strided
//    writes; only the strided reads of the 4096-point plan pay a 2-way conflict.
//  * Pass-0 twiddles w_N^(q*t) depend on the thread only: R_0-1 float2 held in VGPRs for the
//    lifetime of a persistent workgroup; butterfly c > 0 needs w_N^(q*(t+c*T)) = that times
//    w_32^(q*c), a literal.  Pass-1 twiddles depend on (q, n < S_1) only: a small LDS table.
//  * No bounds checks, no global memory traffic in here: callers hand in registers.
//
// Everything is a per-thread function over explicit state (mgx_hd.h), so the same code runs
// under the host emulation used by the CPU tests.
#pragma once

#include "mgx_hd.h"

// The library is built with -ffp-contract=off (build.py): float64 sums and the host emulation are to
// round like the plain expressions read.  The float32 butterflies of this file are the exception: a
// multiply feeding an add may fuse (a sqrt(1/2) rotation into the butterfly that follows it: 4 % fewer
// VALU instructions in k_conv, 160 -> 154.5 us, profiles/r03_a_fp_contract.txt), results stay within
// 1e-7 of the unfused ones.  Scoped to this file and conv2_kernel.h.
#if defined(__clang__)
#pragma clang fp contract(fast)
#endif

namespace mgx {

// (cos, sin)(2 pi k / 32)
MGX_HD float2 unit32(int k) {
    constexpr float C[9] = {1.0f,
                            0.98078528040323044913f,
                            0.92387953251128675613f,
                            0.83146961230254523708f,
                            0.70710678118654752440f,
                            0.55557023301960222474f,
                            0.38268343236508977173f,
                            0.19509032201612826785f,
                            0.0f};
    k &= 31;
    const int quad = k >> 3, r = k & 7;
    const float c = C[r], s = C[8 - r];
    switch (quad) {
        case 0: return make_float2(c, s);
        case 1: return make_float2(-s, c);
        case 2: return make_float2(-c, -s);
        default: return make_float2(s, -c);
    }
}

// t * exp(-/+ 2 pi i k / m), sign - for forward (INV = false); k, m are compile-time constants
// after unrolling so the special cases fold away.
template <bool INV>
MGX_HD float2 rot(float2 t, int k, int m) {
    if (k == 0) return t;
    if (4 * k == m) return INV ? cmul_i(t) : cmul_mi(t);
    if (8 * k == m) {          // (1 -/+ i)/sqrt2
        const float h = 0.70710678118654752440f;
        return INV ? make_float2(h * (t.x - t.y), h * (t.x + t.y)) : make_float2(h * (t.x + t.y), h * (t.y - t.x));
    }
    if (8 * k == 3 * m) {      // (-1 -/+ i)/sqrt2
        const float h = 0.70710678118654752440f;
        return INV ? make_float2(-h * (t.x + t.y), h * (t.x - t.y)) : make_float2(h * (t.y - t.x), -h * (t.x + t.y));
    }
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
// Plans: log2 of the radix of each pass; the last entry is the row pass
// ---------------------------------------------------------------------------
template <int LOG2N>
struct Fft2Plan;
#define MGX_PLAN(L, NP, A, B_, C_, D_)                      \
    template <>                                             \
    struct Fft2Plan<L> {                                    \
        static constexpr int P = NP;                        \
        static constexpr int LR[4] = {A, B_, C_, D_};       \
    };
MGX_PLAN(6, 2, 3, 3, 0, 0)
MGX_PLAN(7, 2, 3, 4, 0, 0)
MGX_PLAN(8, 2, 4, 4, 0, 0)
MGX_PLAN(9, 2, 4, 5, 0, 0)
MGX_PLAN(10, 2, 5, 5, 0, 0)
MGX_PLAN(11, 3, 3, 3, 5, 0)
MGX_PLAN(12, 3, 4, 4, 4, 0)
#ifdef MGX_FFT13_FOUR_PASSES          // A/B build: 8192 = 8 * 8 * 8 * 16 on 512 threads (16 points per thread)
MGX_PLAN(13, 4, 3, 3, 3, 4)
#else
MGX_PLAN(13, 3, 4, 4, 5, 0)
#endif
// 16384 = 16 * 8 * 8 * 16 on 1024 threads (16 points per thread and pass: 128 VGPRs, four waves per SIMD) since round 4.
// Until then 16 * 32 * 32 on 512 threads (32 points, 256 VGPRs, two waves per SIMD, 388 - 532 B of scratch in the
// convolution kernels): one more round trip through LDS buys half the registers -- the plain 8192-tap convolution
// went from 200 to 126 us, the partitioned 16384-tap one from 544 to 495, the analysis from 154 to 150
// (profiles/r04_l_fft14_four_passes.txt).  -DMGX_FFT14_THREE_PASSES builds the old plan for an A/B.
#ifdef MGX_FFT14_THREE_PASSES
MGX_PLAN(14, 3, 4, 5, 5, 0)
#else
MGX_PLAN(14, 4, 4, 3, 3, 4)
#endif
#undef MGX_PLAN

template <int LOG2N>
struct Fft2 {
    using Plan = Fft2Plan<LOG2N>;
    static constexpr int N = 1 << LOG2N;
    static constexpr int P = Plan::P;
    static constexpr int LAST = P - 1;
    static constexpr int L_ROWS = N >> Plan::LR[P - 1];
    static constexpr int T = L_ROWS < 64 ? 64 : L_ROWS;        // threads per workgroup = rows of the last pass

    static constexpr int lr(int p) { return Plan::LR[p]; }
    static constexpr int R(int p) { return 1 << Plan::LR[p]; }
    static constexpr int logM(int p) {
        int l = LOG2N;
        for (int i = 0; i < p; ++i) l -= Plan::LR[i];
        return l;
    }
    static constexpr int M(int p) { return 1 << logM(p); }
    static constexpr int S(int p) { return M(p) >> Plan::LR[p]; }
    static constexpr int NB(int p) { return N >> Plan::LR[p]; }                  // butterflies in pass p
    static constexpr int CNT(int p) { return NB(p) / T > 0 ? NB(p) / T : 1; }   // per thread
    static constexpr bool partial(int p) { return NB(p) < T; }                   // some threads idle

    static constexpr int R0 = 1 << Plan::LR[0];
    static constexpr int RL = 1 << Plan::LR[P - 1];
    static constexpr int L = N / RL;                                             // rows of the last pass
    static_assert(CNT(0) == 1 || T == L, "pass-0 twiddle step between a thread's butterflies is w_RL");
    static_assert(S(LAST) == 1, "last pass works on contiguous rows");

    // LDS layout: two float2 of padding after every 32 (keeps 16-byte alignment of even indices).
    // Tiny transforms (N < 512, only met in tests) are not padded.  In a padded plan every stride
    // S_p of a strided pass is a multiple of 32 and a row of the last pass is exactly 32 long, so
    // the padded index splits into a per-butterfly base (run time) plus a per-element constant
    // (compile time, folded into the DS instruction's immediate offset):
    //     pad(base + e*S) = pad(base) + e*(S + S/16),      pad(row*32 + e) = row*34 + e
    static constexpr bool PADDED = N >= 512;
    static constexpr int LRL = Plan::LR[P - 1];               // log2 of the row length
    static MGX_HD int pad(int i) { return PADDED ? i + ((i >> LRL) << 1) : i; }
    static constexpr int LDS_ELEMS = (PADDED ? N + ((N >> LRL) << 1) : N) + 2;

    template <int PASS>
    static MGX_HD int base(int u) {              // padded index of element 0 of butterfly u
        constexpr int s = S(PASS), m = M(PASS);
        return pad((u / s) * m + (u % s));
    }
    template <int PASS>
    static constexpr int off(int e) {            // padded distance of element e from element 0
        return PADDED ? e * S(PASS) + (((e * S(PASS)) >> LRL) << 1) : e * S(PASS);
    }
    static_assert(!PADDED || P == 1 || S(0) % RL == 0, "padded strides are multiples of the row length");
    static_assert(!PADDED || P < 3 || S(1) % RL == 0, "padded strides are multiples of the row length");
    static_assert(P < 4 || !PADDED || S(2) % RL == 0, "the same for the second middle pass");
    static_assert(P <= 4, "at most two middle passes");

    // ---- who works on what after pass 0 -----------------------------------------------------------
    // Pass 0 leaves R0 independent sub-transforms of M(1) = N/R0 contiguous points.  The W = T/64 wavefronts of
    // the workgroup share them out in order -- wave w owns points [w N/W, (w+1) N/W) -- and every later pass
    // (the middle passes and the row pass, in both directions) deals its butterflies to the waves in the same
    // contiguous runs.  A wave then reads only what it wrote itself between pass 0 and the inverse of pass 0,
    // so the passes in between need no workgroup barrier, only program order within the wave (pass_sync()
    // in mgx_kernels.h): one barrier per transform and direction instead of one per pass.
    static constexpr int W = T / 64;
    static constexpr bool WAVE_LOCAL = P >= 3 && R0 % W == 0;
    template <int PASS>
    static MGX_HD int mid_butterfly(int tid, int c) {       // butterfly c of the thread in a middle pass
        static_assert(P < 3 || !partial(PASS), "every thread owns a butterfly of a middle pass");
        return (tid >> 6) * (CNT(PASS) * 64) + (tid & 63) + 64 * c;
    }
    static_assert(P < 3 || (CNT(1) * 64) % S(1) == 0 || S(1) % (CNT(1) * 64) == 0, "a wave's run starts on a twiddle period");

    // twiddle exponent (units of 2*pi/N) for output q of butterfly u in pass p: (u % S)*q*(N/M)
    template <int PASS>
    static MGX_HD int tw_index(int u, int q) {
        return ((u % S(PASS)) * q) * (N / M(PASS));
    }

    // ---- twiddles ----------------------------------------------------------------------
    // pass 0: w_N^(q*t), q = 1..R0-1.  Only the log2(R0) "base" powers w^(t*2^i) live in registers
    // for the life of a workgroup (exact table values); a phase that needs the full set rebuilds the
    // other R0-1-log2(R0) by products of depth <= log2(R0)-1 (11 complex multiplies for radix 16
    // against a ~230-instruction butterfly), which keeps ~22 VGPRs free.
    static constexpr int LB0 = Plan::LR[0];
    struct Tw0 {
        float2 b[LB0];
    };
    struct Tw0Full {
        float2 w[R0 - 1];
    };
    static MGX_HD void load_tw0(int tid, const float2* tw, Tw0& t) {
        MGX_UNROLL
        for (int i = 0; i < LB0; ++i) t.b[i] = tw[tw_index<0>(tid % S(0), 1 << i) & (N - 1)];
    }
    static MGX_HD void expand_tw0(const Tw0& t, Tw0Full& f) {
        MGX_UNROLL
        for (int q = 1; q < R0; ++q) {
            int hb = 1;
            while (hb * 2 <= q) hb *= 2;                       // highest set bit of q
            f.w[q - 1] = q == hb ? t.b[ilog2(hb)] : cmul(t.b[ilog2(hb)], f.w[q - hb - 1]);
        }
    }
    // middle passes (P >= 3): (R_p - 1) * S_p entries in LDS per pass, [q-1][n]; the second pass's table (P == 4)
    // follows the first
    static constexpr int MID = 1, MID2 = P == 4 ? 2 : 1;
    static constexpr int MID_TABLE1 = P >= 3 ? (R(1) - 1) * S(1) : 0;
    static constexpr int MID_TABLE = MID_TABLE1 + (P == 4 ? (R(2) - 1) * S(2) : 0);
    static MGX_HD void fill_mid_table(int tid, const float2* tw, float2* table) {
        if (P >= 3) {
            for (int e = tid; e < MID_TABLE1; e += T) {
                const int q = e / S(MID) + 1, n = e % S(MID);
                table[e] = tw[tw_index<MID>(n, q)];
            }
        }
        if (P == 4) {
            for (int e = tid; e < MID_TABLE - MID_TABLE1; e += T) {
                const int q = e / S(MID2) + 1, n = e % S(MID2);
                table[MID_TABLE1 + e] = tw[tw_index<MID2>(n, q)];
            }
        }
    }

    // ---- pass 0, forward: registers (natural order v[j] = x[u + j*S0]) -> LDS ----------------
    // c = which of the thread's CNT(0) butterflies (u = tid + c*T)
    static MGX_HD void fwd0_store(float2 (&v)[R0], int tid, int c, const Tw0Full& t, float2* lds) {
        constexpr int bits = lr(0);
        float2* p = lds + base<0>(tid + c * T);
        dft_regs<R0, false>(v);
        MGX_UNROLL
        for (int q = 0; q < R0; ++q) {
            float2 x = v[bitrev(q, bits)];
            if (P > 1 && q != 0) {
                float2 w = t.w[q - 1];
                if (c != 0) w = cmulc(w, unit32(q * c * (32 / RL)));   // w_N^(q*c*T) = exp(-2 pi i q c/RL)
                x = cmul(x, w);
            }
            p[off<0>(q)] = x;
        }
    }
    // ---- pass 0, inverse: LDS -> registers (natural order v[j] = y[u + j*S0]) ----------------
    static MGX_HD void inv0_load(float2 (&v)[R0], int tid, int c, const Tw0Full& t, const float2* lds) {
        constexpr int bits = lr(0);
        const float2* p = lds + base<0>(tid + c * T);
        MGX_UNROLL
        for (int q = 0; q < R0; ++q) {
            float2 x = p[off<0>(q)];
            if (P > 1 && q != 0) {
                float2 w = t.w[q - 1];
                if (c != 0) w = cmulc(w, unit32(q * c * (32 / RL)));
                x = cmulc(x, w);
            }
            v[bitrev(q, bits)] = x;
        }
        dft_regs<R0, true>(v);
    }

    // ---- pass 0 with the twiddles built as they are used (one butterfly per thread: nothing to reuse) ------
    // w_q for q < R0/2 from the bases as expand_tw0 does (R0/2 - 1 values alive), w_q = w_{R0/2} w_{q-R0/2} for
    // the upper half on the fly: the same complex multiplies in another order, sixteen registers fewer at the
    // point where the whole butterfly is in registers too.  For kernels that have none to spare.
    struct Tw0Low {
        float2 w[R0 / 2 - 1];
    };
    static MGX_HD void expand_tw0_low(const Tw0& t, Tw0Low& f) {
        MGX_UNROLL
        for (int q = 1; q < R0 / 2; ++q) {
            int hb = 1;
            while (hb * 2 <= q) hb *= 2;
            f.w[q - 1] = q == hb ? t.b[ilog2(hb)] : cmul(t.b[ilog2(hb)], f.w[q - hb - 1]);
        }
    }
    static MGX_HD float2 tw0_at(const Tw0& t, const Tw0Low& f, int q) {         // q = 1 .. R0-1, compile-time after unrolling
        if (q < R0 / 2) return f.w[q - 1];
        if (q == R0 / 2) return t.b[LB0 - 1];
        return cmul(t.b[LB0 - 1], f.w[q - R0 / 2 - 1]);
    }
    static MGX_HD void fwd0_store_lean(float2 (&v)[R0], int tid, const Tw0& t, float2* lds) {
        static_assert(CNT(0) == 1 || P == 1, "one pass-0 butterfly per thread");
        constexpr int bits = lr(0);
        float2* p = lds + base<0>(tid);
        dft_regs<R0, false>(v);
        Tw0Low low;
        expand_tw0_low(t, low);
        MGX_UNROLL
        for (int q = 0; q < R0; ++q) {
            float2 x = v[bitrev(q, bits)];
            if (P > 1 && q != 0) x = cmul(x, tw0_at(t, low, q));
            p[off<0>(q)] = x;
        }
    }
    static MGX_HD void inv0_load_lean(float2 (&v)[R0], int tid, const Tw0& t, const float2* lds) {
        constexpr int bits = lr(0);
        const float2* p = lds + base<0>(tid);
        Tw0Low low;
        expand_tw0_low(t, low);
        MGX_UNROLL
        for (int q = 0; q < R0; ++q) {
            float2 x = p[off<0>(q)];
            if (P > 1 && q != 0) x = cmulc(x, tw0_at(t, low, q));
            v[bitrev(q, bits)] = x;
        }
        dft_regs<R0, true>(v);
    }

    // ---- middle passes (P >= 3), LDS -> LDS --------------------------------------------------
    template <int PASS>
    static MGX_HD void fwd_mid_pass(int tid, float2* lds, const float2* table) {
        constexpr int r = R(PASS), bits = lr(PASS), s = S(PASS);
        constexpr bool same = s <= 64;                   // the twiddle index n = u % s is the same for all c
        float2 w[r - 1];
        MGX_UNROLL
        for (int c = 0; c < CNT(PASS); ++c) {
            const int u = mid_butterfly<PASS>(tid, c);
            if (c == 0 || !same) {
                const int n = u % s;
                MGX_UNROLL
                for (int q = 1; q < r; ++q) w[q - 1] = table[(q - 1) * s + n];
            }
            float2* p = lds + base<PASS>(u);
            float2 v[r];
            MGX_UNROLL
            for (int j = 0; j < r; ++j) v[j] = p[off<PASS>(j)];
            dft_regs<r, false>(v);
            MGX_UNROLL
            for (int q = 0; q < r; ++q) {
                float2 x = v[bitrev(q, bits)];
                if (q != 0) x = cmul(x, w[q - 1]);
                p[off<PASS>(q)] = x;
            }
        }
    }
    template <int PASS>
    static MGX_HD void inv_mid_pass(int tid, float2* lds, const float2* table) {
        constexpr int r = R(PASS), bits = lr(PASS), s = S(PASS);
        constexpr bool same = s <= 64;
        float2 w[r - 1];
        MGX_UNROLL
        for (int c = 0; c < CNT(PASS); ++c) {
            const int u = mid_butterfly<PASS>(tid, c);
            if (c == 0 || !same) {
                const int n = u % s;
                MGX_UNROLL
                for (int q = 1; q < r; ++q) w[q - 1] = table[(q - 1) * s + n];
            }
            float2* p = lds + base<PASS>(u);
            float2 v[r];
            MGX_UNROLL
            for (int q = 0; q < r; ++q) {
                float2 x = p[off<PASS>(q)];
                if (q != 0) x = cmulc(x, w[q - 1]);
                v[bitrev(q, bits)] = x;
            }
            dft_regs<r, true>(v);
            MGX_UNROLL
            for (int j = 0; j < r; ++j) p[off<PASS>(j)] = v[j];
        }
    }
    // pass 1 (P >= 3) and pass 2 (P == 4); a barrier belongs between any two of them
    static MGX_HD void fwd_mid(int tid, float2* lds, const float2* table) { fwd_mid_pass<MID>(tid, lds, table); }
    static MGX_HD void inv_mid(int tid, float2* lds, const float2* table) { inv_mid_pass<MID>(tid, lds, table); }
    static MGX_HD void fwd_mid2(int tid, float2* lds, const float2* table) { fwd_mid_pass<MID2>(tid, lds, table + MID_TABLE1); }
    static MGX_HD void inv_mid2(int tid, float2* lds, const float2* table) { inv_mid_pass<MID2>(tid, lds, table + MID_TABLE1); }

    // ---- last pass: one contiguous row per thread -------------------------------------------
    static MGX_HD bool has_row(int tid) { return !partial(LAST) || tid < L; }
    // row u = tid: elements tid*RL .. tid*RL + RL-1 (padded)
    static MGX_HD void load_row(float2 (&v)[RL], int row, const float2* lds) {
        const float2* p = lds + base<LAST>(row);
        if (PADDED) {
            MGX_UNROLL
            for (int e = 0; e < RL; e += 2) {
                const float4 t = *reinterpret_cast<const float4*>(p + e);
                v[e] = make_float2(t.x, t.y);
                v[e + 1] = make_float2(t.z, t.w);
            }
        } else {
            MGX_UNROLL
            for (int e = 0; e < RL; ++e) v[e] = p[e];
        }
    }
    // elements [E0, E0+CNT) of a row (E0, CNT even)
    template <int E0, int CNT_>
    static MGX_HD void load_row_part(float2 (&v)[CNT_], int row, const float2* lds) {
        const float2* p = lds + base<LAST>(row) + E0;
        if (PADDED) {
            MGX_UNROLL
            for (int e = 0; e < CNT_; e += 2) {
                const float4 t = *reinterpret_cast<const float4*>(p + e);
                v[e] = make_float2(t.x, t.y);
                v[e + 1] = make_float2(t.z, t.w);
            }
        } else {
            MGX_UNROLL
            for (int e = 0; e < CNT_; ++e) v[e] = p[e];
        }
    }
    // elements [E0, E0+CNT) of a row from v[E0 ..] (E0, CNT even)
    template <int E0, int CNT_>
    static MGX_HD void store_row_part(const float2 (&v)[RL], int row, float2* lds) {
        float2* p = lds + base<LAST>(row);
        if (PADDED) {
            MGX_UNROLL
            for (int e = E0; e < E0 + CNT_; e += 2)
                *reinterpret_cast<float4*>(p + e) = make_float4(v[e].x, v[e].y, v[e + 1].x, v[e + 1].y);
        } else {
            MGX_UNROLL
            for (int e = E0; e < E0 + CNT_; ++e) p[e] = v[e];
        }
    }
    static MGX_HD void store_row(const float2 (&v)[RL], int row, float2* lds) {
        float2* p = lds + base<LAST>(row);
        if (PADDED) {
            MGX_UNROLL
            for (int e = 0; e < RL; e += 2)
                *reinterpret_cast<float4*>(p + e) = make_float4(v[e].x, v[e].y, v[e + 1].x, v[e + 1].y);
        } else {
            MGX_UNROLL
            for (int e = 0; e < RL; ++e) p[e] = v[e];
        }
    }

    // ---- frequency index <-> position ---------------------------------------------------------
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
    // Row holding the mirror bins of row `row` != 0: bin k at (row, q) has N-k at
    // (mirror_row(row), RL-1-q).  Row 0 mirrors into itself: (0, q) <-> (0, (RL-q) % RL).
    static MGX_HD int mirror_row(int row) {
        // row = sum_{p<LAST} q_p * S(p)/RL ; negate the mixed-radix number (q_0 least significant)
        int out = 0;
        bool borrow = false;      // becomes true after the lowest non-zero digit
        MGX_UNROLL
        for (int p = 0; p < LAST; ++p) {
            const int w = S(p) / RL;
            const int q = (row / w) & (R(p) - 1);
            int m;
            if (borrow) m = R(p) - 1 - q;
            else if (q != 0) { m = R(p) - q; borrow = true; }
            else m = 0;
            out += m * w;
        }
        return out;
    }
};

}  // namespace mgx

#if defined(__clang__)
#pragma clang fp contract(off)
#endif
