// Matching-EQ FIR of F taps on blocks of N = 4F points: three quarters of every block are fresh output.
//
// Replaces the same lines as conv2_kernel.h (match_frequencies.py:104-119, two fftconvolve "same" + ms_to_lr) for
// F = 4096, the reference's default fft_size, on N = 16384.  conv2_kernel.h runs that filter on N = 2F = 8192:
// half of every transform's output is overlap, 4 transforms of 8192 points per 8192 frames, 6.2 k VALU
// instructions per thread and pair on a kernel that is bound by exactly that (the VALU issue
// roofline).  On N = 4F a quarter is overlap: 2 transforms of 16384 points per 12288 frames -- a third fewer points
// per frame, 7.5 % fewer instructions (178 against 192.5 per frame: the fourth pass of the larger transform costs a
// layer of twiddles and the un-mixing below four complex multiplies per pair of bins), and 14.5 KB of code instead of
// 38 KB, which is what a slow box of the pool pays for.  Round 1 and round 4 both measured the
// trade and lost it (264 against 148 us) with a 16384-point transform that took 15 us of a CU; the one
// conv_delay_kernel.h was built around takes a third of that, and this kernel is that one with a single filter
// partition (145-147 against 149-151 us on fast boxes, 144 against 157 on slow ones: profiles/r04_wide_blocks.txt):
//
//  * both channels share one transform, z = mid + j side, un-mixed at the multiply through the mirror bin;
//  * a thread multiplies PAIRS of bins -- the lower half of its own row and their mirrors in the upper half of the
//    mirror row -- with one fetch of the two filter values for both (conv_delay_kernel.h has the algebra: this is
//    its Y[k], Y[N-k] without the carry);
//  * no carry, so blocks are independent: workgroup w takes blocks w, w + G, w + 2G, ... and, having 32 registers
//    more than the delay line, asks for the whole window of its next block a block ahead.
//
// Window of block b (output frames [b*HOP, (b+1)*HOP), HOP = N - F): the N frames from b*HOP - F/2 on (scipy's
// "same" centring, the filter delayed by one sample as in conv2_kernel.h); circular outputs [F, N) are the block.
#pragma once

#include "conv_delay_kernel.h"

#if defined(__clang__)
#pragma clang fp contract(fast)        // see fft2.h
#endif

namespace mgx {

template <int LOG2N>
struct ConvWide {
    using CD = ConvDelay<LOG2N>;
    using CB = Conv2Block<LOG2N>;
    using F = Fft2<LOG2N>;
    static constexpr int N = F::N;
    static constexpr int T = F::T;
    static constexpr int TAPS = N / 4;
    static constexpr int HOP = N - TAPS;              // output frames per block
    static constexpr int R0 = F::R0;
    static constexpr int RL = F::RL;
    static constexpr int S0 = F::S(0);
    static constexpr int CNT0 = F::CNT(0);
    static constexpr int SKIP = TAPS / S0;            // leading outputs of a pass-0 butterfly that are circular garbage
    static constexpr int KEEP = R0 - SKIP;
    static constexpr int HALFROW = RL / 2;
    static_assert(!F::partial(0) && !F::partial(F::LAST), "every thread owns a pass-0 butterfly and a row");
    static_assert(TAPS % S0 == 0 && SKIP >= 1, "block geometry must follow the pass-0 stride");
    static_assert(F::PADDED, "row 0's unused pair slot lands in the padding behind the row");
    using Persist = typename CB::Persist;

    static MGX_HD long long first_output(long long b) { return b * (long long)HOP; }
    static MGX_HD long long first_input(long long b) { return first_output(b) - TAPS / 2; }

    // ---- frames -> z = mid + j side -> pass 0 -> LDS ----------------------------------------------------------
    struct Frames {
        float2 f[CNT0][R0];                           // (L, R) as loaded
    };
    static MGX_HD void fetch(int tid, long long b, const Conv2Args& a, Frames& fr) {
        const long long i0 = first_input(b);
        const MemView src = mem_view(a.x, a.n * 8);
        MGX_UNROLL
        for (int c = 0; c < CNT0; ++c) {
            const unsigned lane = ((unsigned)i0 + (unsigned)(tid + c * T)) * 8u;
            if (i0 < 0) {                             // (uniform) the window starts before the track: see conv2_kernel.h
                MGX_UNROLL
                for (int j = 0; j < R0; ++j) fr.f[c][j] = ld_f2_or_zero(src, lane + (unsigned)(j * S0 * 8));
            } else {
                MGX_UNROLL
                for (int j = 0; j < R0; ++j) fr.f[c][j] = ld_f2(src, lane, (unsigned)(j * S0 * 8));
            }
        }
    }
    static MGX_HD void phase_pass0(int tid, const Persist& ps, const Frames& fr, float2* lds) {
        typename F::Tw0Full tw;
        if (CNT0 > 1) F::expand_tw0(ps.tw0, tw);
        MGX_UNROLL
        for (int c = 0; c < CNT0; ++c) {
            float2 v[R0];
            MGX_UNROLL
            for (int j = 0; j < R0; ++j) v[j] = CD::to_ms(fr.f[c][j]);
            if constexpr (CNT0 == 1) F::fwd0_store_lean(v, tid, ps.tw0, lds);
            else F::fwd0_store(v, tid, c, tw, lds);
        }
    }

    // ---- the row passes are the delay line's ----------------------------------------------------------------------
    static MGX_HD void phase_row(int tid, float2* lds) { CD::phase_row(tid, lds); }
    static MGX_HD void phase_row_back(int tid, float2* lds) { CD::phase_row_back(tid, lds); }

    // ---- the multiply: Z -> Y in place (LDS).  Pairs e = 0 .. RL/2-1 of the thread: bin k at (row, e), bin N-k at
    // (mirror_row, RL-1-e); with P = (Hm + Hs)/2, Q = (Hm - Hs)/2 at k
    //     Y[k] = Z P + conj(W) Q,      Y[N-k] = conj(conj(W) P + Z Q)
    // (conv_delay_kernel.h, which also says how thread 0 handles row 0: partner positions one element further up,
    // slot 0 redone as the two self-mirrored bins 0 and N/2).  Filter tables [q*L + row], scaled by gain/N.  All
    // sixteen filter values of a thread (plus thread 0's two of position RL/2) are asked for by the caller a phase
    // early -- fetch_filters() -- so that they are there when the barrier in front of this phase opens.
    struct Filters {
        float2 m[HALFROW], s[HALFROW];                // mid / side spectra at positions 0 .. RL/2-1 of the row
        float2 mh, sh;                                // thread 0: at position RL/2
    };
    static MGX_HD void fetch_filters(int tid, const Conv2Args& a, Filters& f) {
        const MemView hm = mem_view(a.h_mid, (long long)N * 8), hs = mem_view(a.h_side, (long long)N * 8);
        MGX_UNROLL
        for (int e = 0; e < HALFROW; ++e) {
            f.m[e] = ld_f2(hm, (unsigned)tid * 8u, (unsigned)(e * F::L * 8));
            f.s[e] = ld_f2(hs, (unsigned)tid * 8u, (unsigned)(e * F::L * 8));
        }
        f.mh = f.sh = make_float2(0.f, 0.f);
        if (tid == 0) {
            f.mh = ld_f2(hm, 0u, (unsigned)(HALFROW * F::L * 8));
            f.sh = ld_f2(hs, 0u, (unsigned)(HALFROW * F::L * 8));
        }
    }
    static MGX_HD void pair_products(float2 z, float2 w, float2 m, float2 s, float2& yk, float2& ynk) {
        const float2 cw = cconj(w);
        const float2 p = CD::half_sum(m, s), q = CD::half_diff(m, s);
        yk = cadd(cmul(z, p), cmul(cw, q));
        ynk = cconj(cadd(cmul(cw, p), cmul(z, q)));
    }
    static MGX_HD void phase_multiply(int tid, const Filters& f, float2* lds) {
        // (thread 0: the partner positions of row 0 lie one element further up)
        const int partner = F::template base<F::LAST>(F::mirror_row(tid)) + (tid == 0 ? 1 : 0);
        float2* own = lds + F::template base<F::LAST>(tid);
        float2 z[HALFROW], w[HALFROW];
        F::template load_row_part<0, HALFROW>(z, tid, lds);
        MGX_UNROLL
        for (int e = 0; e < HALFROW; ++e) w[e] = lds[partner + RL - 1 - e];
        float2 zh = make_float2(0.f, 0.f);
        if (tid == 0) zh = own[HALFROW];
        MGX_UNROLL
        for (int e = 0; e < HALFROW; ++e) {
            float2 yk, ynk;
            pair_products(z[e], w[e], f.m[e], f.s[e], yk, ynk);
            own[e] = yk;
            lds[partner + RL - 1 - e] = ynk;
        }
        if (tid == 0) {                                // bins 0 and N/2: each its own mirror
            float2 y0, yh, unused;
            pair_products(z[0], z[0], f.m[0], f.s[0], y0, unused);
            pair_products(zh, zh, f.mh, f.sh, yh, unused);
            own[0] = y0;
            own[HALFROW] = yh;
        }
    }

    // ---- the same three phases with the thread's OWN half of the row kept in registers (round 6) ---------------------
    // Row forward -> multiply -> row back moves a row through the LDS twice, but only half of it ever belongs to
    // another thread: the partner needs this row's UPPER half (the mirrors of its own lower half) and hands back the
    // products that live there.  So: the row pass stores the upper half only and keeps the lower half; the multiply
    // reads the partner's upper half, keeps its own products and stores the partner's; the row pass back reads the
    // upper half the partner wrote and starts from registers for the rest.  Per thread 4 + 8 stores and 4 + 4 loads
    // fewer (132 of a block's 1075 LDS cycles per wave), the same arithmetic in the same order: bit-identical.
    struct Kept {
        float2 z[HALFROW];                            // positions 0 .. RL/2-1 of the own row: Z, then Y
    };
    static MGX_HD void phase_row_keep(int tid, Kept& k, float2* lds) {
        float2 v[RL], w[RL];
        F::load_row(v, tid, lds);
        dft_regs<RL, false>(v);
        MGX_UNROLL
        for (int q = 0; q < RL; ++q) w[q] = v[bitrev(q, F::lr(F::LAST))];
        F::template store_row_part<HALFROW, HALFROW>(w, tid, lds);
        MGX_UNROLL
        for (int e = 0; e < HALFROW; ++e) k.z[e] = w[e];
    }
    static MGX_HD void phase_multiply_keep(int tid, const Filters& f, Kept& k, float2* lds) {
        const int partner = F::template base<F::LAST>(F::mirror_row(tid)) + (tid == 0 ? 1 : 0);
        float2* own = lds + F::template base<F::LAST>(tid);
        float2 w[HALFROW];
        MGX_UNROLL
        for (int e = 0; e < HALFROW; ++e) w[e] = lds[partner + RL - 1 - e];
        float2 zh = make_float2(0.f, 0.f);
        if (tid == 0) zh = own[HALFROW];
        const float2 z0 = k.z[0];
        MGX_UNROLL
        for (int e = 0; e < HALFROW; ++e) {
            float2 yk, ynk;
            pair_products(k.z[e], w[e], f.m[e], f.s[e], yk, ynk);
            k.z[e] = yk;
            lds[partner + RL - 1 - e] = ynk;
        }
        if (tid == 0) {                                // bins 0 and N/2: each its own mirror
            float2 y0, yh, unused;
            pair_products(z0, z0, f.m[0], f.s[0], y0, unused);
            pair_products(zh, zh, f.mh, f.sh, yh, unused);
            k.z[0] = y0;
            own[HALFROW] = yh;
        }
    }
    static MGX_HD void phase_row_back_keep(int tid, const Kept& k, float2* lds) {
        float2 w[RL], v[RL], upper[HALFROW];
        F::template load_row_part<HALFROW, HALFROW>(upper, tid, lds);
        MGX_UNROLL
        for (int e = 0; e < HALFROW; ++e) {
            w[e] = k.z[e];
            w[HALFROW + e] = upper[e];
        }
        MGX_UNROLL
        for (int q = 0; q < RL; ++q) v[bitrev(q, F::lr(F::LAST))] = w[q];
        dft_regs<RL, true>(v);
        F::store_row(v, tid, lds);
    }

    // ---- last inverse pass + epilogue: real = mid, imaginary = side; L = mid + side, R = mid - side (dsp.py:67-68).
    // Returns the thread's max(|L|,|R|) over frames of the track; stores past its end are dropped by the range check.
    static MGX_HD float phase_store(int tid, long long b, const Conv2Args& a, const Persist& ps, const float2* lds) {
        float peak = 0.f;
        typename F::Tw0Full tw;
        if (CNT0 > 1) F::expand_tw0(ps.tw0, tw);
        const MemView dst = mem_view(a.y, a.n * 8);
        const MemView dm = mem_view(a.ymid, a.ymid ? a.n * 4 : 0);
        const unsigned frames = (unsigned)a.n, o0 = (unsigned)first_output(b);
        MGX_UNROLL
        for (int c = 0; c < CNT0; ++c) {
            const unsigned first = o0 + (unsigned)(tid + c * T);
            float2 v[R0];
            if constexpr (CNT0 == 1) F::inv0_load_lean(v, tid, ps.tw0, lds);
            else F::inv0_load(v, tid, c, tw, lds);
            MGX_UNROLL
            for (int j = 0; j < KEEP; ++j) {
                const float2 ms = v[SKIP + j];
                const float2 y = make_float2(ms.x + ms.y, ms.x - ms.y);
                st_f2<CONV_STORE_AUX>(dst, first * 8u, (unsigned)(j * S0 * 8), y);
                st_f1<CONV_STORE_AUX>(dm, first * 4u, (unsigned)(j * S0 * 4), ms.x);
                const float p = fmaxf(fabsf(y.x), fabsf(y.y));
                peak = fmaxf(peak, first + (unsigned)(j * S0) < frames ? p : 0.f);
            }
        }
        return peak;
    }

    // ---- filter preparation: the F taps, delayed by one sample, zero-extended to N (conv2_kernel.h phase_load_taps
    // knows N/2 taps only) -> pass 0 -> LDS; the rest of the transform and phase_write_filter are conv2_kernel.h's
    static MGX_HD void phase_load_taps(int tid, const float* taps, const Persist& ps, float2* lds) {
        typename F::Tw0Full tw;
        F::expand_tw0(ps.tw0, tw);
        MGX_UNROLL
        for (int c = 0; c < CNT0; ++c) {
            const int u = tid + c * T;
            float2 v[R0];
            MGX_UNROLL
            for (int j = 0; j < R0; ++j) {
                const int i = u + j * S0;                         // h'[i] = h[i-1], i in [1, F]
                const bool ok = i >= 1 && i <= TAPS;
                const float t = taps[ok ? i - 1 : 0];             // unconditional load, then select
                v[j] = make_float2(ok ? t : 0.f, 0.f);
            }
            F::fwd0_store(v, tid, c, tw, lds);
        }
    }
};

}  // namespace mgx

#if defined(__clang__)
#pragma clang fp contract(off)
#endif
