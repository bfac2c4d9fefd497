// Matching-EQ FIR of TWO partitions (taps = N: config #5, 16384 taps on N = 16384 blocks) as a
// frequency-domain delay line: one forward and one inverse transform per block of N/2 output frames.
//
// Replaces the same lines as conv2_kernel.h (match_frequencies.py:104-119, two fftconvolve "same" + ms_to_lr) for
// that filter length.  The partitioned kernel there transforms every window once per partition AND per channel
// (K forward transforms per channel and pair of blocks, each fetching its own N + N/2 frames): three transforms of
// N points per block and six fetches of every frame for K = 2.  Here
//
//  * both channels share one transform: z = mid + j side, Z = M + jS.  The channels have different filters, so the
//    product is formed through the mirror bin: with P = (Hm + Hs)/2 and Q = (Hm - Hs)/2
//        Y[k] = M Hm + j S Hs = Z[k] P[k] + conj(Z[N-k]) Q[k],
//    and the inverse transform returns the mid result in the real and the side result in the imaginary part;
//  * partition 1's window of block b IS partition 0's window of block b-1 (the partitions are N/2 taps apart
//    and so are the blocks), so a workgroup walks a run of consecutive blocks and carries
//        C_b[k] = Z_b[k] P1[k] + conj(Z_b[N-k]) Q1[k]
//    to the next block in registers (RL float2 per thread): Y_b = Z_b P0 + conj(Zm_b) Q0 + C_{b-1}.
//    A run starts with one forward transform of the block before it (the carry only).
//
//  * a thread multiplies PAIRS of bins: the lower half of its own row and their mirrors, which are the upper half of
//    the mirror row (fft2.h: (row, q) <-> (mirror_row, RL-1-q)).  The filters are real sequences' spectra, so the
//    four filter values at bin N-k are the conjugates of those at k: one fetch serves both bins of a pair, which
//    halves what the phase moves from the L2 -- and that, not arithmetic or latency, is what the phase takes
//    (profiles/r04_u_conv_delay_phases.txt);
//  * the older half of a block's window is the newer half of the block before: it stays in registers, and the newer
//    half is asked for a block ahead (k_conv_delay in mgx_kernels.h; profiles/r04_v_conv_delay_pairs.txt).
//
// Two transforms per block instead of three, every frame fetched once instead of six times, half the filter
// values, at the price of one more trip of the row through LDS (the mirror bins live on another thread) and two
// more barriers per block.
#pragma once

#include "conv2_kernel.h"

#if defined(__clang__)
#pragma clang fp contract(fast)        // see fft2.h
#endif

namespace mgx {

template <int LOG2N>
struct ConvDelay {
    using CB = Conv2Block<LOG2N>;
    using F = Fft2<LOG2N>;
    static constexpr int N = F::N;
    static constexpr int T = F::T;
    static constexpr int HOP = N / 2;                 // output frames per block = taps per partition
    static constexpr int R0 = F::R0;
    static constexpr int RL = F::RL;
    static constexpr int S0 = F::S(0);
    static constexpr int CNT0 = F::CNT(0);
    static constexpr int SKIP = CB::SKIP;             // leading outputs of a pass-0 butterfly that are circular garbage
    static constexpr int HALF = CB::HALF;
    static_assert(!F::partial(0) && !F::partial(F::LAST), "every thread owns a pass-0 butterfly and a row");
    static_assert(HOP % S0 == 0, "block geometry must follow the pass-0 stride");
    using Persist = typename CB::Persist;

    static constexpr int HALFROW = RL / 2;
    static_assert(F::PADDED, "row 0's unused pair slot lands in the padding behind the row");
    struct Carry {                                    // partition 1's product: lo[e] for bin (row, e), hi[e] for its mirror
        float2 lo[HALFROW], hi[HALFROW];
    };
    static MGX_HD void clear(Carry& c) {
        MGX_UNROLL
        for (int e = 0; e < HALFROW; ++e) c.lo[e] = c.hi[e] = make_float2(0.f, 0.f);
    }

    // block b: output frames [b*HOP, (b+1)*HOP), window = the N frames from b*HOP on (conv2_kernel.h first_input
    // with two partitions, k = 0); b = -1 is the carry-only block in front of the track
    static MGX_HD long long first_frame(long long b) { return b * (long long)HOP; }

    // ---- frames -> z = mid + j side -> pass 0 -> LDS ------------------------------------------------------------
    // The older half of a window is kept from the previous block: consecutive windows overlap by
    // HOP = (R0/2) * S0 frames, i.e. inputs R0/2 .. R0-1 of a thread's pass-0 butterfly are its inputs 0 .. R0/2-1
    // one block later.  prime() takes the older half of a run's first window; fetch_half<R0/2>() asks for the newer
    // half of a window (as early as the caller likes); phase_pass0_held() consumes it: every frame is fetched once
    // per run.
    struct HeldFrames {
        float2 z[CNT0][R0 / 2];                       // (mid, side)
    };
    struct HalfFrames {
        float2 f[CNT0][R0 / 2];                       // (L, R) as loaded
    };
    static MGX_HD float2 to_ms(float2 f) {
        const float m = (f.x + f.y) * 0.5f;                              // dsp.py:59-60
        return make_float2(m, m - f.y);                                  // dsp.py:62
    }
    template <int J0>
    static MGX_HD void fetch_half(int tid, long long b, const Conv2Args& a, HalfFrames& h) {
        const long long i0 = first_frame(b);
        const MemView src = mem_view(a.x, a.n * 8);
        MGX_UNROLL
        for (int c = 0; c < CNT0; ++c) {
            const unsigned lane = ((unsigned)i0 + (unsigned)(tid + c * T)) * 8u;
            if (i0 < 0) {
                MGX_UNROLL
                for (int j = 0; j < R0 / 2; ++j) h.f[c][j] = ld_f2_or_zero(src, lane + (unsigned)((J0 + j) * S0 * 8));
            } else {
                MGX_UNROLL
                for (int j = 0; j < R0 / 2; ++j) h.f[c][j] = ld_f2(src, lane, (unsigned)((J0 + j) * S0 * 8));
            }
        }
    }
    static MGX_HD void prime(int tid, long long b, const Conv2Args& a, HeldFrames& held) {
        HalfFrames h;
        fetch_half<0>(tid, b, a, h);
        MGX_UNROLL
        for (int c = 0; c < CNT0; ++c) {
            MGX_UNROLL
            for (int j = 0; j < R0 / 2; ++j) held.z[c][j] = to_ms(h.f[c][j]);
        }
    }
    static MGX_HD void phase_pass0_held(int tid, const Persist& ps, HeldFrames& held, const HalfFrames& newer, float2* lds) {
        typename F::Tw0Full tw;
        if (CNT0 > 1) F::expand_tw0(ps.tw0, tw);
        MGX_UNROLL
        for (int c = 0; c < CNT0; ++c) {
            float2 v[R0];
            MGX_UNROLL
            for (int j = 0; j < R0 / 2; ++j) {
                v[j] = held.z[c][j];
                v[j + R0 / 2] = held.z[c][j] = to_ms(newer.f[c][j]);
            }
            if constexpr (CNT0 == 1) F::fwd0_store_lean(v, tid, ps.tw0, lds);     // (the 16384-point plan: fft2.h)
            else F::fwd0_store(v, tid, c, tw, lds);
        }
    }

    // ---- last forward pass on the thread's row; the bins go back to LDS in position order, where the multiply
    // finds them again: its own thread (the registers are needed for the filter values meanwhile) and the thread
    // of the mirror row ------------------------------------------------------------------------------------------
    static MGX_HD void phase_row(int tid, float2* lds) {
        float2 v[RL], w[RL];
        F::load_row(v, tid, lds);
        dft_regs<RL, false>(v);
        MGX_UNROLL
        for (int q = 0; q < RL; ++q) w[q] = v[bitrev(q, F::lr(F::LAST))];
        F::store_row(w, tid, lds);
    }

    // ---- the multiply: Z -> Y in place (LDS), and the next block's carry ---------------------------------------
    // Thread `row` owns the pairs e = 0 .. RL/2-1: bin k at (row, e) and bin N-k at (mirror_row, RL-1-e).  With
    // Z = Z[k], W = Z[N-k] and the four filter values at k (those at N-k are their conjugates):
    //     Y[k]   = Z P0 + conj(W) Q0 + C[k]                C'[k]   = Z P1 + conj(W) Q1
    //     Y[N-k] = conj(conj(W) P0 + Z Q0) + C[N-k]        C'[N-k] = conj(conj(W) P1 + Z Q1)
    // Y goes back to where Z came from: the lower half of the own row and the upper half of the mirror row -- which
    // no other thread touches in this phase (mirror_row is an involution), so the phase needs a barrier in front
    // (phase_row's stores) and one behind (phase_row_back reads the half the partner wrote) and none inside.
    // Row 0 is its own mirror row, one element further up: (0, e) <-> (0, RL-e) for e >= 1, and bins 0 and N/2
    // (positions 0 and RL/2) are their own mirrors.  Thread 0 runs the common code with its partner positions
    // shifted by one (its slot e = 0 then reads and writes the padding behind the row) and redoes slot 0 as the two
    // self-mirrored bins afterwards (row0_slot0), the filter values of position RL/2 fetched beside chunk 0's.
    // Filter tables as conv2_kernel.h: [partition][q*L + row], scaled by gain/N.  The kernel has 128 registers in
    // all, the carry takes 32: a chunk of CHP pairs (8 registers of filter values each) is asked for, computed and
    // pinned before the next is asked for -- loads tied to an opaque lane offset or they are hoisted, results pinned
    // or the arithmetic sinks below all loads, whose results are then all alive (660 B of scratch).
    static constexpr int CHP = 2;
    template <int CNT_>
    struct Filters {
        float2 m0[CNT_], s0[CNT_], m1[CNT_], s1[CNT_];       // mid / side spectra of partitions 0 and 1 at positions Q ..
    };
    template <int Q, int CNT_>
    static MGX_HD void fetch_filters(int tid, const Conv2Args& a, Filters<CNT_>& f) {
        const MemView hm = mem_view(a.h_mid, (long long)2 * N * 8), hs = mem_view(a.h_side, (long long)2 * N * 8);
        MGX_UNROLL
        for (int e = 0; e < CNT_; ++e) {
            const unsigned at = (unsigned)((Q + e) * F::L * 8);
            f.m0[e] = ld_f2(hm, (unsigned)tid * 8u, at);
            f.s0[e] = ld_f2(hs, (unsigned)tid * 8u, at);
            f.m1[e] = ld_f2(hm, (unsigned)tid * 8u, at + (unsigned)(N * 8));
            f.s1[e] = ld_f2(hs, (unsigned)tid * 8u, at + (unsigned)(N * 8));
        }
    }
    // P = (Hm + Hs)/2, Q = (Hm - Hs)/2: z = mid + j side un-mixed (see the head of the file)
    static MGX_HD float2 half_sum(float2 a, float2 b) { return make_float2(0.5f * (a.x + b.x), 0.5f * (a.y + b.y)); }
    static MGX_HD float2 half_diff(float2 a, float2 b) { return make_float2(0.5f * (a.x - b.x), 0.5f * (a.y - b.y)); }
    static MGX_HD void pair_products(float2 z, float2 w, float2 m0, float2 s0, float2 m1, float2 s1, float2& clo,
                                     float2& chi, float2& yk, float2& ynk) {
        const float2 cw = cconj(w);
        const float2 p0 = half_sum(m0, s0), q0 = half_diff(m0, s0), p1 = half_sum(m1, s1), q1 = half_diff(m1, s1);
        yk = cadd(cadd(cmul(z, p0), cmul(cw, q0)), clo);
        ynk = cadd(cconj(cadd(cmul(cw, p0), cmul(z, q0))), chi);
        clo = cadd(cmul(z, p1), cmul(cw, q1));
        chi = cconj(cadd(cmul(cw, p1), cmul(z, q1)));
    }
    // pairs E0 .. E0+CNT-1 of the thread; `partner` = LDS index of position 0 of the row that holds the mirrors
    template <int E0, int CNT_>
    static MGX_HD void multiply_chunk(int tid, int partner, const Filters<CNT_>& f, Carry& carry, float2* lds) {
        float2* own = lds + F::template base<F::LAST>(tid);
        float2 z[CNT_], w[CNT_];
        if constexpr (CNT_ % 2 == 0) {
            F::template load_row_part<E0, CNT_>(z, tid, lds);
        } else {
            MGX_UNROLL
            for (int e = 0; e < CNT_; ++e) z[e] = own[E0 + e];
        }
        MGX_UNROLL
        for (int e = 0; e < CNT_; ++e) w[e] = lds[partner + RL - 1 - (E0 + e)];
        MGX_UNROLL
        for (int e = 0; e < CNT_; ++e) {
            float2 yk, ynk;
            pair_products(z[e], w[e], f.m0[e], f.s0[e], f.m1[e], f.s1[e], carry.lo[E0 + e], carry.hi[E0 + e], yk, ynk);
            own[E0 + e] = yk;
            lds[partner + RL - 1 - (E0 + e)] = ynk;
            mgx_pin(carry.lo[E0 + e]);                 // (computed here, before the next chunk's filter values arrive)
            mgx_pin(carry.hi[E0 + e]);
        }
    }
    // thread 0, slot 0: bins 0 and N/2, each its own mirror (old: the carries the slot held before this block)
    static MGX_HD void row0_slot0(float2 z0, float2 zh, const Filters<CHP>& f, const Filters<1>& fh, float2 old_lo,
                                  float2 old_hi, Carry& carry, float2* lds) {
        float2* own = lds + F::template base<F::LAST>(0);
        float2 unused_c = make_float2(0.f, 0.f), unused_y;
        float2 clo = old_lo, chi = old_hi, y0, yh;
        pair_products(z0, z0, f.m0[0], f.s0[0], f.m1[0], f.s1[0], clo, unused_c, y0, unused_y);
        pair_products(zh, zh, fh.m0[0], fh.s0[0], fh.m1[0], fh.s1[0], chi, unused_c, yh, unused_y);
        own[0] = y0;
        own[HALFROW] = yh;
        carry.lo[0] = clo;
        carry.hi[0] = chi;
    }
    template <int E0>
    static MGX_HD void multiply_from(int tid, int partner, const Conv2Args& a, Carry& carry, float2* lds) {
        if constexpr (E0 < HALFROW) {
            constexpr int CNT_ = HALFROW - E0 < CHP ? HALFROW - E0 : CHP;
            Filters<CNT_> f;
            fetch_filters<E0, CNT_>(mgx_opaque(tid), a, f);
            if constexpr (E0 == 0) {
                Filters<1> fh;
                fh.m0[0] = fh.s0[0] = fh.m1[0] = fh.s1[0] = make_float2(0.f, 0.f);
                float2 z0 = make_float2(0.f, 0.f), zh = z0;
                const float2 old_lo = carry.lo[0], old_hi = carry.hi[0];
                if (tid == 0) {
                    fetch_filters<HALFROW, 1>(tid, a, fh);
                    z0 = lds[F::template base<F::LAST>(0)];
                    zh = lds[F::template base<F::LAST>(0) + HALFROW];
                }
                multiply_chunk<E0, CNT_>(tid, partner, f, carry, lds);
                if (tid == 0) row0_slot0(z0, zh, f, fh, old_lo, old_hi, carry, lds);
            } else {
                multiply_chunk<E0, CNT_>(tid, partner, f, carry, lds);
            }
            MGX_SCHED_FENCE();
            multiply_from<E0 + CNT_>(tid, partner, a, carry, lds);
        }
    }
    static MGX_HD void phase_multiply(int tid, const Conv2Args& a, Carry& carry, float2* lds) {
        // (thread 0: the partner positions of row 0 lie one element further up, see above)
        const int partner = F::template base<F::LAST>(F::mirror_row(tid)) + (tid == 0 ? 1 : 0);
        multiply_from<0>(tid, partner, a, carry, lds);
    }

    // ---- first inverse pass on the row (a barrier after phase_multiply: the partner has written the upper half) ----
    static MGX_HD void phase_row_back(int tid, float2* lds) {
        float2 w[RL], v[RL];
        F::load_row(w, tid, lds);
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
        const unsigned frames = (unsigned)a.n, o0 = (unsigned)first_frame(b);
        MGX_UNROLL
        for (int c = 0; c < CNT0; ++c) {
            const unsigned first = o0 + (unsigned)(tid + c * T);
            float2 v[R0];
            if constexpr (CNT0 == 1) F::inv0_load_lean(v, tid, ps.tw0, lds);
            else F::inv0_load(v, tid, c, tw, lds);
            MGX_UNROLL
            for (int j = 0; j < HALF; ++j) {
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
};

}  // namespace mgx

#if defined(__clang__)
#pragma clang fp contract(off)
#endif
