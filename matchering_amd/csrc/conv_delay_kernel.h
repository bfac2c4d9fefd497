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
// Two transforms per block instead of three, every frame fetched twice (the second time from the L2: the same
// workgroup asked for it one block earlier) instead of six times, at the price of one more trip of the row through
// LDS (the mirror bins live on another thread) and two more barriers per block.  (Keeping the newer half of a
// window in 16 registers for the next block -- every frame fetched once -- was built and measured: the kernel has
// no register to spare, 341 against 312 us.)
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

    struct Row {                                      // the thread's row of bins, X[q] at v[bitrev(q)]
        float2 v[RL];
    };
    struct Carry {                                    // partition 1's product, same order
        float2 w[RL];
    };
    static MGX_HD void clear(Carry& c) {
        MGX_UNROLL
        for (int q = 0; q < RL; ++q) c.w[q] = make_float2(0.f, 0.f);
    }

    // block b: output frames [b*HOP, (b+1)*HOP), window = the N frames from b*HOP on (conv2_kernel.h first_input
    // with two partitions, k = 0); b = -1 is the carry-only block in front of the track
    static MGX_HD long long first_frame(long long b) { return b * (long long)HOP; }

    // ---- frames -> z = mid + j side -> pass 0 -> LDS -------------------------------------------
    static MGX_HD void phase_load(int tid, long long b, const Conv2Args& a, const Persist& ps, float2* lds) {
        typename F::Tw0Full tw;
        F::expand_tw0(ps.tw0, tw);
        const long long i0 = first_frame(b);
        const MemView src = mem_view(a.x, a.n * 8);
        MGX_UNROLL
        for (int c = 0; c < CNT0; ++c) {
            const unsigned lane = ((unsigned)i0 + (unsigned)(tid + c * T)) * 8u;
            float2 f[R0];
            if (i0 < 0) {                             // (uniform) the window starts before the track: see conv2_kernel.h
                MGX_UNROLL
                for (int j = 0; j < R0; ++j) f[j] = ld_f2_or_zero(src, lane + (unsigned)(j * S0 * 8));
            } else {
                MGX_UNROLL
                for (int j = 0; j < R0; ++j) f[j] = ld_f2(src, lane, (unsigned)(j * S0 * 8));
            }
            float2 v[R0];
            MGX_UNROLL
            for (int j = 0; j < R0; ++j) {
                const float m = (f[j].x + f[j].y) * 0.5f;            // dsp.py:59-60
                v[j] = make_float2(m, m - f[j].y);                   // dsp.py:62
            }
            F::fwd0_store(v, tid, c, tw, lds);
        }
    }

    // ---- last forward pass on the thread's row; the bins also go back to LDS in position order, where the
    // thread of the mirror row finds them ------------------------------------------------------------------
    static MGX_HD void phase_row(int tid, Row& own, float2* lds) {
        F::load_row(own.v, tid, lds);
        dft_regs<RL, false>(own.v);
        float2 w[RL];
        MGX_UNROLL
        for (int q = 0; q < RL; ++q) w[q] = own.v[bitrev(q, F::lr(F::LAST))];
        F::store_row(w, tid, lds);
    }

    // ---- own bins and mirror bins -> this block's spectrum (left in `own`) and the next block's carry.
    // Bin (row, q) mirrors to (mirror_row, RL-1-q); row 0 mirrors into itself one element further up (fft2.h).
    // Filter tables as conv2_kernel.h: [partition][q*L + row], scaled by gain/N.  Four complex filter values per
    // bin are 128 registers per row, and the kernel has 128 in all: the row is worked through in chunks of CH bins
    // -- a chunk's values are asked for (fetch_filters), the chunk is computed and pinned (multiply_chunk), and only
    // then the next chunk is asked for.  Measured on config #5 (profiles/r04_n_conv_delay.txt): chunks of 2 bins
    // and no scratch 312 us, the first chunk asked for a phase early (16 B of scratch) 319, chunks of 4 (80 B) 372;
    // without the pins the compiler sinks all arithmetic below all loads and spills 660 B.
    static constexpr int CH = 2, CHUNKS = RL / CH;
    struct Filters {
        float2 m0[CH], s0[CH], m1[CH], s1[CH];        // mid / side spectra of partitions 0 and 1 at the chunk's bins
    };
    template <int C>
    static MGX_HD void fetch_filters(int tid, const Conv2Args& a, Filters& f) {
        const MemView hm = mem_view(a.h_mid, (long long)2 * N * 8), hs = mem_view(a.h_side, (long long)2 * N * 8);
        MGX_UNROLL
        for (int e = 0; e < CH; ++e) {
            const unsigned at = (unsigned)((C * CH + e) * F::L * 8);
            f.m0[e] = ld_f2(hm, (unsigned)tid * 8u, at);
            f.s0[e] = ld_f2(hs, (unsigned)tid * 8u, at);
            f.m1[e] = ld_f2(hm, (unsigned)tid * 8u, at + (unsigned)(N * 8));
            f.s1[e] = ld_f2(hs, (unsigned)tid * 8u, at + (unsigned)(N * 8));
        }
    }
    template <int C>
    static MGX_HD void multiply_chunk(int tid, int mrow, const Filters& f, Row& own, Carry& carry, const float2* lds) {
        constexpr int bits = F::lr(F::LAST);
        float2 m[CH];                                  // mirror row, elements RL-1-q for the chunk's q, ascending
        F::template load_row_part<RL - (C + 1) * CH, CH>(m, mrow, lds);
        if (tid == 0) {                                // row 0 is its own mirror row, one element further up
            MGX_UNROLL
            for (int e = 0; e < CH; ++e) m[CH - 1 - e] = lds[F::template base<F::LAST>(0) + (RL - (C * CH + e)) % RL];
        }
        MGX_UNROLL
        for (int e = 0; e < CH; ++e) {
            const int q = C * CH + e, i = bitrev(q, bits);
            const float2 z = own.v[i], zm = cconj(m[CH - 1 - e]);             // Z[k], conj Z[N-k]
            const float2 p0 = make_float2(0.5f * (f.m0[e].x + f.s0[e].x), 0.5f * (f.m0[e].y + f.s0[e].y));
            const float2 q0 = make_float2(0.5f * (f.m0[e].x - f.s0[e].x), 0.5f * (f.m0[e].y - f.s0[e].y));
            const float2 p1 = make_float2(0.5f * (f.m1[e].x + f.s1[e].x), 0.5f * (f.m1[e].y + f.s1[e].y));
            const float2 q1 = make_float2(0.5f * (f.m1[e].x - f.s1[e].x), 0.5f * (f.m1[e].y - f.s1[e].y));
            own.v[i] = cadd(cadd(cmul(z, p0), cmul(zm, q0)), carry.w[i]);
            carry.w[i] = cadd(cmul(z, p1), cmul(zm, q1));
            mgx_pin(own.v[i]);                         // (computed here, before the next chunk's filter values arrive)
            mgx_pin(carry.w[i]);
        }
    }
    template <int C>
    static MGX_HD void multiply_from(int tid, int mrow, const Conv2Args& a, Filters& f, Row& own, Carry& carry,
                                     const float2* lds) {
        if constexpr (C < CHUNKS) {
            fetch_filters<C>(mgx_opaque(tid), a, f);                  // (asked for here, not a chunk earlier: no registers)       // (asked for here, not a chunk earlier: no registers)
            multiply_chunk<C>(tid, mrow, f, own, carry, lds);
            MGX_SCHED_FENCE();
            multiply_from<C + 1>(tid, mrow, a, f, own, carry, lds);
        }
    }
    static MGX_HD void phase_multiply(int tid, const Conv2Args& a, Row& own, Carry& carry, const float2* lds) {
        Filters f;
        multiply_from<0>(tid, F::mirror_row(tid), a, f, own, carry, lds);
    }

    // ---- first inverse pass on the row (a barrier after phase_multiply: every mirror row has been read) ----
    static MGX_HD void phase_row_back(int tid, Row& own, float2* lds) {
        dft_regs<RL, true>(own.v);
        F::store_row(own.v, tid, lds);
    }

    // ---- last inverse pass + epilogue: real = mid, imaginary = side; L = mid + side, R = mid - side (dsp.py:67-68).
    // Returns the thread's max(|L|,|R|) over frames of the track; stores past its end are dropped by the range check.
    static MGX_HD float phase_store(int tid, long long b, const Conv2Args& a, const Persist& ps, const float2* lds) {
        float peak = 0.f;
        typename F::Tw0Full tw;
        F::expand_tw0(ps.tw0, tw);
        const MemView dst = mem_view(a.y, a.n * 8);
        const MemView dm = mem_view(a.ymid, a.ymid ? a.n * 4 : 0);
        const unsigned frames = (unsigned)a.n, o0 = (unsigned)first_frame(b);
        MGX_UNROLL
        for (int c = 0; c < CNT0; ++c) {
            const unsigned first = o0 + (unsigned)(tid + c * T);
            float2 v[R0];
            F::inv0_load(v, tid, c, tw, lds);
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
