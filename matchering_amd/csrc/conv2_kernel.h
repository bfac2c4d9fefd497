// Matching-EQ FIR convolution: overlap-save in LDS, two output blocks per complex FFT.
//
// Replaces matchering/stage_helpers/match_frequencies.py:104-119 (`convolve`: two scipy
// fftconvolve(..., "same") calls of ONE giant FFT each, then dsp.py:67-68 ms_to_lr).
// Restructured for the GPU:
//
//  * Block size N = 2F (F = taps = config.fft_size), F fresh output frames per block.  The F-tap
//    filter is treated as F+1 taps with a leading zero (h'[k+1] = h[k]), which turns scipy's
//    "same" offset (F-1)//2 into F/2 and makes every global access aligned: block b reads the N
//    input frames starting at b*F - F/2 (zeros outside the track) and keeps circular outputs
//    [F, 2F) = y[b*F ... (b+1)*F).
//  * A workgroup takes a PAIR of neighbouring blocks (A, B).  For each channel c in (mid, side)
//    it transforms z = c_A + j*c_B with ONE complex FFT; both blocks see the same real filter, so
//    the spectrum is simply multiplied by H_c (no mirror-bin bookkeeping, one complex multiply
//    per bin) and the inverse transform returns y_A in the real and y_B in the imaginary part.
//    dsp.py:57-64 lr_to_ms is applied as the frames are loaded, dsp.py:67-68 ms_to_lr as they are
//    stored; the mid result of both blocks waits in registers while the side channel runs.
//  * The level gain of stages.py:80-88 is folded into H by the filter preparation.
//  * The last forward pass, the multiplication by H and the first inverse pass are fused in
//    registers on one contiguous LDS row per thread (fft2.h).
//
// HBM traffic per output frame: 8 B read (the half-block overlaps and the second channel's
// re-read come from L2) + 8 B write + 4 B for the mid plane the level-correction loop reads
// (stages.py:138-170).
#pragma once

#include "fft2.h"

#if defined(__clang__)
#pragma clang fp contract(fast)        // see fft2.h
#endif

namespace mgx {

// Cache policy of the streaming accesses (aux bits of the buffer instructions: 2 = nt).  The output is
// stored non-temporal: it is not read again by this kernel and must not push the input frames out of
// the L2 before their second use (measured 197 -> 175 us and 70 MB less fetched per 8-minute track).
// Non-temporal or sc0 LOADS were measured slower (198 / 186 us) and stay at the default policy.
constexpr int CONV_STORE_AUX = 2;

struct Conv2Args {
    const float2* x;       // (n,2) interleaved L/R input frames
    long long n;           // frames
    float2* y;             // (n,2) interleaved L/R output frames
    float* ymid;           // (n,) mid of the output, or nullptr
    const float2* h_mid;   // filter spectra, [parts][RL][N/RL] (bin at position row*RL+q stored at q*L+row),
    const float2* h_side;  //   already scaled by gain/N
    const float2* tw;      // exp(-2 pi i k / N), k = 0..N-1
    int parts;             // filter partitions K: taps = K * N/2 (1 = plain overlap-save)
    long long npairs;      // ceil(n / N)
    float* pair_peak;      // [npairs] max(|yL|,|yR|) per pair, or nullptr
    unsigned* queue;       // [8] pairs handed out per XCD beyond the first round, [8] workgroups done; zero between launches
    int run;               // k_conv_delay: blocks per workgroup (npairs counts blocks there)
};

template <int LOG2N>
struct Conv2Block {
    using F = Fft2<LOG2N>;
    static constexpr int N = F::N;
    static constexpr int T = F::T;
    static constexpr int TAPS = N >> 1;               // N = 2F: half of every block is fresh output
    static constexpr int LOUT = N - TAPS;             // fresh output frames per block
    static constexpr int R0 = F::R0;
    static constexpr int RL = F::RL;
    static constexpr int S0 = F::S(0);
    static constexpr int CNT0 = F::CNT(0);
    static constexpr int SKIP = TAPS / S0 > 0 ? TAPS / S0 : 0;   // leading outputs of a butterfly that are circular garbage
    static constexpr int HALF = R0 - SKIP;           // outputs kept per pass-0 butterfly
    static constexpr int BSTEP = LOUT / S0;          // block B = block A advanced by BSTEP butterfly inputs
    static constexpr int NLOAD = R0 + BSTEP;         // frames loaded per pass-0 butterfly
    static_assert((TAPS % S0 == 0 && LOUT % S0 == 0) || F::partial(0), "block geometry must follow the pass-0 stride");

    struct Persist {
        typename F::Tw0 tw0;
    };
    struct Kept {                                     // mid results of the pair: (y_A, y_B)
        float2 v[CNT0][HALF];
    };

    static MGX_HD bool active0(int tid) { return !F::partial(0) || tid < F::NB(0); }

    static MGX_HD void load_persist(int tid, const float2* tw, float2* mid_table, Persist& ps) {
        F::load_tw0(tid, tw, ps.tw0);
        F::fill_mid_table(tid, tw, mid_table);
    }

    // Frames of the pair: block A outputs [pair*N, pair*N + P), block B the next P (P = N/2).
    // With K filter partitions of P taps each (uniformly partitioned overlap-save: taps = K*P,
    // y = sum_k h_k * x delayed by k*P), partition k reads the N + P frames starting at
    //     pair*N + K*P/2 - (k+1)*P
    // (scipy's "same" centring puts half of the filter into the future); K = 1 is the plain case.
    static MGX_HD long long first_output(long long pair) { return pair * (long long)(2 * LOUT); }
    static MGX_HD long long first_input(long long pair, int parts = 1, int k = 0) {
        return first_output(pair) + (long long)parts * TAPS / 2 - (long long)(k + 1) * TAPS;
    }
    // every frame the pair touches lies inside the track
    static MGX_HD bool interior(long long pair, long long n, int parts = 1) {
        return first_input(pair, parts, parts - 1) >= 0 && first_input(pair, parts, 0) + N + LOUT <= n &&
               first_output(pair) + 2 * LOUT <= n;
    }

    // ---- phase F0: global -> registers -> pass 0 -> LDS ----------------------------------------
    // fetch_frames() issues the loads into `raw`, phase_pass0*() consume them.  There is no separate
    // code path for pairs at the ends of the track: frames past its end read as zeros through the
    // buffer range check, and only a window that starts BEFORE the track (the first pair) forms
    // its offsets differently.
    struct Raw {
        float2 f[CNT0][NLOAD];
    };
    static MGX_HD void fetch_frames(int tid, long long pair, const Conv2Args& a, int part, Raw& raw) {
        if (!active0(tid)) return;
        const long long i0 = first_input(pair, a.parts, part);
        MGX_UNROLL
        for (int c = 0; c < CNT0; ++c) {
            const int u = tid + c * T;
            // one 32-bit lane offset (mgx_hd.h MemView); frames outside the track read as zeros
            const MemView src = mem_view(a.x, a.n * 8);
            const unsigned lane = ((unsigned)i0 + (unsigned)u) * 8u;
            if (i0 < 0) {
                // the window starts before the track: the displacement goes into the lane offset, which
                // then wraps to the frame's true offset where there is one
                MGX_UNROLL
                for (int j = 0; j < NLOAD; ++j) raw.f[c][j] = ld_f2_or_zero(src, lane + (unsigned)(j * S0 * 8));
            } else {
                MGX_UNROLL
                for (int j = 0; j < NLOAD; ++j) raw.f[c][j] = ld_f2(src, lane, (unsigned)(j * S0 * 8));   // literal displacements
            }
        }
    }
    // mid/side of the fetched frames (dsp.py:57-64), the two blocks of the pair packed into one
    // complex sequence, pass 0 -> LDS.  The mid pass also forms the side samples and leaves them in
    // `held` (NLOAD*CNT0 floats) for the side pass, which then needs no second read of the frames.
    struct Held {
        float s[CNT0][NLOAD];
    };
    static MGX_HD void phase_pass0_mid(int tid, const Raw& raw, const Persist& ps, float2* lds, Held& held) {
        if (!active0(tid)) return;
        typename F::Tw0Full tw;
        F::expand_tw0(ps.tw0, tw);
        MGX_UNROLL
        for (int c = 0; c < CNT0; ++c) {
            float ch[NLOAD];
            MGX_UNROLL
            for (int j = 0; j < NLOAD; ++j) {
                const float2 f = raw.f[c][j];
                ch[j] = (f.x + f.y) * 0.5f;                      // dsp.py:59-60
                held.s[c][j] = ch[j] - f.y;                      // dsp.py:62
            }
            float2 v[R0];
            MGX_UNROLL
            for (int j = 0; j < R0; ++j) v[j] = make_float2(ch[j], ch[j + BSTEP]);
            F::fwd0_store(v, tid, c, tw, lds);
        }
    }
    static MGX_HD void phase_pass0_side(int tid, const Held& held, const Persist& ps, float2* lds) {
        if (!active0(tid)) return;
        typename F::Tw0Full tw;
        F::expand_tw0(ps.tw0, tw);
        MGX_UNROLL
        for (int c = 0; c < CNT0; ++c) {
            float2 v[R0];
            MGX_UNROLL
            for (int j = 0; j < R0; ++j) v[j] = make_float2(held.s[c][j], held.s[c][j + BSTEP]);
            F::fwd0_store(v, tid, c, tw, lds);
        }
    }
    // one channel straight from the frames (filter partitions: a different window per partition)
    template <bool SIDE>
    static MGX_HD void phase_pass0(int tid, const Raw& raw, const Persist& ps, float2* lds) {
        if (!active0(tid)) return;
        typename F::Tw0Full tw;
        F::expand_tw0(ps.tw0, tw);
        MGX_UNROLL
        for (int c = 0; c < CNT0; ++c) {
            float ch[NLOAD];
            MGX_UNROLL
            for (int j = 0; j < NLOAD; ++j) {
                const float2 f = raw.f[c][j];
                const float m = (f.x + f.y) * 0.5f;
                ch[j] = SIDE ? m - f.y : m;
            }
            float2 v[R0];
            MGX_UNROLL
            for (int j = 0; j < R0; ++j) v[j] = make_float2(ch[j], ch[j + BSTEP]);
            F::fwd0_store(v, tid, c, tw, lds);
        }
    }
    // both at once (filter partitions, tests)
    template <bool SIDE>
    static MGX_HD void phase_load(int tid, long long pair, bool edge, const Conv2Args& a, const Persist& ps,
                                  float2* lds, int part = 0) {
        Raw raw;
        fetch_frames(tid, pair, a, part, raw);
        phase_pass0<SIDE>(tid, raw, ps, lds);
    }

    // ---- middle passes --------------------------------------------------------------------------
    static MGX_HD void phase_fwd_mid(int tid, float2* lds, const float2* mid_table) {
        if (F::P >= 3) F::fwd_mid(tid, lds, mid_table);
    }
    // the second middle pass of a four-pass plan (a barrier apart from the first)
    static MGX_HD void phase_fwd_mid2(int tid, float2* lds, const float2* mid_table) {
        if (F::P == 4) F::fwd_mid2(tid, lds, mid_table);
    }
    static MGX_HD void phase_inv_mid2(int tid, float2* lds, const float2* mid_table) {
        if (F::P == 4) F::inv_mid2(tid, lds, mid_table);
    }
    static MGX_HD void phase_inv_mid(int tid, float2* lds, const float2* mid_table) {
        if (F::P >= 3) F::inv_mid(tid, lds, mid_table);
    }

    // ---- phase FPI: last forward pass, times H, first inverse pass, on the thread's row -------
    // The filter spectrum of the row is fetched one phase early (before the middle pass and its
    // barrier) so that its L2 latency is covered by that pass.
    struct RowFilter {
        float2 h[RL];
    };
    static MGX_HD void fetch_filter(int tid, const float2* h, RowFilter& f) {
        if (!F::has_row(tid)) return;
        const MemView hv = mem_view(h, (long long)N * 8);
        MGX_UNROLL
        for (int q = 0; q < RL; ++q) f.h[q] = ld_f2(hv, (unsigned)tid * 8u, (unsigned)(q * F::L * 8));
    }
    static MGX_HD void phase_filter(int tid, const RowFilter& f, float2* lds) {
        if (!F::has_row(tid)) return;
        float2 v[RL];
        F::load_row(v, tid, lds);
        dft_regs<RL, false>(v);
        MGX_UNROLL
        for (int q = 0; q < RL; ++q) {
            const int i = bitrev(q, F::lr(F::LAST));
            v[i] = cmul(v[i], f.h[q]);
        }
        dft_regs<RL, true>(v);
        F::store_row(v, tid, lds);
    }

    // K > 1: the row's product is accumulated over the partitions (phase_accumulate once per
    // partition, then phase_finish_row)
    struct RowAcc {
        float2 w[RL];
    };
    static MGX_HD void clear_acc(RowAcc& acc) {
        MGX_UNROLL
        for (int q = 0; q < RL; ++q) acc.w[q] = make_float2(0.f, 0.f);
    }
    static MGX_HD void phase_accumulate(int tid, const RowFilter& f, const float2* lds, RowAcc& acc) {
        if (!F::has_row(tid)) return;
        float2 v[RL];
        F::load_row(v, tid, lds);
        dft_regs<RL, false>(v);
        MGX_UNROLL
        for (int q = 0; q < RL; ++q) {
            const int i = bitrev(q, F::lr(F::LAST));
            acc.w[i] = cadd(acc.w[i], cmul(v[i], f.h[q]));
        }
    }
    static MGX_HD void phase_finish_row(int tid, RowAcc& acc, float2* lds) {
        if (!F::has_row(tid)) return;
        dft_regs<RL, true>(acc.w);
        F::store_row(acc.w, tid, lds);
    }

    // ---- phase I0 (mid channel): inverse pass 0, keep the valid half in registers --------------
    static MGX_HD void phase_keep_mid(int tid, const Persist& ps, const float2* lds, Kept& k) {
        if (!active0(tid)) return;
        typename F::Tw0Full tw;
        F::expand_tw0(ps.tw0, tw);
        MGX_UNROLL
        for (int c = 0; c < CNT0; ++c) {
            float2 v[R0];
            F::inv0_load(v, tid, c, tw, lds);
            MGX_UNROLL
            for (int j = 0; j < HALF; ++j) k.v[c][j] = v[SKIP + j];
        }
    }

    // ---- phase I0 (side channel) + epilogue: L = mid + side, R = mid - side (dsp.py:67-68) ----
    // returns this thread's max(|yL|,|yR|) over the frames it stored;  One code path for every pair:
    // stores beyond the end of the track are dropped by the buffer range check, only the peak needs
    // to know which frames exist.
    static MGX_HD float phase_store(int tid, long long pair, bool edge, const Conv2Args& a, const Persist& ps,
                                    const float2* lds, const Kept& k) {
        float peak = 0.f;
        if (!active0(tid)) return peak;
        const long long o0 = first_output(pair);
        typename F::Tw0Full tw;
        F::expand_tw0(ps.tw0, tw);
        const MemView dst = mem_view(a.y, a.n * 8);
        const MemView dm = mem_view(a.ymid, a.ymid ? a.n * 4 : 0);
        const unsigned frames = (unsigned)a.n;
        MGX_UNROLL
        for (int c = 0; c < CNT0; ++c) {
            const unsigned first = (unsigned)o0 + (unsigned)(tid + c * T);
            float2 v[R0];
            F::inv0_load(v, tid, c, tw, lds);
            MGX_UNROLL
            for (int j = 0; j < HALF; ++j) {
                const float2 m = k.v[c][j], s = v[SKIP + j];
                const float2 ya = make_float2(m.x + s.x, m.x - s.x);
                const float2 yb = make_float2(m.y + s.y, m.y - s.y);
                const unsigned fa = first + (unsigned)(j * S0), fb = fa + (unsigned)LOUT;
                // output offsets are never negative, so the literal displacement may stay in the scalar
                // operand: the range check adds it to the lane offset (tools/micro/buffer_range.hip)
                st_f2<CONV_STORE_AUX>(dst, first * 8u, (unsigned)(j * S0 * 8), ya);
                st_f2<CONV_STORE_AUX>(dst, first * 8u, (unsigned)((j * S0 + LOUT) * 8), yb);
                st_f1<CONV_STORE_AUX>(dm, first * 4u, (unsigned)(j * S0 * 4), m.x);
                st_f1<CONV_STORE_AUX>(dm, first * 4u, (unsigned)((j * S0 + LOUT) * 4), m.y);
                const float pa = fmaxf(fabsf(ya.x), fabsf(ya.y)), pb = fmaxf(fabsf(yb.x), fabsf(yb.y));
                peak = fmaxf(peak, fmaxf(!edge || fa < frames ? pa : 0.f, !edge || fb < frames ? pb : 0.f));
            }
        }
        return peak;
    }

    // ---- filter preparation (one workgroup per channel, once per track) ------------------------
    // taps[0..F) -> zero-extended, delayed by one sample -> pass 0 -> LDS
    static MGX_HD void phase_load_taps(int tid, const float* taps, const Persist& ps, float2* lds) {
        if (!active0(tid)) return;
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
    // last pass of the filter transform -> table in the layout phase_filter reads
    static MGX_HD void phase_write_filter(int tid, const float2* lds, float scale, float2* h) {
        if (!F::has_row(tid)) return;
        float2 v[RL];
        F::load_row(v, tid, lds);
        dft_regs<RL, false>(v);
        MGX_UNROLL
        for (int q = 0; q < RL; ++q) {
            const float2 x = v[bitrev(q, F::lr(F::LAST))];
            h[q * F::L + tid] = make_float2(x.x * scale, x.y * scale);
        }
    }
};

}  // namespace mgx

#if defined(__clang__)
#pragma clang fp contract(off)
#endif
