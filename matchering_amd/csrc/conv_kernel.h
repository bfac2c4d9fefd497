// Matching-EQ FIR convolution: segmented overlap-save in LDS.
//
// Replaces matchering/stage_helpers/match_frequencies.py:104-119 (`convolve`:
// two scipy fftconvolve(..., "same") calls of ONE giant FFT each, then
// dsp.py:67-68 ms_to_lr).  Restructured for the GPU:
//
//  * The mid/side filters are folded into a 2x2 filter on the interleaved L/R
//    frames themselves.  With z = L + jR one complex FFT of a block gives Z, and
//        W_k = a_k Z_k + j c_k conj(Z_{B-k}),   a = FFT((h_mid+h_side)/2), c = FFT((h_mid-h_side)/2)
//    is the spectrum of (y_L + j y_R): no mid/side pass over the audio at all
//    (dsp.py:57-64 lr_to_ms and dsp.py:67-68 ms_to_lr are absorbed), and the
//    level gain of stages.py:80-88 is folded into the taps by the caller.
//  * Overlap-save with block B = 2^LOG2N >= 2F: block b loads B input frames
//    starting at b*Lout + off - (F-1) (zero outside the track), Lout = B-F+1,
//    off = (F-1)//2 = scipy's "same" centring, and keeps circular outputs
//    [F-1, B) = y[b*Lout ... (b+1)*Lout).
//  * Forward DIF passes leave the spectrum in position order; the last forward
//    pass, the pointwise product and the first inverse pass are fused in
//    registers: the thread that owns the radix-R butterfly with low index k0 also
//    owns the one with B/R - k0, which holds exactly the mirror bins B-k.
//
// HBM traffic per output frame: 8 B read (+ overlap, served by L2) + 8 B write
// (+ 4 B optional mid plane for the level-correction loop, stages.py:138-170).
#pragma once

#include "fft_core.h"

namespace mgx {

struct ConvArgs {
    const float2* x;       // (n,2) interleaved L/R input frames
    long long n;           // frames
    float2* y;             // (n,2) interleaved L/R output frames
    float* ymid;           // (n,) (yL+yR)/2 or nullptr
    const float2* fa;      // filter a, position order, already scaled by 1/B
    const float2* fc;      // filter c, position order, already scaled by 1/B
    const float2* tw;      // exp(-2 pi i k / B), k = 0..B-1
    int taps;              // F
    long long nblocks;
    float* block_peak;     // [nblocks] max(|yL|,|yR|) per block, or nullptr
};

template <int LOG2N>
struct ConvBlock {
    using F = Fft<LOG2N>;
    static constexpr int N = F::N;
    static constexpr int T = F::T;
    static constexpr int P = F::P;
    static constexpr int LAST = F::LAST;
    static constexpr int RL = F::R(LAST);              // radix of the last pass
    static constexpr int LB = F::lr(LAST);
    static constexpr int L = N / RL;                   // number of last-pass butterflies
    static constexpr int ITEMS = L / 2;                // mirror pairs
    static constexpr int ITEM_CNT = ITEMS / T > 0 ? ITEMS / T : 1;

    static MGX_HD long long lout(int taps) { return (long long)N - taps + 1; }
    static MGX_HD long long first_input(long long blk, int taps) {
        return blk * lout(taps) + (taps - 1) / 2 - (taps - 1);
    }

    // ---- phase A: global -> registers -> pass 0 -> LDS ------------------------
    static MGX_HD void phase_load(int tid, long long blk, const ConvArgs& a, float2* lds) {
        const long long i0 = first_input(blk, a.taps);
        MGX_UNROLL
        for (int i = 0; i < F::CNT(0); ++i) {
            const int u = tid + i * T;
            if (u < F::NB(0)) {
                float2 v[F::R(0)];
                MGX_UNROLL
                for (int j = 0; j < F::R(0); ++j) {
                    const long long gi = i0 + u + (long long)j * F::S(0);
                    v[j] = (gi >= 0 && gi < a.n) ? a.x[gi] : make_float2(0.f, 0.f);
                }
                F::template fwd_store<0>(v, u, lds, a.tw);
            }
        }
    }

    // ---- phase B/D: middle passes ---------------------------------------------
    static MGX_HD void phase_fwd_mid(int tid, float2* lds, const float2* tw) {
        if (P == 3) F::template fwd_pass_lds<(P == 3 ? 1 : 0)>(tid, lds, tw);
    }
    static MGX_HD void phase_inv_mid(int tid, float2* lds, const float2* tw) {
        if (P == 3) F::template inv_pass_lds<(P == 3 ? 1 : 0)>(tid, lds, tw);
    }

    // ---- phase C: last forward pass + pointwise 2x2 filter + first inverse pass --
    static MGX_HD void phase_pointwise(int tid, const ConvArgs& a, float2* lds) {
        MGX_UNROLL
        for (int i = 0; i < ITEM_CNT; ++i) {
            const int it = tid + i * T;
            if (it >= ITEMS) continue;
            const int ka = it == 0 ? 0 : it;
            const int kb = it == 0 ? L / 2 : L - it;
            const int ua = F::position_of(ka) / RL, ub = F::position_of(kb) / RL;
            float2 va[RL], vb[RL];
            F::template load_natural<LAST>(va, ua, lds);
            F::template load_natural<LAST>(vb, ub, lds);
            dft_regs<RL, false>(va);
            dft_regs<RL, false>(vb);
            const float2* fa_a = a.fa + ua * RL;
            const float2* fc_a = a.fc + ua * RL;
            if (it != 0) {
                // bins k = ka + L*q in A mirror bins B-k = kb + L*(RL-1-q) in B
                MGX_UNROLL
                for (int q = 0; q < RL; ++q) {
                    const int ia = bitrev(q, LB), ib = bitrev(RL - 1 - q, LB);
                    const float2 fa = fa_a[q], fc = fc_a[q];
                    const float2 za = va[ia], zb = vb[ib];
                    // WA = fa*za + j*fc*conj(zb) ; WB = conj(fa)*zb + j*conj(fc)*conj(za)
                    const float2 ta = cmulc(fc, zb);            // fc*conj(zb)
                    const float2 tb = cconj(cmul(fc, za));      // conj(fc)*conj(za)
                    va[ia] = cadd(cmul(fa, za), cmul_i(ta));
                    vb[ib] = cadd(cmulc(zb, fa), cmul_i(tb));
                }
            } else {
                // A: k0 = 0, mirror of q is (RL-q)%RL inside A.  B: k0 = L/2, mirror RL-1-q inside B.
                const float2* fa_b = a.fa + ub * RL;
                const float2* fc_b = a.fc + ub * RL;
                float2 wa[RL], wb[RL];
                MGX_UNROLL
                for (int q = 0; q < RL; ++q) {
                    const int ia = bitrev(q, LB), ja = bitrev((RL - q) % RL, LB);
                    const int ib = bitrev(q, LB), jb = bitrev(RL - 1 - q, LB);
                    wa[ia] = cadd(cmul(fa_a[q], va[ia]), cmul_i(cmulc(fc_a[q], va[ja])));
                    wb[ib] = cadd(cmul(fa_b[q], vb[ib]), cmul_i(cmulc(fc_b[q], vb[jb])));
                }
                MGX_UNROLL
                for (int q = 0; q < RL; ++q) { va[q] = wa[q]; vb[q] = wb[q]; }
            }
            dft_regs<RL, true>(va);
            dft_regs<RL, true>(vb);
            F::template store_natural<LAST>(va, ua, lds);
            F::template store_natural<LAST>(vb, ub, lds);
        }
    }

    // ---- phase E: inverse pass 0 + epilogue -------------------------------------
    // returns this thread's max(|yL|,|yR|) over the frames it stored
    static MGX_HD float phase_store(int tid, long long blk, const ConvArgs& a, const float2* lds) {
        const long long n0 = blk * lout(a.taps);
        float peak = 0.f;
        MGX_UNROLL
        for (int i = 0; i < F::CNT(0); ++i) {
            const int u = tid + i * T;
            if (u < F::NB(0)) {
                float2 v[F::R(0)];
                F::template inv_load<0>(v, u, lds, a.tw);
                MGX_UNROLL
                for (int j = 0; j < F::R(0); ++j) {
                    const int ci = u + j * F::S(0);              // circular output index
                    const long long n = n0 + ci - (a.taps - 1);
                    if (ci >= a.taps - 1 && n < a.n) {
                        a.y[n] = v[j];
                        if (a.ymid) a.ymid[n] = 0.5f * (v[j].x + v[j].y);
                        peak = fmaxf(peak, fmaxf(fabsf(v[j].x), fabsf(v[j].y)));
                    }
                }
            }
        }
        return peak;
    }

    // ---- FIR spectra in position order (run once per track by one workgroup) -----
    // lds holds FFT_B(ha + j*hc) in position order; writes fa, fc scaled by `scale`.
    static MGX_HD void phase_split_filters(int tid, const float2* lds, float2* fa, float2* fc,
                                           float scale) {
        for (int posn = tid; posn < N; posn += T) {
            const int k = F::frequency_at(posn);
            const int mirror = F::position_of((N - k) & (N - 1));
            const float2 z = lds[F::pad(posn)], zm = cconj(lds[F::pad(mirror)]);
            // A = (Z_k + conj Z_{-k})/2,  C = (Z_k - conj Z_{-k})/(2j)
            const float2 s = cadd(z, zm), d = csub(z, zm);
            fa[posn] = make_float2(0.5f * scale * s.x, 0.5f * scale * s.y);
            fc[posn] = make_float2(0.5f * scale * d.y, -0.5f * scale * d.x);
        }
    }
    static MGX_HD void phase_load_taps(int tid, const float* h_mid, const float* h_side, int taps,
                                       float2* lds, const float2* tw) {
        MGX_UNROLL
        for (int i = 0; i < F::CNT(0); ++i) {
            const int u = tid + i * T;
            if (u < F::NB(0)) {
                float2 v[F::R(0)];
                MGX_UNROLL
                for (int j = 0; j < F::R(0); ++j) {
                    const int t = u + j * F::S(0);
                    const float m = t < taps ? h_mid[t] : 0.f, s = t < taps ? h_side[t] : 0.f;
                    v[j] = make_float2(0.5f * (m + s), 0.5f * (m - s));
                }
                F::template fwd_store<0>(v, u, lds, tw);
            }
        }
    }
};

}  // namespace mgx
