// One-pass level + spectrum statistics of a track.
//
// Replaces, for one track, matchering/stage_helpers/match_levels.py:134-161
// (`analyze_levels`: lr_to_ms, piece RMS, loudest-piece extraction), the peak
// scan of dsp.py:93-100 (`normalize`) and the four scipy.signal.stft calls of
// match_frequencies.py:30-42 (`__average_fft`).  The reference first selects the
// loud pieces, copies them and then transforms them; here every piece is
// transformed while it streams through once (8 B/frame read, nothing written but
// per-workgroup partials) and the selection is applied afterwards to the
// statistics, which is exact because every statistic is a plain sum over frames
// or segments (validated restructuring (i)/(ii)).
//
// Work decomposition: piece d covers frames [d*p, (d+1)*p); its first q = p // F
// segments of F frames are transformed.  A workgroup takes SEGS consecutive
// segments of one piece; the workgroup with the piece's last chunk also sums the
// < F leftover frames of the piece (they count for RMS, not for the spectrum),
// and the very last workgroup scans the ignored tail [D*p, n) for the peak.
//
// Per segment: z = mid + j*side, one complex FFT_F (two-for-one), then
// |M_k| = |Z_k + conj Z_{F-k}|/2 and |S_k| = |Z_k - conj Z_{F-k}|/2 for
// k = 0..F/2, accumulated in registers across the workgroup's segments (each
// thread owns the same mirror pair of butterflies in every segment).
#pragma once

#include "fft_core.h"

namespace mgx {

struct AnalysisArgs {
    const float2* x;        // (n,2) interleaved
    long long n;
    int fft;                // F
    long long piece;        // p
    int divisions;          // D
    int segs_per_piece;     // q = p // F
    int segs_per_wg;        // SEGS
    int chunks_per_piece;   // ceil(q / SEGS)
    // outputs, one slot per workgroup
    double* wg_sumsq;       // sum of mid^2 over the frames this workgroup owns
    float* wg_peak;         // max(|L|,|R|) over the frames this workgroup owns
    float* wg_spec;         // [wg][2][F/2+1] sum over segments of |M_k|, |S_k| (unscaled)
    const float2* tw;
};

template <int LOG2N>
struct AnalysisBlock {
    using F = Fft<LOG2N>;
    static constexpr int N = F::N;
    static constexpr int T = F::T;
    static constexpr int P = F::P;
    static constexpr int LAST = F::LAST;
    static constexpr int RL = F::R(LAST);
    static constexpr int LB = F::lr(LAST);
    static constexpr int L = N / RL;
    static constexpr int ITEMS = L / 2;
    static constexpr int ITEM_CNT = ITEMS / T > 0 ? ITEMS / T : 1;

    struct Thread {
        double sumsq;
        float peak;
        float acc_mid[ITEM_CNT][2 * RL];   // magnitudes of the bins this thread owns
        float acc_side[ITEM_CNT][2 * RL];
    };

    static MGX_HD void init(Thread& t) {
        t.sumsq = 0.0;
        t.peak = 0.f;
        MGX_UNROLL
        for (int i = 0; i < ITEM_CNT; ++i) {
            MGX_UNROLL
            for (int q = 0; q < 2 * RL; ++q) { t.acc_mid[i][q] = 0.f; t.acc_side[i][q] = 0.f; }
        }
    }

    static MGX_HD void to_ms(float2 lr, float& m, float& s) {
        m = (lr.x + lr.y) * 0.5f;      // dsp.py:59-60
        s = m - lr.y;                  // dsp.py:62
    }

    // segment starting at frame `start` (always fully inside the track)
    static MGX_HD void phase_load(int tid, long long start, const AnalysisArgs& a, Thread& t,
                                  float2* lds) {
        MGX_UNROLL
        for (int i = 0; i < F::CNT(0); ++i) {
            const int u = tid + i * T;
            if (u < F::NB(0)) {
                float2 v[F::R(0)];
                float ss = 0.f;
                MGX_UNROLL
                for (int j = 0; j < F::R(0); ++j) {
                    const float2 lr = a.x[start + u + (long long)j * F::S(0)];
                    float m, s;
                    to_ms(lr, m, s);
                    v[j] = make_float2(m, s);
                    ss = fmaf(m, m, ss);
                    t.peak = fmaxf(t.peak, fmaxf(fabsf(lr.x), fabsf(lr.y)));
                }
                t.sumsq += (double)ss;
                F::template fwd_store<0>(v, u, lds, a.tw);
            }
        }
    }

    static MGX_HD void phase_fwd_mid(int tid, float2* lds, const float2* tw) {
        if (P == 3) F::template fwd_pass_lds<(P == 3 ? 1 : 0)>(tid, lds, tw);
    }

    // last forward pass + magnitude accumulation
    static MGX_HD void phase_magnitudes(int tid, Thread& t, const float2* lds) {
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
            MGX_UNROLL
            for (int q = 0; q < RL; ++q) {
                // slot q: bin of A (k = ka + L*q) against its mirror; slot RL+q: same for B
                float2 za, zam, zb, zbm;
                if (it != 0) {
                    za = va[bitrev(q, LB)];
                    zam = vb[bitrev(RL - 1 - q, LB)];
                    zb = za;  zbm = zam;               // B's bins are A's mirrors: nothing new
                } else {
                    za = va[bitrev(q, LB)];
                    zam = va[bitrev((RL - q) % RL, LB)];
                    zb = vb[bitrev(q, LB)];
                    zbm = vb[bitrev(RL - 1 - q, LB)];
                }
                // M = (Z + conj Zm)/2, S = (Z - conj Zm)/(2j): |.| only
                {
                    const float mx = za.x + zam.x, my = za.y - zam.y;
                    const float sx = za.x - zam.x, sy = za.y + zam.y;
                    t.acc_mid[i][q] += 0.5f * sqrtf(fmaf(mx, mx, my * my));
                    t.acc_side[i][q] += 0.5f * sqrtf(fmaf(sx, sx, sy * sy));
                }
                if (it == 0) {
                    const float mx = zb.x + zbm.x, my = zb.y - zbm.y;
                    const float sx = zb.x - zbm.x, sy = zb.y + zbm.y;
                    t.acc_mid[i][RL + q] += 0.5f * sqrtf(fmaf(mx, mx, my * my));
                    t.acc_side[i][RL + q] += 0.5f * sqrtf(fmaf(sx, sx, sy * sy));
                }
            }
        }
    }

    // frames outside whole segments: RMS (optional) and peak only
    static MGX_HD void phase_loose_frames(int tid, long long begin, long long end, bool count_rms,
                                          const AnalysisArgs& a, Thread& t) {
        for (long long f = begin + tid; f < end; f += T) {
            const float2 lr = a.x[f];
            float m, s;
            to_ms(lr, m, s);
            if (count_rms) t.sumsq += (double)(m * m);
            t.peak = fmaxf(t.peak, fmaxf(fabsf(lr.x), fabsf(lr.y)));
        }
    }

    // write this thread's spectrum sums: every bin k in [0, F/2] is owned by exactly one slot
    static MGX_HD void phase_write_spectrum(int tid, int wg, const AnalysisArgs& a, const Thread& t) {
        const int half = N / 2;
        float* mid = a.wg_spec + (size_t)wg * 2 * (half + 1);
        float* side = mid + (half + 1);
        MGX_UNROLL
        for (int i = 0; i < ITEM_CNT; ++i) {
            const int it = tid + i * T;
            if (it >= ITEMS) continue;
            const int ka = it == 0 ? 0 : it;
            MGX_UNROLL
            for (int q = 0; q < RL; ++q) {
                const int k = ka + L * q;
                if (it != 0) {
                    const int kk = k <= half ? k : N - k;          // |X_k| = |X_{N-k}| for real input
                    mid[kk] = t.acc_mid[i][q];
                    side[kk] = t.acc_side[i][q];
                } else {
                    if (k <= half) { mid[k] = t.acc_mid[i][q]; side[k] = t.acc_side[i][q]; }
                    const int k2 = L / 2 + L * q;
                    if (k2 <= half) { mid[k2] = t.acc_mid[i][RL + q]; side[k2] = t.acc_side[i][RL + q]; }
                }
            }
        }
    }
};

// ---------------------------------------------------------------------------
// Scalar epilogue of the analysis (one workgroup, thread 0 does the decisions).
// match_levels.py:62-71,93-103 and the mean of match_frequencies.py:42.
// ---------------------------------------------------------------------------
struct TrackStats {
    double peak;             // max |x| over the whole track (dsp.py:97)
    double amplitude_c;      // reference only: final_amplitude_coefficient (dsp.py:93-100), else 1
    double average_rms;
    double match_rms;        // of the (normalised) track
    int divisions;
    int loud_count;
    long long piece;
};

// rms[d] and loud[d] are outputs of size `divisions`
MGX_HD void finish_levels(const double* wg_sumsq, const float* wg_peak, int chunks_per_piece,
                          int divisions, long long piece, bool is_reference, double threshold,
                          double eps, double* rms, int* loud, TrackStats& st) {
    double peak = 0.0;
    for (int w = 0; w < divisions * chunks_per_piece; ++w) peak = fmax(peak, (double)wg_peak[w]);
    double c = 1.0;
    if (is_reference && peak < threshold) c = fmax(eps, peak / threshold);   // dsp.py:98-99
    double mean_sq = 0.0;
    for (int d = 0; d < divisions; ++d) {
        double s = 0.0;
        for (int ch = 0; ch < chunks_per_piece; ++ch) s += wg_sumsq[d * chunks_per_piece + ch];
        rms[d] = sqrt(s / (double)piece) / c;                 // dsp.py:80-86 on x/c
        mean_sq += rms[d] * rms[d];
    }
    const double avg = sqrt(mean_sq / divisions);             // dsp.py:76-77
    double loud_sq = 0.0;
    int cnt = 0;
    for (int d = 0; d < divisions; ++d) {
        loud[d] = rms[d] >= avg;                              // match_levels.py:65
        if (loud[d]) { loud_sq += rms[d] * rms[d]; ++cnt; }
    }
    st.peak = peak;
    st.amplitude_c = c;
    st.average_rms = avg;
    st.match_rms = sqrt(loud_sq / cnt);                       // match_levels.py:67-68
    st.divisions = divisions;
    st.loud_count = cnt;
    st.piece = piece;
}

}  // namespace mgx
