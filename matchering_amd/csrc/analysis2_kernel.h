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
// Per segment: z = mid + j*side, one complex FFT_F (two-for-one, fft2.h).  The last pass
// leaves each thread with one row of RL bins, which it writes back to LDS in position order;
// after a barrier it reads the upper half of the row holding the mirror bins F-k and accumulates
// |M_k| = |Z_k + conj Z_{F-k}|/2 and |S_k| = |Z_k - conj Z_{F-k}|/2 for the LOWER half of its own
// row in registers across the workgroup's segments.  Bin (row, q) mirrors to
// (mirror_row, RL-1-q), so the lower halves of all rows cover every pair {k, F-k} exactly once
// (|X_k| = |X_{F-k}| for the real signals mid and side); the self-mirrored bin F/2 = (0, RL/2) is
// one extra accumulator of thread 0.
#pragma once

#include "fft2.h"

namespace mgx {

struct AnalysisArgs {
    const float2* x;        // (n,2) interleaved
    long long n;
    int fft;                // F
    long long piece;        // p
    int divisions;          // D
    int segs_per_piece;     // q = p // F
    int chunks_per_piece;   // workgroups per piece: chunk c takes segments [c*q/C, (c+1)*q/C)
    // outputs, one slot per workgroup
    double* wg_sumsq;       // sum of mid^2 over the frames this workgroup owns
    float* wg_peak;         // max(|L|,|R|) over the frames this workgroup owns
    float* wg_spec;         // [wg][2][F/2+1] sum over segments of |M_k|, |S_k| (unscaled)
    const float2* tw;
    float2* wg_pack;        // fft_size 65536 only (AnalysisQuad): [wg][AnalysisQuad::SCRATCH_FLOAT2] scratch
};

template <int LOG2N>
struct Analysis2Block {
    using F = Fft2<LOG2N>;
    static constexpr int N = F::N;
    static constexpr int T = F::T;
    static constexpr int R0 = F::R0;
    static constexpr int RL = F::RL;
    static constexpr int S0 = F::S(0);
    static constexpr int CNT0 = F::CNT(0);

    struct Persist {
        typename F::Tw0 tw0;
    };
    struct Thread {
        double sumsq;
        float peak;
        float acc_mid[RL / 2 + 1];     // magnitudes of bins (row, q < RL/2); [RL/2]: bin F/2 on thread 0
        float acc_side[RL / 2 + 1];
    };

    static MGX_HD bool active0(int tid) { return !F::partial(0) || tid < F::NB(0); }

    // segments of chunk `ch` of a piece: the q segments are dealt out as evenly as integers allow
    static MGX_HD void chunk_segments(const AnalysisArgs& a, int ch, int& s0, int& s1) {
        s0 = (int)((long long)ch * a.segs_per_piece / a.chunks_per_piece);
        s1 = (int)((long long)(ch + 1) * a.segs_per_piece / a.chunks_per_piece);
    }

    static MGX_HD void load_persist(int tid, const float2* tw, float2* mid_table, Persist& ps) {
        F::load_tw0(tid, tw, ps.tw0);
        F::fill_mid_table(tid, tw, mid_table);
    }
    static MGX_HD void init(Thread& t) {
        t.sumsq = 0.0;
        t.peak = 0.f;
        MGX_UNROLL
        for (int q = 0; q <= RL / 2; ++q) { t.acc_mid[q] = 0.f; t.acc_side[q] = 0.f; }
    }
    static MGX_HD void to_ms(float2 lr, float& m, float& s) {
        m = (lr.x + lr.y) * 0.5f;      // dsp.py:59-60
        s = m - lr.y;                  // dsp.py:62
    }

    // frames of one segment as a thread needs them for pass 0 (segments always lie inside the track)
    struct Raw {
        float2 f[CNT0][R0];
    };
    static MGX_HD void fetch(int tid, long long start, const AnalysisArgs& a, Raw& r) {
        if (!active0(tid)) return;
        const MemView src = mem_view(a.x, a.n * 8);
        MGX_UNROLL
        for (int c = 0; c < CNT0; ++c) {
            const unsigned lane = ((unsigned)start + (unsigned)(tid + c * T)) * 8u;
            MGX_UNROLL
            for (int j = 0; j < R0; ++j) r.f[c][j] = ld_f2(src, lane, (unsigned)(j * S0 * 8));
        }
    }
    // mid/side, level statistics, pass 0 -> LDS
    static MGX_HD void phase_load(int tid, const Raw& r, const Persist& ps, Thread& t, float2* lds) {
        if (!active0(tid)) return;
        typename F::Tw0Full tw;
        F::expand_tw0(ps.tw0, tw);
        MGX_UNROLL
        for (int c = 0; c < CNT0; ++c) {
            float2 v[R0];
            float ss = 0.f;
            MGX_UNROLL
            for (int j = 0; j < R0; ++j) {
                const float2 lr = r.f[c][j];
                float m, s;
                to_ms(lr, m, s);
                v[j] = make_float2(m, s);
                ss = fmaf(m, m, ss);
                t.peak = fmaxf(t.peak, fmaxf(fabsf(lr.x), fabsf(lr.y)));
            }
            t.sumsq += (double)ss;
            F::fwd0_store(v, tid, c, tw, lds);
        }
    }
    static MGX_HD void phase_fwd_mid(int tid, float2* lds, const float2* mid_table) {
        if (F::P >= 3) F::fwd_mid(tid, lds, mid_table);
    }
    static MGX_HD void phase_fwd_mid2(int tid, float2* lds, const float2* mid_table) {
        if (F::P == 4) F::fwd_mid2(tid, lds, mid_table);
    }
    // last forward pass on the thread's row.  Only the UPPER half of the row goes back to LDS (position order): that is
    // what the thread of the mirror row reads; the lower half -- and element RL/2, the self-mirrored bin on row 0 --
    // waits in `own` for phase_magnitudes.  (Until round 3 the whole row was written and the lower half read back:
    // eight 16-byte stores and five loads per thread and segment that nobody else needed, and LDS stores are the
    // slowest thing this kernel does -- 13 cycles per ds_write_b128, MI355X_MICROARCH.md.)
    struct Row {
        float2 z[RL / 2 + 1];
    };
    static MGX_HD void phase_row(int tid, Row& own, float2* lds) {
        if (!F::has_row(tid)) return;
        float2 v[RL], w[RL];
        F::load_row(v, tid, lds);
        dft_regs<RL, false>(v);
        MGX_UNROLL
        for (int q = 0; q < RL; ++q) w[q] = v[bitrev(q, F::lr(F::LAST))];
        F::template store_row_part<RL / 2, RL / 2>(w, tid, lds);
        MGX_UNROLL
        for (int q = 0; q <= RL / 2; ++q) own.z[q] = w[q];
    }
    // own lower half + mirror row's upper half -> magnitudes.  Bin (row, q) mirrors to
    // (mirror_row, RL-1-q); row 0 mirrors into itself, (0, q) <-> (0, (RL-q) % RL), i.e. one
    // element further up, with bins 0 and F/2 their own mirrors.
    static MGX_HD void phase_magnitudes(int tid, const Row& own, Thread& t, const float2* lds) {
        if (!F::has_row(tid)) return;
        float2 m[RL / 2];
        F::template load_row_part<RL / 2, RL / 2>(m, F::mirror_row(tid), lds);  // mirror row, elements RL/2 .. RL-1
        const float2 (&z)[RL / 2 + 1] = own.z;
        const bool r0 = tid == 0;
        MGX_UNROLL
        for (int q = 0; q < RL / 2; ++q) {
            const float2 a = m[RL / 2 - 1 - q];                                  // element RL-1-q
            const float2 b = q == 0 ? z[0] : m[RL / 2 - q];                      // element (RL-q) % RL of row 0
            const float2 zm = make_float2(r0 ? b.x : a.x, r0 ? b.y : a.y);
            // M = (Z + conj Zm)/2, S = (Z - conj Zm)/(2j): |.| only
            const float mx = z[q].x + zm.x, my = z[q].y - zm.y;
            const float sx = z[q].x - zm.x, sy = z[q].y + zm.y;
            t.acc_mid[q] += 0.5f * fast_sqrt(fmaf(mx, mx, my * my));
            t.acc_side[q] += 0.5f * fast_sqrt(fmaf(sx, sx, sy * sy));
        }
        // bin F/2 mirrors into itself: M = Re Z, S = Im Z (meaningful on thread 0 only)
        t.acc_mid[RL / 2] += fabsf(z[RL / 2].x);
        t.acc_side[RL / 2] += fabsf(z[RL / 2].y);
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

    // write this thread's spectrum sums: the pair {k, F-k} of bin (row, q < RL/2) goes to min(k, F-k)
    static MGX_HD void phase_write_spectrum(int tid, int wg, const AnalysisArgs& a, const Thread& t) {
        if (!F::has_row(tid)) return;
        const int half = N / 2;
        float* mid = a.wg_spec + (size_t)wg * 2 * (half + 1);
        float* side = mid + (half + 1);
        const int k0 = F::frequency_at(tid * RL);                // low digits of the row's bins
        MGX_UNROLL
        for (int q = 0; q < RL / 2; ++q) {
            const int k = k0 + q * F::L;
            const int kk = k <= half ? k : N - k;
            mid[kk] = t.acc_mid[q];
            side[kk] = t.acc_side[q];
        }
        if (tid == 0) { mid[half] = t.acc_mid[RL / 2]; side[half] = t.acc_side[RL / 2]; }
    }
};

// ---------------------------------------------------------------------------
// fft_size = 2 N', N' = the longest transform a workgroup's LDS holds: a segment's spectra from two
// transforms of N' points instead of one of 2 N'.  A REAL sequence r of 2 N' points packs as
// y[n] = r[2n] + j r[2n+1]; with Y = FFT_N'(y), A = (Y_k + conj Y_{N'-k})/2 and B = (Y_k - conj Y_{N'-k})/(2j)
// are the spectra of the even and of the odd samples, and
//     R_k = A + w^k B,   R_{N'-k} = conj(A - w^k B),   w = exp(-j pi / N'),   k = 0 .. N'/2
// (R_{N'} comes out of k = 0).  A and B are exactly what the two-for-one code above forms from a bin and its
// mirror, so a segment is: transform of the mid samples, magnitudes |A +- w^k B| into the accumulators of
// bins k and N'-k; the same for the side samples.  Frames are read twice (the second time from the L2).
template <int LOG2H>
struct AnalysisDouble {
    using AB = Analysis2Block<LOG2H>;
    using F = Fft2<LOG2H>;
    static constexpr int N = F::N;                 // N' (complex points per transform); the segment has 2 N frames
    static constexpr int T = F::T;
    static constexpr int R0 = F::R0;
    static constexpr int RL = F::RL;
    static constexpr int S0 = F::S(0);
    static constexpr int CNT0 = F::CNT(0);
    using Persist = typename AB::Persist;

    struct Thread {
        double sumsq;
        float peak;
        float lo[2][RL / 2 + 1];       // [mid | side][q]: bin k of (row, q); [RL/2]: bin N'/2 on thread 0
        float hi[2][RL / 2];           // bin N' - k
    };
    static MGX_HD void init(Thread& t) {
        t.sumsq = 0.0;
        t.peak = 0.f;
        MGX_UNROLL
        for (int c = 0; c < 2; ++c) {
            MGX_UNROLL
            for (int q = 0; q <= RL / 2; ++q) t.lo[c][q] = 0.f;
            MGX_UNROLL
            for (int q = 0; q < RL / 2; ++q) t.hi[c][q] = 0.f;
        }
    }
    // frames of one segment -> packed mid (SIDE = false: also the level statistics) or side samples, pass 0 -> LDS
    template <bool SIDE>
    static MGX_HD void phase_load(int tid, long long start, const AnalysisArgs& a, const Persist& ps, Thread& t,
                                  float2* lds) {
        if (!AB::active0(tid)) return;
        typename F::Tw0Full tw;
        F::expand_tw0(ps.tw0, tw);
        const MemView src = mem_view(a.x, a.n * 8);
        MGX_UNROLL
        for (int c = 0; c < CNT0; ++c) {
            const unsigned lane = ((unsigned)start + 2u * (unsigned)(tid + c * T)) * 8u;
            float2 f0[R0], f1[R0];
            MGX_UNROLL
            for (int j = 0; j < R0; ++j) {
                f0[j] = ld_f2(src, lane, (unsigned)(j * S0 * 16));
                f1[j] = ld_f2(src, lane, (unsigned)(j * S0 * 16 + 8));
            }
            float2 v[R0];
            float ss = 0.f;
            MGX_UNROLL
            for (int j = 0; j < R0; ++j) {
                float m0, s0, m1, s1;
                AB::to_ms(f0[j], m0, s0);
                AB::to_ms(f1[j], m1, s1);
                v[j] = SIDE ? make_float2(s0, s1) : make_float2(m0, m1);
                if (!SIDE) {
                    ss = fmaf(m0, m0, fmaf(m1, m1, ss));
                    t.peak = fmaxf(t.peak, fmaxf(fmaxf(fabsf(f0[j].x), fabsf(f0[j].y)), fmaxf(fabsf(f1[j].x), fabsf(f1[j].y))));
                }
            }
            if (!SIDE) t.sumsq += (double)ss;
            F::fwd0_store(v, tid, c, tw, lds);
        }
    }
    // last forward pass on the thread's row; bins go back to LDS in position order (as Analysis2Block)
    static MGX_HD void phase_row(int tid, float2* lds) {
        if (!F::has_row(tid)) return;
        float2 v[RL], w[RL];
        F::load_row(v, tid, lds);
        dft_regs<RL, false>(v);
        MGX_UNROLL
        for (int q = 0; q < RL; ++q) w[q] = v[bitrev(q, F::lr(F::LAST))];
        F::store_row(w, tid, lds);
    }
    // own lower half + mirror row's upper half -> |A + w^k B| (bin k) and |A - w^k B| (bin N' - k)
    template <bool SIDE>
    static MGX_HD void phase_magnitudes(int tid, Thread& t, const float2* lds) {
        if (!F::has_row(tid)) return;
        float2 z[RL / 2 + 2], m[RL / 2];
        F::template load_row_part<0, RL / 2 + 2>(z, tid, lds);
        F::template load_row_part<RL / 2, RL / 2>(m, F::mirror_row(tid), lds);
        const bool r0 = tid == 0;
        const int k0 = F::frequency_at(tid * RL);
        MGX_UNROLL
        for (int q = 0; q < RL / 2; ++q) {
            const float2 a = m[RL / 2 - 1 - q];
            const float2 b = q == 0 ? z[0] : m[RL / 2 - q];
            const float2 zm = make_float2(r0 ? b.x : a.x, r0 ? b.y : a.y);
            // 2A = Z + conj Zm, 2B = (Z - conj Zm) / j
            const float ax = z[q].x + zm.x, ay = z[q].y - zm.y;
            const float bx = z[q].y + zm.y, by = -(z[q].x - zm.x);
            const int k = k0 + q * F::L;
            float sn, cs;
            sincos_pi((float)k * (1.0f / (float)N), sn, cs);              // w^k = cs - j sn
            const float wx = fmaf(cs, bx, sn * by), wy = fmaf(cs, by, -sn * bx);
            const float px = ax + wx, py = ay + wy, mx = ax - wx, my = ay - wy;
            t.lo[SIDE][q] += 0.5f * fast_sqrt(fmaf(px, px, py * py));
            t.hi[SIDE][q] += 0.5f * fast_sqrt(fmaf(mx, mx, my * my));
        }
        // k = N'/2 mirrors into itself: A = Re Y, B = Im Y, w^k = -j: |A - jB| = |Y| (meaningful on thread 0 only)
        t.lo[SIDE][RL / 2] += fast_sqrt(fmaf(z[RL / 2].x, z[RL / 2].x, z[RL / 2].y * z[RL / 2].y));
    }
    // frames outside whole segments: RMS (optional) and peak only
    static MGX_HD void phase_loose_frames(int tid, long long begin, long long end, bool count_rms, const AnalysisArgs& a,
                                          Thread& t) {
        for (long long f = begin + tid; f < end; f += T) {
            const float2 lr = a.x[f];
            float m, s;
            AB::to_ms(lr, m, s);
            if (count_rms) t.sumsq += (double)(m * m);
            t.peak = fmaxf(t.peak, fmaxf(fabsf(lr.x), fabsf(lr.y)));
        }
    }
    // bins k and N' - k of (row, q < RL/2); wg_spec rows have N' + 1 bins
    static MGX_HD void phase_write_spectrum(int tid, int wg, const AnalysisArgs& a, const Thread& t) {
        if (!F::has_row(tid)) return;
        float* mid = a.wg_spec + (size_t)wg * 2 * (N + 1);
        float* side = mid + (N + 1);
        const int k0 = F::frequency_at(tid * RL);
        MGX_UNROLL
        for (int q = 0; q < RL / 2; ++q) {
            const int k = k0 + q * F::L;
            mid[k] = t.lo[0][q];
            side[k] = t.lo[1][q];
            mid[N - k] = t.hi[0][q];
            side[N - k] = t.hi[1][q];
        }
        if (tid == 0) { mid[N / 2] = t.lo[0][RL / 2]; side[N / 2] = t.lo[1][RL / 2]; }
    }
};

// ---------------------------------------------------------------------------
// fft_size = 4 N': a segment's spectra from FOUR transforms of N' points per channel pair -- the split above applied
// twice.  The 4 N' real samples r of a channel fall into the even ones e[n] = r[2n] and the odd ones o[n] = r[2n+1],
// two real sequences of 2 N' points whose spectra E_k, O_k (k = 0 .. N') are what AnalysisDouble forms, as complex
// values now, not magnitudes, and
//     R_k = E_k + w^k O_k,   R_{2N'-k} = conj(E_k - w^k O_k),   w = exp(-j pi / (2 N')),   k = 0 .. N'
// (bin N' comes out of k = N' twice: counted once).  Passes of a segment: mid even, mid odd, side even, side odd.
// 4 bins per pair {k, N'-k} and channel are 68 accumulators, and the even pass's 17 complex values have to outlive
// the odd pass's transform: a thread of the 1024 has registers for neither, so both live in the workgroup's slice of
// global scratch (a.wg_pack), laid out [slot][thread] -- every access is a wave's 64 consecutive words, a thread
// only ever reads back what it wrote itself (no barrier involved), and the slice of the one workgroup a CU runs
// stays in the L2.  The spectrum rows in wg_spec are written once, at the end, like the other analysis kernels'.
// Frames are read four times (three of them from the L2).
template <int LOG2H>
struct AnalysisQuad {
    using AB = Analysis2Block<LOG2H>;
    using AD = AnalysisDouble<LOG2H>;
    using F = Fft2<LOG2H>;
    static constexpr int N = F::N;                 // N'; the segment has 4 N frames, the spectrum 2 N + 1 bins
    static constexpr int T = F::T;
    static constexpr int R0 = F::R0;
    static constexpr int RL = F::RL;
    static constexpr int S0 = F::S(0);
    static constexpr int CNT0 = F::CNT(0);
    static constexpr int HALF = RL / 2;
    using Persist = typename AB::Persist;
    // a workgroup's scratch, in float2 units: E of the even pass [2 HALF + 1][T], then the accumulators as floats
    // [2 channels][4 HALF + 2][T]
    static constexpr int PACK_SLOTS = 2 * HALF + 1;
    static constexpr int ACC_SLOTS = 4 * HALF + 2;                      // per channel
    static constexpr size_t SCRATCH_FLOAT2 = (size_t)(PACK_SLOTS + ACC_SLOTS) * T;

    struct Thread {
        double sumsq;
        float peak;
    };
    static MGX_HD void init(Thread& t) {
        t.sumsq = 0.0;
        t.peak = 0.f;
    }
    static MGX_HD float2* pack_of(const AnalysisArgs& a, int wg) { return a.wg_pack + (size_t)wg * SCRATCH_FLOAT2; }
    static MGX_HD float* acc_of(const AnalysisArgs& a, int wg, bool side) {
        return reinterpret_cast<float*>(pack_of(a, wg) + (size_t)PACK_SLOTS * T) + (side ? (size_t)ACC_SLOTS * T : 0);
    }
    // the accumulators start from zero
    static MGX_HD void phase_clear(int tid, int wg, const AnalysisArgs& a) {
        float* acc = acc_of(a, wg, false);
        MGX_UNROLL
        for (int s = 0; s < 2 * ACC_SLOTS; ++s) acc[(size_t)s * T + tid] = 0.f;
    }
    // frames 4m + PARITY and 4m + 2 + PARITY of the segment -> packed mid (or side) samples, pass 0 -> LDS; the mid
    // passes also gather the level statistics (both parities together see every frame once)
    template <bool SIDE, int PARITY>
    static MGX_HD void phase_load(int tid, long long start, const AnalysisArgs& a, const Persist& ps, Thread& t,
                                  float2* lds) {
        if (!AB::active0(tid)) return;
        typename F::Tw0Full tw;
        F::expand_tw0(ps.tw0, tw);
        const MemView src = mem_view(a.x, a.n * 8);
        MGX_UNROLL
        for (int c = 0; c < CNT0; ++c) {
            const unsigned lane = ((unsigned)start + 4u * (unsigned)(tid + c * T) + (unsigned)PARITY) * 8u;
            float2 f0[R0], f1[R0];
            MGX_UNROLL
            for (int j = 0; j < R0; ++j) {
                f0[j] = ld_f2(src, lane, (unsigned)(j * S0 * 32));
                f1[j] = ld_f2(src, lane, (unsigned)(j * S0 * 32 + 16));
            }
            float2 v[R0];
            float ss = 0.f;
            MGX_UNROLL
            for (int j = 0; j < R0; ++j) {
                float m0, s0, m1, s1;
                AB::to_ms(f0[j], m0, s0);
                AB::to_ms(f1[j], m1, s1);
                v[j] = SIDE ? make_float2(s0, s1) : make_float2(m0, m1);
                if (!SIDE) {
                    ss = fmaf(m0, m0, fmaf(m1, m1, ss));
                    t.peak = fmaxf(t.peak, fmaxf(fmaxf(fabsf(f0[j].x), fabsf(f0[j].y)), fmaxf(fabsf(f1[j].x), fabsf(f1[j].y))));
                }
            }
            if (!SIDE) t.sumsq += (double)ss;
            F::fwd0_store(v, tid, c, tw, lds);
        }
    }
    static MGX_HD void phase_row(int tid, float2* lds) { AD::phase_row(tid, lds); }

    // the spectrum U (of the 2 N' even or odd samples) at the thread's pairs: lo[q] = U_k, hi[q] = U_{N'-k} for the bin
    // k of (row, q < RL/2); on thread 0 lo[0] = U_0, hi[0] = U_{N'} (both real) and half = U_{N'/2}
    struct Pairs {
        float2 lo[HALF], hi[HALF];
        float2 half;
    };
    static MGX_HD void phase_unmix(int tid, Pairs& u, const float2* lds) {
        if (!F::has_row(tid)) return;
        float2 z[HALF + 2], m[HALF];
        F::template load_row_part<0, HALF + 2>(z, tid, lds);
        F::template load_row_part<HALF, HALF>(m, F::mirror_row(tid), lds);
        const bool r0 = tid == 0;
        const int k0 = F::frequency_at(tid * RL);
        MGX_UNROLL
        for (int q = 0; q < HALF; ++q) {
            const float2 a = m[HALF - 1 - q];
            const float2 b = q == 0 ? z[0] : m[HALF - q];
            const float2 zm = make_float2(r0 ? b.x : a.x, r0 ? b.y : a.y);
            // 2A = Z + conj Zm, 2B = (Z - conj Zm) / j;  U_k = A + v^k B, U_{N'-k} = conj(A - v^k B), v = exp(-j pi / N')
            const float ax = z[q].x + zm.x, ay = z[q].y - zm.y;
            const float bx = z[q].y + zm.y, by = -(z[q].x - zm.x);
            const int k = k0 + q * F::L;
            float sn, cs;
            sincos_pi((float)k * (1.0f / (float)N), sn, cs);
            const float wx = fmaf(cs, bx, sn * by), wy = fmaf(cs, by, -sn * bx);
            u.lo[q] = make_float2(0.5f * (ax + wx), 0.5f * (ay + wy));
            u.hi[q] = make_float2(0.5f * (ax - wx), -0.5f * (ay - wy));
        }
        // k = N'/2 mirrors into itself: A = Re Y, B = Im Y, v^k = -j: U = A - j B = conj(Y)
        u.half = make_float2(z[HALF].x, -z[HALF].y);
    }
    // even pass: E to the workgroup's scratch
    static MGX_HD void phase_keep(int tid, int wg, const AnalysisArgs& a, const Pairs& e) {
        if (!F::has_row(tid)) return;
        float2* pack = pack_of(a, wg) + tid;
        MGX_UNROLL
        for (int q = 0; q < HALF; ++q) {
            pack[(size_t)q * T] = e.lo[q];
            pack[(size_t)(HALF + q) * T] = e.hi[q];
        }
        pack[(size_t)(2 * HALF) * T] = e.half;
    }
    // |E + w^k O| and |E - w^k O| (bins k and 2N' - k) onto two accumulators
    static MGX_HD void add_bins(float* lo, float* hi, int k, float2 e, float2 o) {
        float sn, cs;
        sincos_pi((float)k * (0.5f / (float)N), sn, cs);                  // w^k = cs - j sn
        const float tx = fmaf(cs, o.x, sn * o.y), ty = fmaf(cs, o.y, -sn * o.x);
        const float px = e.x + tx, py = e.y + ty, mx = e.x - tx, my = e.y - ty;
        *lo += fast_sqrt(fmaf(px, px, py * py));
        *hi += fast_sqrt(fmaf(mx, mx, my * my));
    }
    // odd pass: O in registers, E back from the scratch, magnitudes onto the accumulators.  Slots of a channel:
    // 4q .. 4q+3 = bins k, 2N'-k, N'-k, N'+k of pair q; 4 HALF, 4 HALF + 1 = bins N'/2, 3N'/2 (thread 0)
    template <bool SIDE>
    static MGX_HD void phase_magnitudes(int tid, int wg, const AnalysisArgs& a, const Pairs& o) {
        if (!F::has_row(tid)) return;
        const float2* pack = pack_of(a, wg) + tid;
        float* acc = acc_of(a, wg, SIDE) + tid;
        const int k0 = F::frequency_at(tid * RL);
        MGX_UNROLL
        for (int q = 0; q < HALF; ++q) {
            const int k = k0 + q * F::L;
            add_bins(acc + (size_t)(4 * q) * T, acc + (size_t)(4 * q + 1) * T, k, pack[(size_t)q * T], o.lo[q]);
            add_bins(acc + (size_t)(4 * q + 2) * T, acc + (size_t)(4 * q + 3) * T, N - k, pack[(size_t)(HALF + q) * T], o.hi[q]);
        }
        add_bins(acc + (size_t)(4 * HALF) * T, acc + (size_t)(4 * HALF + 1) * T, N / 2, pack[(size_t)(2 * HALF) * T], o.half);
    }
    static MGX_HD void phase_loose_frames(int tid, long long begin, long long end, bool count_rms, const AnalysisArgs& a,
                                          Thread& t) {
        for (long long f = begin + tid; f < end; f += T) {
            const float2 lr = a.x[f];
            float m, s;
            AB::to_ms(lr, m, s);
            if (count_rms) t.sumsq += (double)(m * m);
            t.peak = fmaxf(t.peak, fmaxf(fabsf(lr.x), fabsf(lr.y)));
        }
    }
    // accumulators -> the workgroup's spectrum rows ([2][2 N' + 1]); the pair of bin 0 holds bins 0, 2N', N' (and N'
    // once more, left out)
    static MGX_HD void phase_write_spectrum(int tid, int wg, const AnalysisArgs& a) {
        if (!F::has_row(tid)) return;
        const int k0 = F::frequency_at(tid * RL);
        for (int c = 0; c < 2; ++c) {
            float* row = a.wg_spec + ((size_t)wg * 2 + c) * (2 * N + 1);
            const float* acc = acc_of(a, wg, c != 0) + tid;
            MGX_UNROLL
            for (int q = 0; q < HALF; ++q) {
                const int k = k0 + q * F::L;
                row[k] = acc[(size_t)(4 * q) * T];
                row[2 * N - k] = acc[(size_t)(4 * q + 1) * T];
                row[N - k] = acc[(size_t)(4 * q + 2) * T];
                if (k != 0) row[N + k] = acc[(size_t)(4 * q + 3) * T];
            }
            if (tid == 0) {
                row[N / 2] = acc[(size_t)(4 * HALF) * T];
                row[3 * N / 2] = acc[(size_t)(4 * HALF + 1) * T];
            }
        }
    }
};

// ---------------------------------------------------------------------------
// Scalar epilogue of the analysis (host restatement used by the CPU emulation; the device
// version is k_levels in mgx_kernels.h).  match_levels.py:62-71,93-103.
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
