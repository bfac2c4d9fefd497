// Hyrax brick-wall limiter (matchering/limiter/hyrax.py:78-99) as three
// streaming passes over the un-limited result.
//
// Reference data flow (every array is length n, float64):
//   rect  = max(|L|,|R|) floored at thr, / thr                    dsp.py:117-121
//   g0    = 1 - 1/rect                                            hyrax.py:87
//   sl    = centred sliding max of g0, window 2w-1, w = odd(att)  hyrax.py:35-37
//   gA    = filtfilt(1-pole rho = exp(coef/att), sl)              hyrax.py:48-51
//   sh    = trailing sliding max of sl over `hold` samples        hyrax.py:38-40
//   ho    = lfilter(butter(hold order, hold Hz), sh)              hyrax.py:61-66
//   ro    = lfilter(butter(rel order, rel Hz), max(sh, ho))       hyrax.py:68-73
//   gain  = 1 - max(g0, gA, max(ho, ro));  out = x * gain         hyrax.py:75,97,99
//
// GPU formulation.  A chunk of C frames is one 256-thread workgroup; thread t
// owns the 16 consecutive frames of "block" t (halo blocks on both sides feed
// the windows).  Sliding maxima use per-block prefix/suffix/total maxima in LDS
// (two LDS reads per output, any window >= 16; a direct loop serves tiny
// windows).  Each first-order recurrence (state z: y[n] = b0 x[n] + z[n-1],
// z[n] = alpha z[n-1] + beta x[n], i.e. scipy's transposed direct form II) is a
// per-thread run with zero carry + an ordered affine scan (scan_util.h) inside
// the workgroup + a tiny scan of chunk aggregates between passes:
//
//   pass 1: g0, sl, sh -> chunk aggregates of ho (input sh) and of the forward
//           attack smoother yf (input sl)
//   scan 1: carries of ho, yf; exact filtfilt edge states (odd extension by 6,
//           steady-state initial conditions, scipy.signal.filtfilt defaults)
//   pass 2: exact ho, yf -> chunk aggregates of ro (input max(sh,ho), which is
//           why ho must be exact first) and of the backward attack smoother yb
//   scan 2: carries of ro (left to right) and yb (right to left)
//   pass 3: everything with exact carries -> gain -> out = x*gain*post_gain
//
// Filter state is float64; window maxima and g0 are float32 (|error| <= 6e-8 on
// a gain that multiplies samples <= ~4).  HBM traffic: 3 x 8 B read + 8 B write
// per frame.
#pragma once

#include "scan_util.h"

namespace mgx {

struct Iir1 {
    double b0, alpha, beta;        // y = b0*x + z_prev ; z = alpha*z_prev + beta*x
};

struct LimiterArgs {
    const float2* y;               // (n,2) level-corrected result before the final gains
    long long n;
    float2* out;                   // (n,2) limited output (pass 3)
    const double* gain;            // device scalar: accumulated level-correction gain
    const double* post_gain;       // device scalar: final amplitude coefficient (stages.py:203)
    const int* active;             // device flag: 0 => limiter early-out (hyrax.py:83-85)
    float threshold;
    int hw;                        // attack half window = odd(attack) - 1
    int hb;                        // hold look-back     = hold - 1
    Iir1 att, hold, rel;
    long long nchunks;
    Affine* agg;                   // [4][nchunks]: ho, yf, ro, yb chunk aggregates
    double* carry;                 // [4][nchunks]: state entering each chunk (yb: from the right)
    float* edge_sl;                // [14]: sl[0..6] and sl[n-7..n-1] (filtfilt odd extension)
    double* edge_state;            // [2]: yf state after frame n-1, yb state entering frame n-1
};

struct LimiterBlock {
    static constexpr int T = 256;
    static constexpr int E = 16;
    static constexpr int STRIDE = E + 1;              // LDS row stride (floats): conflict-free columns
    static constexpr int G = 16;
    using Scan = WgScan<T, G, 2>;

    // LDS carve (bytes): three float planes + two per-block maxima + scan scratch
    static constexpr int PLANE = T * STRIDE;
    static constexpr int LDS_FLOATS = 3 * PLANE + 2 * T;
    static constexpr size_t LDS_BYTES = (size_t)LDS_FLOATS * 4 + (size_t)Scan::SCRATCH * sizeof(Affine) + 16;

    struct Geometry {
        int ga, gh, gl, gr, core_blocks, chunk;
    };
    static MGX_HD Geometry geometry(int hw, int hb) {
        Geometry g;
        g.ga = hw / E + 1;
        g.gh = hb / E + 1;
        g.gl = g.ga + g.gh;
        g.gr = g.ga;
        g.core_blocks = T - g.gl - g.gr;
        g.chunk = g.core_blocks * E;
        return g;
    }

    struct Thread {
        float2 v[E];              // frames * gain (float32, = result_no_limiter)
        float g0[E], sl[E], slp[E], sh[E];
        // float32 snapshots of float64 recurrences (the recurrences themselves and every
        // carry are float64; a snapshot only rounds one output by <= 6e-8)
        float x2[E];              // max(sh, ho)
        float ho[E], yf[E];
        float zl0[E], zl1[E];     // zero-carry run states of the two recurrences in flight
        long long base;           // first frame of this thread's block
        int valid;                // frames of the block inside [0, n)
        bool core, has_sl;
    };

    static MGX_HD float* plane(float* lds, int i) { return lds + i * PLANE; }
    static MGX_HD float* block_max(float* lds, int i) { return lds + 3 * PLANE + i * T; }
    static MGX_HD Affine* scan_area(float* lds) {
        size_t off = ((size_t)LDS_FLOATS * 4 + 15) & ~(size_t)15;
        return reinterpret_cast<Affine*>(reinterpret_cast<char*>(lds) + off);
    }
    static MGX_HD int lds_index(int sample) {       // sample index relative to block 0 of the chunk
        return (sample >> 4) * STRIDE + (sample & 15);
    }

    // ---- phase 1: load, g0, block prefix/suffix maxima ---------------------------
    static MGX_HD void phase_g0(int tid, long long chunk, const LimiterArgs& a, Thread& th, float* lds) {
        const Geometry geo = geometry(a.hw, a.hb);
        th.base = chunk * geo.chunk + (long long)(tid - geo.gl) * E;
        th.core = tid >= geo.gl && tid < T - geo.gr;
        th.has_sl = tid >= geo.ga && tid < T - geo.gr;
        const long long left = a.n - th.base;
        th.valid = th.base < 0 ? 0 : (left >= E ? E : (left > 0 ? (int)left : 0));
        const double g = *a.gain;
        MGX_UNROLL
        for (int j = 0; j < E; ++j) {
            const long long n = th.base + j;
            float2 v = make_float2(0.f, 0.f);
            if (n >= 0 && n < a.n) {
                const float2 y = a.y[n];
                v = make_float2((float)((double)y.x * g), (float)((double)y.y * g));
            }
            th.v[j] = v;
            const float amax = fmaxf(fabsf(v.x), fabsf(v.y));
            th.g0[j] = amax > a.threshold ? 1.0f - a.threshold / amax : 0.f;
        }
        float* gp = plane(lds, 0) + tid * STRIDE;
        float* gs = plane(lds, 1) + tid * STRIDE;
        if (a.hw >= E) {
            float run = 0.f;
            MGX_UNROLL
            for (int j = 0; j < E; ++j) { run = fmaxf(run, th.g0[j]); gp[j] = run; }
            block_max(lds, 0)[tid] = run;
            run = 0.f;
            MGX_UNROLL
            for (int j = E - 1; j >= 0; --j) { run = fmaxf(run, th.g0[j]); gs[j] = run; }
        } else {
            MGX_UNROLL
            for (int j = 0; j < E; ++j) gp[j] = th.g0[j];
        }
    }

    // ---- phase 2: sl = max g0[n-hw .. n+hw] ------------------------------------------
    static MGX_HD void phase_sl(int tid, const LimiterArgs& a, Thread& th, float* lds) {
        const float* gp = plane(lds, 0);
        const float* gs = plane(lds, 1);
        float* ss = plane(lds, 2) + tid * STRIDE;
        if (th.has_sl) {
            if (a.hw >= E) {
                const int aa = a.hw >> 4, bb = a.hw & 15;
                const float* bm = block_max(lds, 0);
                float core = 0.f;
                for (int k = tid - aa + 1; k <= tid + aa - 1; ++k) core = fmaxf(core, bm[k]);
                const float ml = bm[tid - aa], mr = bm[tid + aa];
                MGX_UNROLL
                for (int j = 0; j < E; ++j) {
                    const float lft = j >= bb ? gs[(tid - aa) * STRIDE + (j - bb)]
                                              : fmaxf(gs[(tid - aa - 1) * STRIDE + (E + j - bb)], ml);
                    const float rgt = j + bb < E ? gp[(tid + aa) * STRIDE + (j + bb)]
                                                 : fmaxf(gp[(tid + aa + 1) * STRIDE + (j + bb - E)], mr);
                    th.sl[j] = fmaxf(core, fmaxf(lft, rgt));
                }
            } else {
                MGX_UNROLL
                for (int j = 0; j < E; ++j) {
                    float m = 0.f;
                    const int c = tid * E + j;
                    for (int k = -a.hw; k <= a.hw; ++k) m = fmaxf(m, gp[lds_index(c + k)]);
                    th.sl[j] = m;
                }
            }
            MGX_UNROLL
            for (int j = 0; j < E; ++j) {
                const long long n = th.base + j;
                if (n < 0 || n >= a.n) th.sl[j] = 0.f;      // windows are truncated at the array ends
            }
            // record the filtfilt edge samples
            MGX_UNROLL
            for (int j = 0; j < E; ++j) {
                const long long n = th.base + j;
                if (th.core && n >= 0 && n < a.n) {
                    if (n < 7) a.edge_sl[n] = th.sl[j];
                    if (n >= a.n - 7) a.edge_sl[7 + (n - (a.n - 7))] = th.sl[j];
                }
            }
            if (a.hb >= E) {
                float run = 0.f;
                MGX_UNROLL
                for (int j = 0; j < E; ++j) { run = fmaxf(run, th.sl[j]); th.slp[j] = run; }
                block_max(lds, 1)[tid] = run;
                run = 0.f;
                MGX_UNROLL
                for (int j = E - 1; j >= 0; --j) { run = fmaxf(run, th.sl[j]); ss[j] = run; }
            } else {
                MGX_UNROLL
                for (int j = 0; j < E; ++j) ss[j] = th.sl[j];
            }
        }
    }

    // ---- first-order recurrence over a thread's run ------------------------------------
    // forward: zl[j] = state after frame j with zero carry, for j < valid
    static MGX_HD Affine run_forward(const Iir1& f, const float (&x)[E], int valid, float (&zl)[E]) {
        double z = 0.0, pa = 1.0;
        MGX_UNROLL
        for (int j = 0; j < E; ++j) {
            if (j < valid) { z = fma(f.alpha, z, f.beta * (double)x[j]); pa *= f.alpha; }
            zl[j] = (float)z;
        }
        return Affine{pa, z};
    }
    // y[j] for a forward run given the state entering the run
    static MGX_HD void out_forward(const Iir1& f, const float (&x)[E], const float (&zl)[E],
                                   double carry, double (&y)[E]) {
        double pw = 1.0;
        MGX_UNROLL
        for (int j = 0; j < E; ++j) {
            const double zprev = (j == 0 ? 0.0 : (double)zl[j - 1]) + pw * carry;
            y[j] = fma(f.b0, (double)x[j], zprev);
            pw *= f.alpha;
        }
    }
    // backward (right to left) over the valid frames of the run
    static MGX_HD Affine run_backward(const Iir1& f, const float (&x)[E], int valid, float (&zl)[E]) {
        double z = 0.0, pa = 1.0;
        MGX_UNROLL
        for (int j = E - 1; j >= 0; --j) {
            if (j < valid) { z = fma(f.alpha, z, f.beta * (double)x[j]); pa *= f.alpha; }
            zl[j] = (float)z;
        }
        return Affine{pa, z};
    }
    static MGX_HD void out_backward(const Iir1& f, const float (&x)[E], const float (&zl)[E], int valid,
                                    double carry, double (&y)[E]) {
        double pw = 1.0;
        MGX_UNROLL
        for (int j = E - 1; j >= 0; --j) {
            if (j < valid) {
                const double znext = (j == valid - 1 ? 0.0 : (double)zl[j + 1]) + pw * carry;
                y[j] = fma(f.b0, (double)x[j], znext);
                pw *= f.alpha;
            } else {
                y[j] = 0.0;
            }
        }
    }

    // ---- phase 3: sh, then zero-carry runs of ho (input sh) and yf (input sl) -----------
    static MGX_HD void phase_sh_runs(int tid, const LimiterArgs& a, Thread& th, float* lds) {
        Affine m_ho = affine_identity(), m_yf = affine_identity();
        if (th.core) {
            const float* ss = plane(lds, 2);
            if (a.hb >= E) {
                const int aa = a.hb >> 4, bb = a.hb & 15;
                const float* bm = block_max(lds, 1);
                float core = 0.f;
                for (int k = tid - aa + 1; k <= tid - 1; ++k) core = fmaxf(core, bm[k]);
                const float ml = bm[tid - aa];
                MGX_UNROLL
                for (int j = 0; j < E; ++j) {
                    const float lft = j >= bb ? ss[(tid - aa) * STRIDE + (j - bb)]
                                              : fmaxf(ss[(tid - aa - 1) * STRIDE + (E + j - bb)], ml);
                    th.sh[j] = fmaxf(core, fmaxf(lft, th.slp[j]));
                }
            } else {
                MGX_UNROLL
                for (int j = 0; j < E; ++j) {
                    float m = 0.f;
                    const int c = tid * E + j;
                    for (int k = 0; k <= a.hb; ++k) m = fmaxf(m, ss[lds_index(c - k)]);
                    th.sh[j] = m;
                }
            }
            m_ho = run_forward(a.hold, th.sh, th.valid, th.zl0);
            m_yf = run_forward(a.att, th.sl, th.valid, th.zl1);
        }
        Affine* sc = scan_area(lds);
        Scan::put(sc, 0, tid, m_ho);
        Scan::put(sc, 1, tid, m_yf);
    }

    // pass 1 ends here: thread 0 publishes the chunk aggregates (call after the scan phases)
    static MGX_HD void phase_publish(int tid, long long chunk, const LimiterArgs& a, const float* lds,
                                     int slot0, int slot1) {
        if (tid == 0) {
            const Affine* sc = scan_area(const_cast<float*>(lds));
            a.agg[slot0 * a.nchunks + chunk] = Scan::whole(sc, 0);
            a.agg[slot1 * a.nchunks + chunk] = Scan::whole(sc, 1);
        }
    }

    // ---- phase 4 (passes 2, 3): exact ho, yf; zero-carry runs of ro and of backward yb ----
    // reads the scan results of phase 3, then (after the caller's barrier) the scan area is
    // reused: this phase only computes, phase_put_second stores.
    static MGX_HD void phase_exact_first(int tid, long long chunk, const LimiterArgs& a, Thread& th,
                                         const float* lds, Affine& m_ro, Affine& m_yb) {
        m_ro = affine_identity();
        m_yb = affine_identity();
        if (th.core) {
            const Affine* sc = scan_area(const_cast<float*>(lds));
            const double c_ho = affine_apply(Scan::prefix(sc, 0, tid), a.carry[0 * a.nchunks + chunk]);
            const double c_yf = affine_apply(Scan::prefix(sc, 1, tid), a.carry[1 * a.nchunks + chunk]);
            double ho[E], yf[E];
            out_forward(a.hold, th.sh, th.zl0, c_ho, ho);
            out_forward(a.att, th.sl, th.zl1, c_yf, yf);
            MGX_UNROLL
            for (int j = 0; j < E; ++j) {
                th.ho[j] = (float)ho[j];
                th.yf[j] = (float)yf[j];
                th.x2[j] = fmaxf(th.sh[j], th.ho[j]);                             // hyrax.py:73
            }
            m_ro = run_forward(a.rel, th.x2, th.valid, th.zl0);
            m_yb = run_backward(a.att, th.yf, th.valid, th.zl1);
        }
    }
    static MGX_HD void phase_put_second(int tid, float* lds, Affine m_ro, Affine m_yb) {
        Affine* sc = scan_area(lds);
        Scan::put(sc, 0, tid, m_ro);
        Scan::put(sc, 1, T - 1 - tid, m_yb);          // right-to-left scan order
    }

    // ---- phase 5 (pass 3): exact ro, yb -> gain -> output ------------------------------------
    static MGX_HD void phase_output(int tid, long long chunk, const LimiterArgs& a, Thread& th,
                                    const float* lds) {
        if (!th.core) return;
        const Affine* sc = scan_area(const_cast<float*>(lds));
        const double c_ro = affine_apply(Scan::prefix(sc, 0, tid), a.carry[2 * a.nchunks + chunk]);
        const double c_yb = affine_apply(Scan::prefix(sc, 1, T - 1 - tid), a.carry[3 * a.nchunks + chunk]);
        double ro[E], yb[E];
        out_forward(a.rel, th.x2, th.zl0, c_ro, ro);
        out_backward(a.att, th.yf, th.zl1, th.valid, c_yb, yb);
        const double post = *a.post_gain;
        MGX_UNROLL
        for (int j = 0; j < E; ++j) {
            if (j < th.valid) {
                const double grel = fmax((double)th.ho[j], ro[j]);                  // hyrax.py:75
                const double gain = 1.0 - fmax((double)th.g0[j], fmax(yb[j], grel));   // hyrax.py:97
                const double s = gain * post;
                a.out[th.base + j] = make_float2((float)((double)th.v[j].x * s), (float)((double)th.v[j].y * s));
            }
        }
    }
};

// ---------------------------------------------------------------------------
// Scans of chunk aggregates (one 1024-thread workgroup).
// ---------------------------------------------------------------------------
struct ChunkScan {
    static constexpr int T = 1024;
    static constexpr int G = 32;
    using Scan = WgScan<T, G, 2>;
    static constexpr size_t LDS_BYTES = (size_t)Scan::SCRATCH * sizeof(Affine);

    static MGX_HD long long per_thread(long long nchunks) { return (nchunks + T - 1) / T; }

    // slotA scans left to right; slotB left to right (forward_b) or right to left
    static MGX_HD void phase_put(int tid, const LimiterArgs& a, int slot_a, int slot_b, bool forward_b,
                                 Affine* sc) {
        const long long per = per_thread(a.nchunks);
        Affine ma = affine_identity(), mb = affine_identity();
        for (long long c = tid * per; c < (tid + 1) * per && c < a.nchunks; ++c)
            ma = affine_then(ma, a.agg[slot_a * a.nchunks + c]);
        if (forward_b) {
            for (long long c = tid * per; c < (tid + 1) * per && c < a.nchunks; ++c)
                mb = affine_then(mb, a.agg[slot_b * a.nchunks + c]);
            Scan::put(sc, 1, tid, mb);
        } else {
            for (long long c = (tid + 1) * per - 1; c >= tid * per; --c)
                if (c < a.nchunks) mb = affine_then(mb, a.agg[slot_b * a.nchunks + c]);
            Scan::put(sc, 1, T - 1 - tid, mb);
        }
        Scan::put(sc, 0, tid, ma);
    }

    // initial states: scan 1 (ho zero state; yf = filtfilt left edge), scan 2 (ro zero; yb = stored edge)
    static MGX_HD double filtfilt_left_state(const LimiterArgs& a) {
        const double x0 = (double)a.edge_sl[0];
        const double zi = a.att.beta / (1.0 - a.att.alpha);          // scipy lfilter_zi for this section
        double z = 0.0;
        for (int i = 0; i < 6; ++i) {
            const double e = 2.0 * x0 - (double)a.edge_sl[6 - i];    // odd extension, padlen 6
            if (i == 0) z = zi * e;
            z = fma(a.att.alpha, z, a.att.beta * e);
        }
        return z;
    }
    // given the forward state after frame n-1, run the right odd extension forward, then the
    // backward filter over it: returns the backward state entering frame n-1
    static MGX_HD double filtfilt_right_state(const LimiterArgs& a, double z_end) {
        const double xl = (double)a.edge_sl[13];
        const double zi = a.att.beta / (1.0 - a.att.alpha);
        double yfe[6];
        double z = z_end;
        for (int i = 0; i < 6; ++i) {
            const double e = 2.0 * xl - (double)a.edge_sl[12 - i];   // 2 x[n-1] - x[n-2-i]
            yfe[i] = fma(a.att.b0, e, z);
            z = fma(a.att.alpha, z, a.att.beta * e);
        }
        double zb = zi * yfe[5];
        for (int i = 5; i >= 0; --i) zb = fma(a.att.alpha, zb, a.att.beta * yfe[i]);
        return zb;
    }

    static MGX_HD void phase_write(int tid, const LimiterArgs& a, int slot_a, int slot_b, bool forward_b,
                                   double init_a, double init_b, const Affine* sc) {
        const long long per = per_thread(a.nchunks);
        double za = affine_apply(Scan::prefix(sc, 0, tid), init_a);
        for (long long c = tid * per; c < (tid + 1) * per && c < a.nchunks; ++c) {
            a.carry[slot_a * a.nchunks + c] = za;
            za = affine_apply(a.agg[slot_a * a.nchunks + c], za);
        }
        if (forward_b) {
            double zb = affine_apply(Scan::prefix(sc, 1, tid), init_b);
            for (long long c = tid * per; c < (tid + 1) * per && c < a.nchunks; ++c) {
                a.carry[slot_b * a.nchunks + c] = zb;
                zb = affine_apply(a.agg[slot_b * a.nchunks + c], zb);
                // scan 1: the forward smoother's state after the last frame seeds the
                // right-hand filtfilt edge (consumed by scan 2 through filtfilt_right_state)
                if (c == a.nchunks - 1 && a.edge_state) a.edge_state[0] = zb;
            }
        } else {
            double zb = affine_apply(Scan::prefix(sc, 1, T - 1 - tid), init_b);
            for (long long c = (tid + 1) * per - 1; c >= tid * per; --c) {
                if (c < a.nchunks) {
                    a.carry[slot_b * a.nchunks + c] = zb;
                    zb = affine_apply(a.agg[slot_b * a.nchunks + c], zb);
                }
            }
        }
    }
};

}  // namespace mgx
