// Hold and release low-passes of order 2 and 3 (hyrax.py:55-73 with hold_filter_order /
// release_filter_order > 1) inside the chunked limiter of limiter_kernel.h.  (Written for order K; see
// LIMITER_MAX_ORDER for which filters are run and which are refused.)
//
// scipy.signal.lfilter runs an order-K filter as a transposed direct form II section with K states:
//     y = b0 x + z[0];   z[i] = z[i+1] + b[i+1] x - a[i+1] y   (i < K-1);   z[K-1] = b[K] x - a[K] y
// The state update is linear, z' = A z + B x, so everything limiter_kernel.h does with first-order
// filters carries over with K x K matrices where it has scalars.
// State maps are NOT formed in the section's own basis.  A is the companion matrix of p(lambda) = lambda^K + a1
// lambda^(K-1) + ... whose roots lie at a distance d = 2 pi fc / fs from 1 and from each other: its powers have
// entries ~n^(K-1) that cancel to O(1) results, and a composition of such maps in float64 loses a factor d^-(K-1)
// (1e6 for a third-order 7 Hz filter: RMS error 2e-2 when round 6 first tried; the second-order filters got away with
// 4e-8).  Read z as the coefficients of a polynomial s(lambda) = sum_i z_i lambda^(K-1-i): A is multiplication by
// lambda modulo p.  In the basis q^(K-1-i), q = (lambda - 1) / d -- the delta-operator form -- the same
// multiplication is I + d C with C the companion matrix of p(1 + d q) / d^K, whose entries are O(1): products of
// such maps are as accurate as float64 is (2e-8 against scipy.signal.lfilter for that 7 Hz filter, 1e-10 at order 2,
// on a 9 s burst signal).  The host derives dm = d C, bd and zw from the section's float64 coefficients in extended
// precision (host_params.h shifted_basis); a thread runs its 16 frames from zero in the shifted basis for the block
// map, takes its entering state back to the section's basis (z = zw w: O(1) entries, no cancellation) and produces
// its outputs with scipy's own recursion, so what comes out is the reference's arithmetic on a state good to 1e-15.
// With that:
//   * a thread's 16 frames act on the carried state as z -> A^count z + v (v = the state its frames
//     produce from zero): a StateMap; maps compose across the workgroup by an ordered scan;
//   * a chunk publishes the K state words its core frames produce from a zero carry, and a chunk's
//     carry is sum_m (A^chunk)^m * published[chunk-1-m], truncated where the power's largest entry is
//     below 1e-10 (the matrices come from the host: extended precision, rounded to float64).
// Everything here is float64: near z = 1 the poles of an order-K Butterworth low-pass lie within
// ~1e-5 of each other and float32 coefficients would move them by more than that.  This is a path for
// unusual configurations: it is written for clarity and exactness, the scans go through LDS in
// barrier-separated phases (so the CPU emulation runs the very same code), and it shares the load,
// window, attack and store phases with the first-order kernel.  A filter of lower order than K is
// carried as an order-K one with zero trailing coefficients (exact).
#pragma once

#include "limiter_kernel.h"

namespace mgx {

// Which orders run.  The reference runs these filters in transfer-function form (scipy.signal.butter -> lfilter),
// whose K poles lie at a distance d = 2 pi fc / fs from z = 1 and from each other: ~4e-5 for the release filter
// (0.27 Hz), ~1e-3 for the hold filter (7 Hz) at the default settings.  That form is ill-conditioned in float64 when
// d^(K - 1/2) gets small: the reference's own sample-by-sample recursion then carries rounding noise of about
// 1.1e-16 / d^(K - 1/2) of full scale -- 1.2e-5 for a third-order release filter (measured 1.9e-5 against the same
// recursion in 80-bit arithmetic, tests/test_limiter_order3_conditioning.py), more than the signal at order 4 --
// which no reordered evaluation can reproduce, and the state matrix's power A^3520 computed in double has spectral
// radius 141 where the exact one has 0.94, so chunk aggregates cannot be formed either.  A third-order HOLD filter
// is clean (3.5e-9).  So the rule is the conditioning, not the order (host_params.h limiter_params): a filter is
// refused when that estimate exceeds 1e-6; what passes runs here with K = the larger order (<= 3: the orders this
// file is instantiated for), a filter of lower order carried with zero trailing coefficients.
constexpr int LIMITER_MAX_ORDER = 3;

template <int K>
struct IirK {
    double b[K + 1];               // b[0..K]
    double a[K + 1];               // a[0] = 1
    // the same filter's state update in the SHIFTED basis the block maps and chunk words live in (see "state maps" below):
    // w' = w + dm w + bd x, and the section's own states are z = zw w
    double dm[K][K];
    double bd[K];
    double zw[K][K];
};
template <int K>
struct StateMap {
    double m[K][K];
    double v[K];
};

template <int K>
struct GeneralArgs {
    IirK<K> hold, rel;
    const double* pow_hold;        // [17][K][K]: A_hold^j, j = 0..16
    const double* pow_rel;
    const double* w_hold;          // [n_hold][K][K]: (A_hold^chunk)^m
    const double* w_rel;
    int n_hold, n_rel;
    unsigned long long* words;     // [2][K][nchunks]: hold, release state words of every chunk
};

template <int K>
struct LimiterGeneral {
    using LB = LimiterBlock<256>;
    using Thread = typename LB::Thread;
    static constexpr int T = LB::T, E = LB::E, G = 16, NG = T / G;
    static constexpr int MAP_DOUBLES = K * K + K;
    // LDS (doubles), after the first-order kernel's carve: T maps | NG group maps | 1 total | 2*K carries
    static constexpr size_t SCAN_OFF_BYTES = (LB::LDS_BYTES + 15) / 16 * 16;
    static constexpr int SCAN_DOUBLES = (T + NG + 1) * MAP_DOUBLES + 2 * K;
    static constexpr size_t LDS_BYTES = SCAN_OFF_BYTES + (size_t)SCAN_DOUBLES * 8;

    static MGX_HD StateMap<K>* maps(float* lds) {
        return reinterpret_cast<StateMap<K>*>(reinterpret_cast<char*>(lds) + SCAN_OFF_BYTES);
    }
    static MGX_HD double* carries(float* lds) { return reinterpret_cast<double*>(maps(lds) + T + NG + 1); }

    static MGX_HD StateMap<K> identity() {
        StateMap<K> r;
        for (int i = 0; i < K; ++i) {
            for (int j = 0; j < K; ++j) r.m[i][j] = i == j ? 1.0 : 0.0;
            r.v[i] = 0.0;
        }
        return r;
    }
    // apply `first`, then `second`
    static MGX_HD StateMap<K> then(const StateMap<K>& first, const StateMap<K>& second) {
        StateMap<K> r;
        for (int i = 0; i < K; ++i) {
            for (int j = 0; j < K; ++j) {
                double s = 0.0;
                for (int k = 0; k < K; ++k) s = fma(second.m[i][k], first.m[k][j], s);
                r.m[i][j] = s;
            }
            double s = second.v[i];
            for (int k = 0; k < K; ++k) s = fma(second.m[i][k], first.v[k], s);
            r.v[i] = s;
        }
        return r;
    }
    static MGX_HD void apply(const StateMap<K>& map, const double (&c)[K], double (&z)[K]) {
        for (int i = 0; i < K; ++i) {
            double s = map.v[i];
            for (int k = 0; k < K; ++k) s = fma(map.m[i][k], c[k], s);
            z[i] = s;
        }
    }
    // one frame of scipy's lfilter (transposed direct form II)
    static MGX_HD double step(const IirK<K>& f, double (&z)[K], double x) {
        const double y = fma(f.b[0], x, z[0]);
        for (int i = 0; i < K - 1; ++i) z[i] = z[i + 1] + f.b[i + 1] * x - f.a[i + 1] * y;
        z[K - 1] = f.b[K] * x - f.a[K] * y;
        return y;
    }
    // one frame of the same filter's state in the shifted basis: w' = w + dm w + bd x
    static MGX_HD void step_shifted(const IirK<K>& f, double (&w)[K], double x) {
        double n[K];
        for (int i = 0; i < K; ++i) {
            double s = f.bd[i] * x;
            for (int k = 0; k < K; ++k) s = fma(f.dm[i][k], w[k], s);
            n[i] = w[i] + s;
        }
        for (int i = 0; i < K; ++i) w[i] = n[i];
    }
    // shifted state -> the section's own states
    static MGX_HD void to_section(const IirK<K>& f, const double (&w)[K], double (&z)[K]) {
        for (int i = 0; i < K; ++i) {
            double s = 0.0;
            for (int k = 0; k < K; ++k) s = fma(f.zw[i][k], w[k], s);
            z[i] = s;
        }
    }
    // the map of a thread's first `count` frames (shifted basis; pow = powers of I + dm)
    static MGX_HD StateMap<K> block_map(const IirK<K>& f, const float (&x)[E], int count, const double* pow) {
        StateMap<K> r;
        double w[K];
        for (int i = 0; i < K; ++i) w[i] = 0.0;
        for (int j = 0; j < E; ++j)
            if (j < count) step_shifted(f, w, (double)x[j]);
        const double* p = pow + (size_t)count * K * K;
        for (int i = 0; i < K; ++i) {
            for (int j = 0; j < K; ++j) r.m[i][j] = p[i * K + j];
            r.v[i] = w[i];
        }
        return r;
    }

    // ---- ordered scan of the T maps through LDS: put | barrier | groups | barrier | top | barrier | prefix
    static MGX_HD void scan_put(float* lds, int tid, const StateMap<K>& m) { maps(lds)[tid] = m; }
    static MGX_HD void scan_groups(float* lds, int tid) {
        if (tid < NG) {
            StateMap<K>* e = maps(lds) + tid * G;
            StateMap<K> run = identity();
            for (int i = 0; i < G; ++i) {
                const StateMap<K> m = e[i];
                e[i] = run;
                run = then(run, m);
            }
            maps(lds)[T + tid] = run;
        }
    }
    static MGX_HD void scan_top(float* lds, int tid) {
        if (tid == 0) {
            StateMap<K>* g = maps(lds) + T;
            StateMap<K> run = identity();
            for (int i = 0; i < NG; ++i) {
                const StateMap<K> m = g[i];
                g[i] = run;
                run = then(run, m);
            }
            maps(lds)[T + NG] = run;
        }
    }
    // composition of the maps of all threads before `tid`
    static MGX_HD StateMap<K> scan_prefix(float* lds, int tid) { return then(maps(lds)[T + tid / G], maps(lds)[tid]); }
    static MGX_HD StateMap<K> scan_whole(float* lds) { return maps(lds)[T + NG]; }

    // ---- look-back over the K state words of the predecessors ---------------------------------------
    static MGX_HD unsigned long long* word(const GeneralArgs<K>& g, long long nchunks, int filter, int k, long long chunk) {
        return g.words + ((size_t)filter * K + k) * nchunks + chunk;
    }
    static MGX_HD void publish(const GeneralArgs<K>& g, long long nchunks, int filter, long long chunk, const double (&v)[K]) {
        for (int k = 0; k < K; ++k) publish_word(word(g, nchunks, filter, k, chunk), double_bits(v[k]));
    }
    // this lane's share of sum_m W[m] * words[chunk-1-m] (the caller adds the 64 shares per component)
    static MGX_HD void take(int lane, long long chunk, int filter, const GeneralArgs<K>& g, const LimiterArgs& a,
                            double (&acc)[K]) {
        const double* w = filter == 0 ? g.w_hold : g.w_rel;
        const int count = filter == 0 ? g.n_hold : g.n_rel;
        for (int i = 0; i < K; ++i) acc[i] = 0.0;
        for (int m = lane; m < count; m += 64) {
            const long long c = chunk - 1 - m;
            if (c < 0) break;
            double b[K];
            for (int k = 0; k < K; ++k) {
                unsigned long long* q = word(g, a.nchunks, filter, k, c);
                unsigned long long v = poll_word(q);
                int spins = 0;
                long long t0 = 0;
                while (v == LIMITER_UNPUBLISHED && wait_on(spins, t0, a.gave_up, LB::MAX_SPINS)) {
                    backoff(spins);
                    v = poll_word(q);
                    ++spins;
                }
                if (v == LIMITER_UNPUBLISHED) {
                    *a.error = 1;
                    v = 0;
                }
                b[k] = bits_double(v);
            }
            const double* wm = w + (size_t)m * K * K;
            for (int i = 0; i < K; ++i)
                for (int k = 0; k < K; ++k) acc[i] = fma(wm[i * K + k], b[k], acc[i]);
        }
    }

    // ---- the phases that replace phase_hold / phase_gain of the first-order kernel -------------------
    // hold output from the true entering state, attack carry term, max(sh, ho) -> map of the release filter
    static MGX_HD StateMap<K> phase_hold(int tid, const LimiterArgs& a, const GeneralArgs<K>& g, Thread& th,
                                         const StateMap<K>& pre, const double (&hold_carry)[K], double att_deferred) {
        for (int j = 0; j < E; ++j) { th.x2[j] = 0.f; th.mx[j] = 0.f; }
        if (!th.core) return identity();
        float pw = (float)(att_deferred * LB::attack_kappa(a.att) * th.att_decay);
        double w[K], z[K];
        apply(pre, hold_carry, w);
        to_section(g.hold, w, z);
        for (int j = 0; j < E; ++j) {
            const bool in = j < th.valid;
            const float ho = in ? (float)step(g.hold, z, (double)th.sh[j]) : 0.f;
            const float ga = in ? th.yb[j] + pw : 0.f;
            pw *= a.attf.alpha;
            th.x2[j] = fmaxf(th.sh[j], ho);                      // hyrax.py:73
            th.mx[j] = fmaxf(ho, ga);
        }
        return th.valid > 0 ? block_map(g.rel, th.x2, th.valid, g.pow_rel) : identity();
    }
    // release output -> gain -> plane (hyrax.py:75,97 without g0: phase_store adds it)
    static MGX_HD void phase_gain(int tid, const GeneralArgs<K>& g, Thread& th, const StateMap<K>& pre,
                                  const double (&rel_carry)[K], float* lds) {
        if (!th.core) return;
        double w[K], z[K];
        apply(pre, rel_carry, w);
        to_section(g.rel, w, z);
        float* gn = LB::plane(lds) + tid * LB::STRIDE;
        for (int j = 0; j < E; ++j) {
            const float ro = j < th.valid ? (float)step(g.rel, z, (double)th.x2[j]) : 0.f;
            gn[j] = 1.0f - fmaxf(th.mx[j], ro);
        }
    }
};

}  // namespace mgx
