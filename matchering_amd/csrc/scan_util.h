// Ordered composition scan of affine maps z -> a*z + b across the threads of a
// workgroup (and, with the same routine, across chunk aggregates).
//
// Every recurrence of the limiter (limiter/hyrax.py:43-75: filtfilt attack
// smoother, Butterworth hold and release low-passes) is first order, so a run of
// samples acts on the carried filter state as one affine map.  Threads own
// contiguous runs, compute their map with a zero carry, and this scan hands each
// thread the composition of everything before it.  State is float64 throughout:
// the release pole is 0.99996.
//
// Written as barrier-separated phases over an LDS scratch area (mgx_hd.h) so the
// CPU emulation can run it; K independent scans share the barriers.
#pragma once

#include "mgx_hd.h"

namespace mgx {

struct Affine {
    double a, b;
};
MGX_HD Affine affine_identity() { return Affine{1.0, 0.0}; }
// apply `first`, then `second`
MGX_HD Affine affine_then(Affine first, Affine second) {
    return Affine{second.a * first.a, fma(second.a, first.b, second.b)};
}
MGX_HD double affine_apply(Affine m, double z) { return fma(m.a, z, m.b); }

// T threads, groups of G, K simultaneous scans.  Scratch: K*(T + T/G + 1) Affine.
template <int T, int G, int K>
struct WgScan {
    static constexpr int NG = T / G;
    static constexpr int SCRATCH = K * (T + NG + 1);   // in units of Affine
    static MGX_HD Affine* entries(Affine* s, int k) { return s + k * T; }
    static MGX_HD Affine* groups(Affine* s, int k) { return s + K * T + k * NG; }
    static MGX_HD Affine* total(Affine* s, int k) { return s + K * T + K * NG + k; }

    // phase 1: slot = position in scan order (tid, or T-1-tid for a right-to-left scan)
    static MGX_HD void put(Affine* s, int k, int slot, Affine m) { entries(s, k)[slot] = m; }
    // phase 2 (after barrier)
    static MGX_HD void scan_groups(Affine* s, int tid) {
        if (tid < K * NG) {
            const int k = tid / NG, g = tid % NG;
            Affine* e = entries(s, k) + g * G;
            Affine run = affine_identity();
            for (int i = 0; i < G; ++i) {
                const Affine m = e[i];
                e[i] = run;
                run = affine_then(run, m);
            }
            groups(s, k)[g] = run;
        }
    }
    // phase 3 (after barrier)
    static MGX_HD void scan_top(Affine* s, int tid) {
        if (tid < K) {
            Affine* gr = groups(s, tid);
            Affine run = affine_identity();
            for (int g = 0; g < NG; ++g) {
                const Affine m = gr[g];
                gr[g] = run;
                run = affine_then(run, m);
            }
            *total(s, tid) = run;
        }
    }
    // phase 4 (after barrier): composition of all slots before `slot`
    static MGX_HD Affine prefix(const Affine* s, int k, int slot) {
        const Affine* base = s;
        return affine_then(base[K * T + k * NG + slot / G], base[k * T + slot]);
    }
    static MGX_HD Affine whole(const Affine* s, int k) { return s[K * T + K * NG + k]; }
};

}  // namespace mgx
