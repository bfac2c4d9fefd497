// Matching-EQ FIR design as fixed linear operators + two tridiagonal scans.
//
// Everything between the averaged spectra and the FIR taps
// (matchering/stage_helpers/match_frequencies.py:45-101) is, for a given Config,
// a chain of LINEAR maps with data-independent structure:
//
//   raw    = A_ref / max(eps, c0*A_tgt)                       (the only non-linear step)
//   log    = S1(raw)      not-a-knot cubic spline, lin grid -> log grid   (interp1d "cubic")
//   smooth = L(log)       LOWESS, it=0: for a fixed x grid the tricube neighbourhoods,
//                         weights and the skipped-point interpolation depend on x only,
//                         so L is a fixed sparse matrix (one row of <= k weights per anchor)
//   lin    = S2(smooth)   spline, log grid -> lin grid; bins 0 and 1 pinned
//   taps   = hann * ifftshift(irfft(lin))
//
// A spline application = right-hand side (second differences) -> tridiagonal solve
// with a matrix that depends on the knots only (LU factors precomputed, so the solve
// is two first-order recurrences = two affine scans) -> per-query 4-term combination
// with precomputed coefficients.  FirPlan holds all the precomputed tables; it is
// built once per Config on the host (fir_plan.cpp) and uploaded.  The phase
// functions below run it: on the GPU as one 1024-thread workgroup per channel
// (k_fir_design), on the host (mgx_design_fir, CPU tests) as the same phases in a
// loop.  float64 throughout.
#pragma once

#include "mgx_hd.h"
#include "scan_util.h"

#include <memory>

namespace mgx {

struct SplineTables {
    int n;                  // knots
    int nq;                 // queries
    const double* h;        // [n-1] knot spacings
    const double* w;        // [n-2] LU: multiplier of the previous row (w[0] unused)
    const double* inv_di;   // [n-2] LU: 1 / pivot
    const double* up;       // [n-2] super-diagonal
    double e0a, e0b, e1a, e1b;   // not-a-knot closures: m[0] = e0a*m[1] + e0b*m[2], m[n-1] = e1a*m[n-2] + e1b*m[n-3]
    const int* qi;          // [nq] interval of each query
    const double* qc;       // [nq][4] coefficients of m[i], m[i+1], y[i], y[i+1]
};

struct LowessTables {
    int n;                  // points (= log grid size)
    int anchors;            // fitted points
    int k;                  // row length (neighbourhood size)
    const int* lo;          // [anchors] first neighbour of each anchor
    const int* pos;         // [anchors] the anchor's own point
    double step;            // x[j] = j * step (x = numpy.linspace(0, 1, n)), x[n-1] = 1
    const double* p;        // [anchors][k] regression weights (zero padded), robustness weights all 1
    const int* a0;          // [n] anchor index left of (or at) each point
    const int* a1;          // [n] anchor index right of (or at) each point
    const double* alpha;    // [n] interpolation weight of a1
};

struct FirPlanView {
    int fft, bins, nlog;
    double min_value;
    SplineTables s1, s2;
    LowessTables lw;
    const double* cos_table;   // [fft] cos(2 pi j / fft)
    const double* hann;        // [fft] symmetric Hann window
};

// scratch of one channel (global memory on the GPU)
struct FirScratch {
    double* raw;      // [bins]
    double* m1;       // [bins]   also holds the forward-substituted rhs
    double* on_log;   // [nlog]
    double* fit;      // [anchors]
    double* log_s;    // [nlog]
    double* m2;       // [nlog]
    double* smooth;   // [bins]
};

struct FirDesign {
    static constexpr int T = 1024;
    static constexpr int G = 32;
    using Scan = WgScan<T, G, 1>;

    // ---- raw matching curve (match_frequencies.py:93-94) ---------------------------
    static MGX_HD void phase_raw(int tid, const FirPlanView& pl, const double* avg_target, const double* avg_reference,
                                 double target_gain, FirScratch& s) {
        for (int k = tid; k < pl.bins; k += T)
            s.raw[k] = avg_reference[k] / fmax(pl.min_value, avg_target[k] * target_gain);
    }

    // ---- tridiagonal solve, forward substitution: d'[r] = rhs[r] - w[r]*d'[r-1] ---------
    static MGX_HD double rhs_at(const SplineTables& sp, const double* y, int r) {
        const int i = r + 1;
        return 6.0 * ((y[i + 1] - y[i]) / sp.h[i] - (y[i] - y[i - 1]) / sp.h[i - 1]);
    }
    static MGX_HD void rows_of(int tid, int k, int& r0, int& r1) {
        const int per = (k + T - 1) / T;
        r0 = tid * per;
        r1 = r0 + per < k ? r0 + per : k;
        if (r0 > k) r0 = k;
    }
    static MGX_HD void phase_fwd_local(int tid, const SplineTables& sp, const double* y, Affine* sc) {
        int r0, r1;
        rows_of(tid, sp.n - 2, r0, r1);
        Affine m = affine_identity();
        for (int r = r0; r < r1; ++r)
            m = affine_then(m, Affine{r == 0 ? 0.0 : -sp.w[r], rhs_at(sp, y, r)});
        Scan::put(sc, 0, tid, m);
    }
    static MGX_HD void phase_fwd_apply(int tid, const SplineTables& sp, const double* y, const Affine* sc, double* m) {
        int r0, r1;
        rows_of(tid, sp.n - 2, r0, r1);
        double d = affine_apply(Scan::prefix(sc, 0, tid), 0.0);
        for (int r = r0; r < r1; ++r) {
            d = fma(r == 0 ? 0.0 : -sp.w[r], d, rhs_at(sp, y, r));
            m[r + 1] = d;                               // d' stored where m[r+1] will live
        }
    }
    // ---- back substitution: m[r+1] = (d'[r] - up[r]*m[r+2]) * inv_di[r], right to left ----
    static MGX_HD void phase_bwd_local(int tid, const SplineTables& sp, const double* m, Affine* sc) {
        int r0, r1;
        rows_of(tid, sp.n - 2, r0, r1);
        const int k = sp.n - 2;
        Affine mm = affine_identity();
        for (int r = r1 - 1; r >= r0; --r) {
            const double a = r == k - 1 ? 0.0 : -sp.up[r] * sp.inv_di[r];
            mm = affine_then(mm, Affine{a, m[r + 1] * sp.inv_di[r]});
        }
        Scan::put(sc, 0, T - 1 - tid, mm);
    }
    static MGX_HD void phase_bwd_apply(int tid, const SplineTables& sp, const Affine* sc, double* m) {
        int r0, r1;
        rows_of(tid, sp.n - 2, r0, r1);
        const int k = sp.n - 2;
        double v = affine_apply(Scan::prefix(sc, 0, T - 1 - tid), 0.0);
        for (int r = r1 - 1; r >= r0; --r) {
            const double a = r == k - 1 ? 0.0 : -sp.up[r] * sp.inv_di[r];
            v = fma(a, v, m[r + 1] * sp.inv_di[r]);
            m[r + 1] = v;
        }
    }
    static MGX_HD void phase_closure(int tid, const SplineTables& sp, double* m) {
        if (tid == 0) {
            m[0] = sp.e0a * m[1] + sp.e0b * m[2];
            m[sp.n - 1] = sp.e1a * m[sp.n - 2] + sp.e1b * m[sp.n - 3];
        }
    }
    static MGX_HD void phase_eval(int tid, const SplineTables& sp, const double* y, const double* m, double* out) {
        for (int q = tid; q < sp.nq; q += T) {
            const int i = sp.qi[q];
            const double* c = sp.qc + 4 * (size_t)q;
            out[q] = c[0] * m[i] + c[1] * m[i + 1] + c[2] * y[i] + c[3] * y[i + 1];
        }
    }

    // ---- LOWESS as a fixed operator ----------------------------------------------------
    // one anchor per (emulated) wave lane group: lane-strided partial sums, summed in lane order
    static MGX_HD void phase_lowess_fit(int tid, const LowessTables& lw, const double* y, double* fit) {
        // thread-per-anchor on the host emulation; the GPU kernel overrides this with a
        // wave-per-anchor version (k_fir_design) that produces the same sums in a fixed tree
        for (int a = tid; a < lw.anchors; a += T) {
            const double* p = lw.p + (size_t)a * lw.k;
            const double* yy = y + lw.lo[a];
            double acc = 0.0;
            for (int j = 0; j < lw.k; ++j) acc = fma(p[j], yy[j], acc);      // lo + k <= n always
            fit[a] = acc;
        }
    }
    static MGX_HD void phase_lowess_fill(int tid, const LowessTables& lw, const double* fit, double* out) {
        for (int q = tid; q < lw.n; q += T) {
            const double al = lw.alpha[q];
            out[q] = al * fit[lw.a1[q]] + (1.0 - al) * fit[lw.a0[q]];
        }
    }

    // ---- LOWESS with robustness iterations (lowess_it > 0): the regression weights depend on the data --
    // statsmodels' _smoothers_lowess.pyx restated (oracle/mastering_oracle.py: lowess): tricube weights
    // times the robustness weights, the local linear fit in closed form; one anchor per thread, the
    // neighbourhood walked four times (sum of weights, weighted mean, weighted spread, fit) instead of
    // keeping k weights per thread.
    static MGX_HD double lowess_x(const LowessTables& lw, int j) { return j == lw.n - 1 ? 1.0 : j * lw.step; }
    static MGX_HD void phase_robust_init(int tid, int n, double* robust) {
        for (int q = tid; q < n; q += T) robust[q] = 1.0;
    }
    static MGX_HD void phase_lowess_fit_robust(int tid, const LowessTables& lw, const double* y, const double* robust,
                                               double* fit) {
        for (int a = tid; a < lw.anchors; a += T) {
            const int i = lw.pos[a], lo = lw.lo[a], hi = lo + lw.k;
            const double xi = lowess_x(lw, i);
            const double radius = fmax(xi - lowess_x(lw, lo), lowess_x(lw, hi - 1) - xi);
            auto weight = [&](int j) {
                const double d = fabs(lowess_x(lw, j) - xi) / radius;
                const double t = 1.0 - d * d * d;
                return t * t * t * robust[j];
            };
            double sw = 0.0;
            int nonzero = 0;
            for (int j = lo; j < hi; ++j) {
                const double w = weight(j);
                sw += w;
                nonzero += w != 0.0;
            }
            if (sw <= 0.0 || nonzero == 1) {
                fit[a] = y[i];
                continue;
            }
            double xbar = 0.0, dev = 0.0, acc = 0.0;
            for (int j = lo; j < hi; ++j) xbar += weight(j) / sw * lowess_x(lw, j);
            for (int j = lo; j < hi; ++j) {
                const double c = lowess_x(lw, j) - xbar;
                dev += weight(j) / sw * c * c;
            }
            for (int j = lo; j < hi; ++j)
                acc += weight(j) / sw * (1.0 + (xi - xbar) * (lowess_x(lw, j) - xbar) / dev) * y[j];
            fit[a] = acc;
        }
    }
    static MGX_HD void phase_residuals(int tid, int n, const double* y, const double* smooth, double* resid) {
        for (int q = tid; q < n; q += T) resid[q] = fabs(y[q] - smooth[q]);
    }
    // bisquare of |residual| / (6 median), clipped at 1; a zero median leaves weights 1 (exact fit) or 0
    static MGX_HD void phase_robust_weights(int tid, int n, const double* resid, double median, double* robust) {
        for (int q = tid; q < n; q += T) {
            double r = median == 0.0 ? (resid[q] > 0.0 ? 1.0 : 0.0) : resid[q] / (6.0 * median);
            r = fmin(r, 1.0);
            const double t = 1.0 - r * r;
            robust[q] = t * t;
        }
    }

    // ---- pinning + inverse real FFT + shift + window (match_frequencies.py:72-73,98-99) ----
    static MGX_HD void phase_pin(int tid, FirScratch& s) {
        if (tid == 0) {
            s.smooth[0] = 0.0;
            s.smooth[1] = s.raw[1];
        }
    }
    // taps[i] = hann[i] * t[(i + F/2) mod F],  t = irfft(smooth)
    static MGX_HD void phase_taps(int tid, const FirPlanView& pl, const double* smooth, float* taps_f32, double* taps_f64) {
        const int f = pl.fft, half = f / 2;
        for (int i = tid; i < f; i += T) {
            const int mm = (i + half) & (f - 1);
            double acc = 0.0;
            int idx = mm;                                  // (k*mm) mod f, advanced incrementally
            for (int k = 1; k < half; ++k) {
                acc = fma(smooth[k], pl.cos_table[idx], acc);
                idx = (idx + mm) & (f - 1);
            }
            const double t = (smooth[0] + ((mm & 1) ? -smooth[half] : smooth[half]) + 2.0 * acc) / f;
            const double v = t * pl.hann[i];
            if (taps_f32) taps_f32[i] = (float)v;
            if (taps_f64) taps_f64[i] = v;
        }
    }
};

struct FirDesignParams;
// Host owner of the tables: one contiguous blob (uploaded verbatim) + the offsets into it.
class FirPlanHost {
public:
    explicit FirPlanHost(const FirDesignParams& p);
    ~FirPlanHost();
    FirPlanHost(const FirPlanHost&) = delete;
    FirPlanHost& operator=(const FirPlanHost&) = delete;
    static std::shared_ptr<FirPlanHost> get(const FirDesignParams& p);     // cached per parameter set
    const void* blob() const;
    size_t blob_bytes() const;
    FirPlanView view(const void* base) const;      // base = blob() on the host or its device copy
    int bins() const;
    int nlog() const;
    int anchors() const;
    // runs the phases on the host: taps[F] float64 (target_gain multiplies avg_target)
    void design(const double* avg_target, const double* avg_reference, double target_gain, int lowess_it, double* taps,
                double* curve_raw, double* curve_smooth) const;

private:
    struct Impl;
    Impl* impl_;
};

}  // namespace mgx
