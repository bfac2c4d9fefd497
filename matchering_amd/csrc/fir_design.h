// Matching-EQ FIR design on the host, float64.
//
// Follows matchering/stage_helpers/match_frequencies.py:45-101 (`get_fir` and
// `__smooth_exponentially`) from the averaged spectra onward: everything here is
// O(F) .. O(F log F) on ~2k..8k points, i.e. not bandwidth work, and it carries
// discrete structure (LOWESS neighbourhoods) that is simplest to keep in
// float64 on the CPU.  Third-party routines the reference calls are restated:
//   scipy.interpolate.interp1d(kind="cubic")  -> not-a-knot cubic spline
//   statsmodels lowess(frac, it, delta)       -> lowess (Cleveland's LOWESS with `it` robustness passes)
//   numpy.fft.irfft / ifftshift, scipy.signal.windows.hann (symmetric)
#pragma once

#include <vector>

namespace mgx {

struct FirDesignParams {
    int fft_size;               // F
    int sample_rate;
    int lin_log_oversampling;
    double lowess_frac;
    int lowess_it;              // robustness iterations (the reference's default is 0)
    double lowess_delta;
    double min_value;
};

// Not-a-knot cubic spline through (x[i], y[i]), i < n (n >= 4, x strictly increasing),
// evaluated at xq[0..nq) (points outside [x0, x_{n-1}] use the end polynomials).
void cubic_spline_nak(const double* x, const double* y, int n, const double* xq, int nq, double* out);

// LOWESS with zero robustness iterations on the index grid x = linspace(0,1,n).
void lowess(const double* y, int n, double frac, double delta, int it, double* fit);
void lowess_it0(const double* y, int n, double frac, double delta, double* fit);

// match_frequencies.py:45-75
void smooth_matching_curve(const double* curve, const FirDesignParams& p, double* smooth);

// match_frequencies.py:93-99.  avg_* have F/2+1 entries and are already expressed for
// the level-matched target / normalised reference.  taps gets F entries.  curve_raw /
// curve_smooth (F/2+1 each) may be null.  Runs the precomputed-operator plan (fir_plan.h),
// i.e. the same phases the GPU kernel runs.
void design_fir(const double* avg_target, const double* avg_reference, const FirDesignParams& p,
                double* taps, double* curve_raw, double* curve_smooth);

// The same design written out step by step with the restated third-party routines above
// (no precomputation); kept as an independent cross-check of the plan.
void design_fir_direct(const double* avg_target, const double* avg_reference, const FirDesignParams& p,
                       double* taps, double* curve_raw, double* curve_smooth);

}  // namespace mgx
