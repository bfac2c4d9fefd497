#include "fir_design.h"
#include "fir_plan.h"

#include <algorithm>
#include <cmath>
#include <complex>

namespace mgx {

void cubic_spline_nak(const double* x, const double* y, int n, const double* xq, int nq, double* out) {
    // second derivatives m[0..n) from the classical tridiagonal system with the
    // not-a-knot closure folded into the first and last interior rows
    std::vector<double> h(n - 1), m(n, 0.0);
    for (int i = 0; i < n - 1; ++i) h[i] = x[i + 1] - x[i];
    const int k = n - 2;                       // interior unknowns m[1..n-2]
    std::vector<double> lo(k, 0.0), di(k, 0.0), up(k, 0.0), rhs(k, 0.0);
    for (int r = 0; r < k; ++r) {
        const int i = r + 1;
        lo[r] = h[i - 1];
        di[r] = 2.0 * (h[i - 1] + h[i]);
        up[r] = h[i];
        rhs[r] = 6.0 * ((y[i + 1] - y[i]) / h[i] - (y[i] - y[i - 1]) / h[i - 1]);
    }
    {
        const double h0 = h[0], h1 = h[1];
        di[0] = (h0 + h1) * (h0 + 2.0 * h1) / h1;
        up[0] = (h1 - h0) * (h1 + h0) / h1;
        const double a = h[n - 3], b = h[n - 2];
        di[k - 1] = (a + b) * (2.0 * a + b) / a;
        lo[k - 1] = (a - b) * (a + b) / a;
    }
    for (int r = 1; r < k; ++r) {              // Thomas elimination
        const double w = lo[r] / di[r - 1];
        di[r] -= w * up[r - 1];
        rhs[r] -= w * rhs[r - 1];
    }
    m[k] = rhs[k - 1] / di[k - 1];
    for (int r = k - 2; r >= 0; --r) m[r + 1] = (rhs[r] - up[r] * m[r + 2]) / di[r];
    m[0] = (1.0 + h[0] / h[1]) * m[1] - (h[0] / h[1]) * m[2];
    m[n - 1] = (1.0 + h[n - 2] / h[n - 3]) * m[n - 2] - (h[n - 2] / h[n - 3]) * m[n - 3];

    for (int q = 0; q < nq; ++q) {
        const double v = xq[q];
        int i = (int)(std::upper_bound(x, x + n, v) - x) - 1;
        i = std::min(std::max(i, 0), n - 2);
        const double hi = h[i], a = x[i + 1] - v, b = v - x[i];
        out[q] = m[i] * a * a * a / (6.0 * hi) + m[i + 1] * b * b * b / (6.0 * hi) +
                 (y[i] / hi - m[i] * hi / 6.0) * a + (y[i + 1] / hi - m[i + 1] * hi / 6.0) * b;
    }
}

static std::vector<double> linspace(double start, double stop, int n) {
    // numpy.linspace: k*step + start with the last point pinned to stop
    std::vector<double> v(n);
    const double step = (stop - start) / (n - 1);
    for (int k = 0; k < n; ++k) v[k] = k * step + start;
    v[n - 1] = stop;
    return v;
}

void lowess(const double* y, int n, double frac, double delta, int it, double* fit) {
    const std::vector<double> x = linspace(0.0, 1.0, n);
    int k = (int)(frac * n + 1e-10);
    k = std::min(std::max(k, 2), n);
    std::vector<double> w(n), robust(n, 1.0);
    for (int pass = 0; pass <= it; ++pass) {
        int i = 0, last = -1, lo = 0, hi = k;
        while (true) {
            while (hi < n && x[i] > (x[lo] + x[hi]) / 2.0) { ++lo; ++hi; }
            const double radius = std::max(x[i] - x[lo], x[hi - 1] - x[i]);
            double sw = 0.0;
            int nonzero = 0;
            for (int j = lo; j < hi; ++j) {
                const double d = std::fabs(x[j] - x[i]) / radius;
                const double t = 1.0 - d * d * d;
                w[j] = t * t * t * robust[j];
                sw += w[j];
                nonzero += w[j] != 0.0;
            }
            if (sw <= 0.0 || nonzero == 1) {
                fit[i] = y[i];
            } else {
                double xbar = 0.0;
                for (int j = lo; j < hi; ++j) { w[j] /= sw; xbar += w[j] * x[j]; }
                double dev = 0.0;
                for (int j = lo; j < hi; ++j) dev += w[j] * (x[j] - xbar) * (x[j] - xbar);
                double acc = 0.0;
                for (int j = lo; j < hi; ++j)
                    acc += w[j] * (1.0 + (x[i] - xbar) * (x[j] - xbar) / dev) * y[j];
                fit[i] = acc;
            }
            if (last < i - 1) {
                const double denom = x[i] - x[last];
                for (int j = last + 1; j < i; ++j) {
                    const double a = (x[j] - x[last]) / denom;
                    fit[j] = a * fit[i] + (1.0 - a) * fit[last];
                }
            }
            last = i;
            const double cut = x[last] + delta;
            int kk = last;
            for (kk = last + 1; kk < n; ++kk) {
                if (x[kk] > cut) break;
                if (x[kk] == x[last]) { fit[kk] = fit[last]; last = kk; }
            }
            if (kk >= n) kk = n - 1;               // loop ran off the end: Python leaves kk = n-1
            i = std::max(kk - 1, last + 1);
            if (last >= n - 1) break;
        }
        if (pass == it) break;
        // robustness weights of the next pass: bisquare(|residual| / (6 median |residual|))
        std::vector<double> r(n);
        for (int j = 0; j < n; ++j) r[j] = std::fabs(y[j] - fit[j]);
        std::vector<double> sorted(r);
        std::sort(sorted.begin(), sorted.end());
        const double median = n & 1 ? sorted[n / 2] : 0.5 * (sorted[n / 2 - 1] + sorted[n / 2]);
        for (int j = 0; j < n; ++j) {
            double u = median == 0.0 ? (r[j] > 0.0 ? 1.0 : 0.0) : r[j] / (6.0 * median);
            u = std::min(u, 1.0);
            robust[j] = (1.0 - u * u) * (1.0 - u * u);
        }
    }
}
void lowess_it0(const double* y, int n, double frac, double delta, double* fit) { lowess(y, n, frac, delta, 0, fit); }

void smooth_matching_curve(const double* curve, const FirDesignParams& p, double* smooth) {
    const int half = p.fft_size / 2;
    const int nlin = half + 1, nlog = half * p.lin_log_oversampling + 1;
    const double nyq = p.sample_rate * 0.5;
    std::vector<double> g_lin = linspace(0.0, 1.0, nlin);
    for (double& v : g_lin) v *= nyq;
    std::vector<double> g_log = linspace(std::log10(4.0 / p.fft_size), 0.0, nlog);
    for (double& v : g_log) v = nyq * std::pow(10.0, v);
    std::vector<double> on_log(nlog), on_log_s(nlog);
    cubic_spline_nak(g_lin.data(), curve, nlin, g_log.data(), nlog, on_log.data());
    lowess(on_log.data(), nlog, p.lowess_frac, p.lowess_delta, p.lowess_it, on_log_s.data());
    cubic_spline_nak(g_log.data(), on_log_s.data(), nlog, g_lin.data(), nlin, smooth);
    smooth[0] = 0.0;                            // match_frequencies.py:72-73
    smooth[1] = curve[1];
}

// in-place radix-2 complex FFT, sign = -1 forward / +1 inverse (unnormalised)
static void fft_pow2(std::vector<std::complex<double>>& a, int sign) {
    const int n = (int)a.size();
    for (int i = 1, j = 0; i < n; ++i) {
        int bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) std::swap(a[i], a[j]);
    }
    const double pi = 3.14159265358979323846;
    for (int len = 2; len <= n; len <<= 1) {
        const int halfl = len / 2;
        for (int i = 0; i < n; i += len) {
            for (int k = 0; k < halfl; ++k) {
                const double ang = sign * 2.0 * pi * k / len;
                const std::complex<double> w(std::cos(ang), std::sin(ang));
                const std::complex<double> u = a[i + k], v = a[i + k + halfl] * w;
                a[i + k] = u + v;
                a[i + k + halfl] = u - v;
            }
        }
    }
}

void design_fir(const double* avg_target, const double* avg_reference, const FirDesignParams& p,
                double* taps, double* curve_raw, double* curve_smooth) {
    FirPlanHost::get(p)->design(avg_target, avg_reference, 1.0, p.lowess_it, taps, curve_raw, curve_smooth);
}

void design_fir_direct(const double* avg_target, const double* avg_reference, const FirDesignParams& p,
                       double* taps, double* curve_raw, double* curve_smooth) {
    const int f = p.fft_size, half = f / 2;
    std::vector<double> raw(half + 1), smooth(half + 1);
    for (int k = 0; k <= half; ++k)
        raw[k] = avg_reference[k] / std::max(p.min_value, avg_target[k]);    // :93-94
    smooth_matching_curve(raw.data(), p, smooth.data());
    // numpy.fft.irfft of a real half spectrum (Hermitian extension, imaginary parts zero)
    std::vector<std::complex<double>> spec(f);
    for (int k = 0; k <= half; ++k) spec[k] = smooth[k];
    for (int k = 1; k < half; ++k) spec[f - k] = smooth[k];
    fft_pow2(spec, +1);
    const double pi = 3.14159265358979323846;
    for (int i = 0; i < f; ++i) {
        const double t = spec[(i + half) % f].real() / f;                     // ifftshift
        const double w = 0.5 - 0.5 * std::cos(2.0 * pi * i / (f - 1));       // symmetric Hann
        taps[i] = t * w;
    }
    if (curve_raw) std::copy(raw.begin(), raw.end(), curve_raw);
    if (curve_smooth) std::copy(smooth.begin(), smooth.end(), curve_smooth);
}

}  // namespace mgx
