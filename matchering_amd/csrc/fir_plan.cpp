#include "fir_plan.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <tuple>

#include "fir_design.h"

namespace mgx {

namespace {

std::vector<double> np_linspace(double start, double stop, int n) {
    std::vector<double> v(n);
    const double step = (stop - start) / (n - 1);
    for (int k = 0; k < n; ++k) v[k] = k * step + start;
    v[n - 1] = stop;
    return v;
}

struct Blob {
    std::vector<char> bytes;
    template <typename T>
    size_t put(const std::vector<T>& v) {
        const size_t off = (bytes.size() + 15) & ~(size_t)15;
        bytes.resize(off + v.size() * sizeof(T));
        std::memcpy(bytes.data() + off, v.data(), v.size() * sizeof(T));
        return off;
    }
};

struct SplineOffsets {
    int n, nq;
    size_t h, w, inv_di, up, qi, qc;
    double e0a, e0b, e1a, e1b;
};

SplineOffsets build_spline(Blob& blob, const std::vector<double>& x, const std::vector<double>& xq) {
    const int n = (int)x.size(), k = n - 2, nq = (int)xq.size();
    std::vector<double> h(n - 1), lo(k), di(k), up(k), w(k, 0.0), inv_di(k);
    for (int i = 0; i < n - 1; ++i) h[i] = x[i + 1] - x[i];
    for (int r = 0; r < k; ++r) {
        const int i = r + 1;
        lo[r] = h[i - 1];
        di[r] = 2.0 * (h[i - 1] + h[i]);
        up[r] = h[i];
    }
    {   // not-a-knot closure folded into the first and last interior rows
        const double h0 = h[0], h1 = h[1];
        di[0] = (h0 + h1) * (h0 + 2.0 * h1) / h1;
        up[0] = (h1 - h0) * (h1 + h0) / h1;
        const double a = h[n - 3], b = h[n - 2];
        di[k - 1] = (a + b) * (2.0 * a + b) / a;
        lo[k - 1] = (a - b) * (a + b) / a;
    }
    inv_di[0] = 1.0 / di[0];
    for (int r = 1; r < k; ++r) {
        w[r] = lo[r] / di[r - 1];
        di[r] -= w[r] * up[r - 1];
        inv_di[r] = 1.0 / di[r];
    }
    std::vector<int> qi(nq);
    std::vector<double> qc((size_t)nq * 4);
    for (int q = 0; q < nq; ++q) {
        const double v = xq[q];
        int i = (int)(std::upper_bound(x.begin(), x.end(), v) - x.begin()) - 1;
        i = std::min(std::max(i, 0), n - 2);
        const double hi = h[i], a = x[i + 1] - v, b = v - x[i];
        qi[q] = i;
        qc[4 * q + 0] = a * a * a / (6.0 * hi) - hi * a / 6.0;
        qc[4 * q + 1] = b * b * b / (6.0 * hi) - hi * b / 6.0;
        qc[4 * q + 2] = a / hi;
        qc[4 * q + 3] = b / hi;
    }
    SplineOffsets o;
    o.n = n;
    o.nq = nq;
    o.h = blob.put(h);
    o.w = blob.put(w);
    o.inv_di = blob.put(inv_di);
    o.up = blob.put(up);
    o.qi = blob.put(qi);
    o.qc = blob.put(qc);
    o.e0a = 1.0 + h[0] / h[1];
    o.e0b = -h[0] / h[1];
    o.e1a = 1.0 + h[n - 2] / h[n - 3];
    o.e1b = -h[n - 2] / h[n - 3];
    return o;
}

struct LowessOffsets {
    int n, anchors, k;
    size_t lo, pos, p, a0, a1, alpha;
    double step;
};

// The index walk of LOWESS (it = 0) on x = linspace(0,1,n): which points get a regression,
// with which neighbourhood and weights, and how the skipped points are interpolated.
LowessOffsets build_lowess(Blob& blob, int n, double frac, double delta) {
    const std::vector<double> x = np_linspace(0.0, 1.0, n);
    int k = (int)(frac * n + 1e-10);
    k = std::min(std::max(k, 2), n);
    std::vector<int> los, poss, a0(n, 0), a1(n, 0);
    std::vector<double> rows, alpha(n, 0.0), w(k);
    int i = 0, last = -1, lo = 0, hi = k, last_anchor = -1;
    while (true) {
        while (hi < n && x[i] > (x[lo] + x[hi]) / 2.0) { ++lo; ++hi; }
        const double radius = std::max(x[i] - x[lo], x[hi - 1] - x[i]);
        double sw = 0.0;
        int nonzero = 0;
        for (int j = 0; j < k; ++j) {
            const double d = std::fabs(x[lo + j] - x[i]) / radius;
            const double t = 1.0 - d * d * d;
            w[j] = t * t * t;
            sw += w[j];
            nonzero += w[j] != 0.0;
        }
        const int anchor = (int)los.size();
        los.push_back(lo);
        poss.push_back(i);
        const size_t row = rows.size();
        rows.resize(row + k, 0.0);
        if (sw <= 0.0 || nonzero == 1) {
            rows[row + (i - lo)] = 1.0;                       // fit = y[i]
        } else {
            double xbar = 0.0, dev = 0.0;
            for (int j = 0; j < k; ++j) { w[j] /= sw; xbar += w[j] * x[lo + j]; }
            for (int j = 0; j < k; ++j) dev += w[j] * (x[lo + j] - xbar) * (x[lo + j] - xbar);
            for (int j = 0; j < k; ++j)
                rows[row + j] = w[j] * (1.0 + (x[i] - xbar) * (x[lo + j] - xbar) / dev);
        }
        if (last < i - 1) {
            const double denom = x[i] - x[last];
            for (int j = last + 1; j < i; ++j) {
                a0[j] = last_anchor;
                a1[j] = anchor;
                alpha[j] = (x[j] - x[last]) / denom;
            }
        }
        a0[i] = a1[i] = anchor;
        alpha[i] = 0.0;
        last = i;
        last_anchor = anchor;
        const double cut = x[last] + delta;
        int kk = last;
        for (kk = last + 1; kk < n; ++kk) {
            if (x[kk] > cut) break;
            if (x[kk] == x[last]) { a0[kk] = a1[kk] = anchor; alpha[kk] = 0.0; last = kk; }
        }
        if (kk >= n) kk = n - 1;
        i = std::max(kk - 1, last + 1);
        if (last >= n - 1) break;
    }
    LowessOffsets o;
    o.n = n;
    o.anchors = (int)los.size();
    o.k = k;
    o.lo = blob.put(los);
    o.pos = blob.put(poss);
    o.step = (1.0 - 0.0) / (n - 1);
    o.p = blob.put(rows);
    o.a0 = blob.put(a0);
    o.a1 = blob.put(a1);
    o.alpha = blob.put(alpha);
    return o;
}

SplineTables spline_view(const SplineOffsets& o, const char* base) {
    SplineTables t;
    t.n = o.n;
    t.nq = o.nq;
    t.h = (const double*)(base + o.h);
    t.w = (const double*)(base + o.w);
    t.inv_di = (const double*)(base + o.inv_di);
    t.up = (const double*)(base + o.up);
    t.qi = (const int*)(base + o.qi);
    t.qc = (const double*)(base + o.qc);
    t.e0a = o.e0a;
    t.e0b = o.e0b;
    t.e1a = o.e1a;
    t.e1b = o.e1b;
    return t;
}

}  // namespace

struct FirPlanHost::Impl {
    FirDesignParams p;
    Blob blob;
    SplineOffsets s1, s2;
    LowessOffsets lw;
    size_t cos_table, hann;
    int bins, nlog;
};

FirPlanHost::FirPlanHost(const FirDesignParams& p) : impl_(new Impl) {
    Impl& m = *impl_;
    m.p = p;
    const int half = p.fft_size / 2;
    m.bins = half + 1;
    m.nlog = half * p.lin_log_oversampling + 1;
    const double nyq = p.sample_rate * 0.5;
    std::vector<double> g_lin = np_linspace(0.0, 1.0, m.bins);
    for (double& v : g_lin) v *= nyq;
    std::vector<double> g_log = np_linspace(std::log10(4.0 / p.fft_size), 0.0, m.nlog);
    for (double& v : g_log) v = nyq * std::pow(10.0, v);
    m.s1 = build_spline(m.blob, g_lin, g_log);
    m.s2 = build_spline(m.blob, g_log, g_lin);
    m.lw = build_lowess(m.blob, m.nlog, p.lowess_frac, p.lowess_delta);
    const double pi = 3.14159265358979323846;
    std::vector<double> ct(p.fft_size), hn(p.fft_size);
    for (int j = 0; j < p.fft_size; ++j) {
        ct[j] = std::cos(2.0 * pi * j / p.fft_size);
        hn[j] = 0.5 - 0.5 * std::cos(2.0 * pi * j / (p.fft_size - 1));
    }
    m.cos_table = m.blob.put(ct);
    m.hann = m.blob.put(hn);
}
FirPlanHost::~FirPlanHost() { delete impl_; }

const void* FirPlanHost::blob() const { return impl_->blob.bytes.data(); }
size_t FirPlanHost::blob_bytes() const { return impl_->blob.bytes.size(); }
int FirPlanHost::bins() const { return impl_->bins; }
int FirPlanHost::nlog() const { return impl_->nlog; }
int FirPlanHost::anchors() const { return impl_->lw.anchors; }

FirPlanView FirPlanHost::view(const void* base_ptr) const {
    const Impl& m = *impl_;
    const char* base = (const char*)base_ptr;
    FirPlanView v;
    v.fft = m.p.fft_size;
    v.bins = m.bins;
    v.nlog = m.nlog;
    v.min_value = m.p.min_value;
    v.s1 = spline_view(m.s1, base);
    v.s2 = spline_view(m.s2, base);
    v.lw.n = m.lw.n;
    v.lw.anchors = m.lw.anchors;
    v.lw.k = m.lw.k;
    v.lw.lo = (const int*)(base + m.lw.lo);
    v.lw.pos = (const int*)(base + m.lw.pos);
    v.lw.step = m.lw.step;
    v.lw.p = (const double*)(base + m.lw.p);
    v.lw.a0 = (const int*)(base + m.lw.a0);
    v.lw.a1 = (const int*)(base + m.lw.a1);
    v.lw.alpha = (const double*)(base + m.lw.alpha);
    v.cos_table = (const double*)(base + m.cos_table);
    v.hann = (const double*)(base + m.hann);
    return v;
}

std::shared_ptr<FirPlanHost> FirPlanHost::get(const FirDesignParams& p) {
    using Key = std::tuple<int, int, int, double, double, double>;
    static std::mutex mu;
    static std::map<Key, std::shared_ptr<FirPlanHost>> cache;
    const Key key(p.fft_size, p.sample_rate, p.lin_log_oversampling, p.lowess_frac, p.lowess_delta, p.min_value);
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    auto plan = std::make_shared<FirPlanHost>(p);
    cache[key] = plan;
    return plan;
}

// the device phases, run as a loop over thread ids (host path of mgx_design_fir, CPU tests)
void FirPlanHost::design(const double* avg_target, const double* avg_reference, double target_gain, int lowess_it,
                         double* taps, double* curve_raw, double* curve_smooth) const {
    using FD = FirDesign;
    const FirPlanView pl = view(blob());
    std::vector<double> raw(pl.bins), m1(pl.bins), on_log(pl.nlog), fit(pl.lw.anchors), log_s(pl.nlog),
        m2(pl.nlog), smooth(pl.bins);
    FirScratch s{raw.data(), m1.data(), on_log.data(), fit.data(), log_s.data(), m2.data(), smooth.data()};
    std::vector<Affine> sc(FD::Scan::SCRATCH);
#define ALL(stmt) for (int tid = 0; tid < FD::T; ++tid) { stmt; }
    ALL(FD::phase_raw(tid, pl, avg_target, avg_reference, target_gain, s))
    auto solve = [&](const SplineTables& sp, const double* y, double* m) {
        ALL(FD::phase_fwd_local(tid, sp, y, sc.data()))
        ALL(FD::Scan::scan_groups(sc.data(), tid))
        ALL(FD::Scan::scan_top(sc.data(), tid))
        ALL(FD::phase_fwd_apply(tid, sp, y, sc.data(), m))
        ALL(FD::phase_bwd_local(tid, sp, m, sc.data()))
        ALL(FD::Scan::scan_groups(sc.data(), tid))
        ALL(FD::Scan::scan_top(sc.data(), tid))
        ALL(FD::phase_bwd_apply(tid, sp, sc.data(), m))
        ALL(FD::phase_closure(tid, sp, m))
    };
    solve(pl.s1, s.raw, s.m1);
    ALL(FD::phase_eval(tid, pl.s1, s.raw, s.m1, s.on_log))
    if (lowess_it == 0) {
        ALL(FD::phase_lowess_fit(tid, pl.lw, s.on_log, s.fit))
        ALL(FD::phase_lowess_fill(tid, pl.lw, s.fit, s.log_s))
    } else {
        std::vector<double> robust(pl.nlog), resid(pl.nlog);
        ALL(FD::phase_robust_init(tid, pl.nlog, robust.data()))
        for (int pass = 0; pass <= lowess_it; ++pass) {
            ALL(FD::phase_lowess_fit_robust(tid, pl.lw, s.on_log, robust.data(), s.fit))
            ALL(FD::phase_lowess_fill(tid, pl.lw, s.fit, s.log_s))
            if (pass == lowess_it) break;
            ALL(FD::phase_residuals(tid, pl.nlog, s.on_log, s.log_s, resid.data()))
            std::vector<double> sorted(resid);                      // numpy.median: mean of the two middle values
            std::sort(sorted.begin(), sorted.end());
            const int n = pl.nlog;
            const double median = n & 1 ? sorted[n / 2] : 0.5 * (sorted[n / 2 - 1] + sorted[n / 2]);
            ALL(FD::phase_robust_weights(tid, pl.nlog, resid.data(), median, robust.data()))
        }
    }
    solve(pl.s2, s.log_s, s.m2);
    ALL(FD::phase_eval(tid, pl.s2, s.log_s, s.m2, s.smooth))
    ALL(FD::phase_pin(tid, s))
    ALL(FD::phase_taps(tid, pl, s.smooth, nullptr, taps))
#undef ALL
    if (curve_raw) std::copy(raw.begin(), raw.end(), curve_raw);
    if (curve_smooth) std::copy(smooth.begin(), smooth.end(), curve_smooth);
}

}  // namespace mgx
