// Host-side derivation of kernel parameters from mgx_config (plain C++; shared
// by the HIP library and the CPU emulation harness).
#pragma once

#include <cstdio>

#include <algorithm>
#include <cmath>
#include <complex>
#include <string>

#include "../../include/mgx.h"
#include <vector>

#include "limiter_general.h"

namespace mgx {

// match_levels.py:47-59
inline void piece_geometry(long long n, double max_piece_size, int& divisions, long long& piece) {
    divisions = (int)((double)n / max_piece_size) + 1;
    piece = (long long)((double)n / (double)divisions);
}

// scipy.signal.butter(1, fc, fs=fs) -> transposed-direct-form-II section
inline Iir1 butter1(double fc, double fs) {
    const double pi = 3.14159265358979323846;
    const double k = std::tan(pi * fc / fs);
    const double b = k / (1.0 + k), a1 = (k - 1.0) / (k + 1.0);
    return Iir1{b, -a1, b - a1 * b};
}

// scipy.signal.butter(order, fc, fs=fs) (low-pass, transfer-function form): analogue prototype poles
// on the unit circle, scaled to the pre-warped cut-off, bilinear transform with fs = 2, numerator
// k (z + 1)^order.  b, a: order + 1 coefficients each.
inline void butter_tf(int order, double fc, double fs, double* b, double* a) {
    using cd = std::complex<double>;
    const double pi = 3.14159265358979323846;
    const double wn = 2.0 * fc / fs;
    const double warped = 4.0 * std::tan(pi * wn / 2.0);
    cd den = 1.0;
    std::vector<cd> poly(order + 1, cd(0.0));
    poly[0] = 1.0;
    for (int k = 0; k < order; ++k) {
        const int m = -order + 1 + 2 * k;
        const cd pa = -std::exp(cd(0.0, pi * m / (2.0 * order))) * warped;
        const cd pd = (4.0 + pa) / (4.0 - pa);
        den *= (4.0 - pa);
        for (int i = k + 1; i >= 1; --i) poly[i] -= pd * poly[i - 1];
    }
    const double kd = std::pow(warped, order) * std::real(1.0 / den);
    double binom = 1.0;
    for (int i = 0; i <= order; ++i) {
        a[i] = std::real(poly[i]);
        b[i] = kd * binom;
        binom = binom * (order - i) / (i + 1);
    }
}

// ---- the shifted basis of limiter_general.h ("state maps"), from a section's float64 coefficients ---------------
// order k <= K; d = 2 pi fc / fs.  With m = k-1-i, t = k-1-j (binomials C):
//     w_i = d^m sum_j C(k-1-j, m) z_j            z_j = sum_i d^-(k-1-i) C(k-1-i, t) (-1)^(k-1-i-t) w_i
//     w' = (I + dm) w + bd x:   dm[i][i+1] = d,   dm[i][0] -= d^-i S_{i+1},   S_s = sum_m a_m C(k-m, k-s)
//     bd_i = d^m sum_j C(k-1-j, m) B_j,   B_j = b_{j+1} - a_{j+1} b_0
// (S_k = p(1) ~ d^k: the sums cancel, which is why they are taken in extended precision from the float64 coefficients
// the reference really uses -- exact to 1e-19 absolute, 1e-10 of their size.)  States beyond the filter's own order
// (a lower-order filter carried in a K-state section) are identically zero: their rows and columns are left empty.
struct ShiftedBasis {
    std::vector<long double> step;      // [K][K]  I + dm on the leading k x k block, zero elsewhere
    std::vector<long double> dm, zw;    // [K][K]
    std::vector<long double> bd;        // [K]
};
inline long double binomial(int n, int r) {
    if (r < 0 || r > n) return 0.0L;
    long double v = 1.0L;
    for (int i = 0; i < r; ++i) v = v * (long double)(n - i) / (long double)(i + 1);
    return v;
}
inline ShiftedBasis shifted_basis(int k, int K, const double* b, const double* a, double fc, double fs) {
    using W = long double;
    const W d = 6.283185307179586476925286766559L * (W)fc / (W)fs;
    ShiftedBasis s;
    s.step.assign((size_t)K * K, 0.0L);
    s.dm.assign((size_t)K * K, 0.0L);
    s.zw.assign((size_t)K * K, 0.0L);
    s.bd.assign(K, 0.0L);
    auto dpow = [&](int e) { return std::pow(d, (W)e); };
    for (int i = 0; i < k; ++i) {
        const int m = k - 1 - i;
        if (i + 1 < k) s.dm[(size_t)i * K + i + 1] = d;
        W sum = 0.0L;                                         // S_{i+1}
        for (int q = 0; q <= k; ++q) sum += (W)a[q] * binomial(k - q, k - (i + 1));
        s.dm[(size_t)i * K] -= dpow(-i) * sum;
        W bsum = 0.0L;
        for (int j = 0; j < k; ++j) bsum += binomial(k - 1 - j, m) * ((W)b[j + 1] - (W)a[j + 1] * (W)b[0]);
        s.bd[i] = dpow(m) * bsum;
        for (int j = 0; j < k; ++j) {                         // z_j from w_i
            const int t = k - 1 - j;
            if (t <= m) s.zw[(size_t)j * K + i] = dpow(-m) * binomial(m, t) * (((m - t) & 1) ? -1.0L : 1.0L);
        }
    }
    for (int i = 0; i < k; ++i)
        for (int j = 0; j < k; ++j) s.step[(size_t)i * K + j] = (i == j ? 1.0L : 0.0L) + s.dm[(size_t)i * K + j];
    return s;
}
// an order-K section (limiter_general.h) from a filter of order <= K with its shifted-basis form
template <int K>
inline IirK<K> section_of(int order, const double* b, const double* a, const ShiftedBasis& sb) {
    IirK<K> f;
    for (int i = 0; i <= K; ++i) {
        f.b[i] = i <= order ? b[i] : 0.0;
        f.a[i] = i <= order ? a[i] : 0.0;
    }
    for (int i = 0; i < K; ++i) {
        f.bd[i] = (double)sb.bd[i];
        for (int j = 0; j < K; ++j) {
            f.dm[i][j] = (double)sb.dm[(size_t)i * K + j];
            f.zw[i][j] = (double)sb.zw[(size_t)i * K + j];
        }
    }
    return f;
}
// k x k matrices in extended precision (x87 long double on the host: 64-bit significands; the powers of
// a state matrix whose eigenvalues sit 1e-5 from 1 lose ~1e4 in conditioning)
using Wide = long double;
inline void mat_mul(int k, const std::vector<Wide>& x, const std::vector<Wide>& y, std::vector<Wide>& out) {
    std::vector<Wide> r((size_t)k * k, 0.0L);
    for (int i = 0; i < k; ++i)
        for (int j = 0; j < k; ++j) {
            Wide s = 0.0L;
            for (int l = 0; l < k; ++l) s += x[(size_t)i * k + l] * y[(size_t)l * k + j];
            r[(size_t)i * k + j] = s;
        }
    out = r;
}
inline std::vector<Wide> mat_identity(int k) {
    std::vector<Wide> r((size_t)k * k, 0.0L);
    for (int i = 0; i < k; ++i) r[(size_t)i * k + i] = 1.0L;
    return r;
}
inline std::vector<Wide> mat_power(int k, const std::vector<Wide>& m, long long e) {
    std::vector<Wide> r = mat_identity(k), base(m);
    while (e > 0) {
        if (e & 1) mat_mul(k, base, r, r);
        mat_mul(k, base, base, base);
        e >>= 1;
    }
    return r;
}
inline void append_rounded(std::vector<double>& out, const std::vector<Wide>& m) {
    for (Wide v : m) out.push_back((double)v);
}
// A^j for j = 0..16, [17][k][k]
inline std::vector<double> block_powers(int k, const std::vector<Wide>& m) {
    std::vector<double> out;
    std::vector<Wide> cur = mat_identity(k);
    for (int j = 0; j <= 16; ++j) {
        append_rounded(out, cur);
        mat_mul(k, m, cur, cur);
    }
    return out;
}
// (A^chunk)^m until its largest entry drops below 1e-13 (at least one, at most `cap`: empty = did not decay).
// (In the shifted basis the entries are O(1) times the decay -- no n^(K-1) factors as in the section's own basis, where
// 1e-10 used to be conservative -- and the states' own sizes differ by powers of d: three decades of margin.)
inline std::vector<double> lookback_matrices(int k, const std::vector<Wide>& m, int chunk, int cap) {
    const std::vector<Wide> step = mat_power(k, m, chunk);
    std::vector<double> out;
    std::vector<Wide> cur = mat_identity(k);
    for (int n = 0; n < cap; ++n) {
        Wide big = 0.0L;
        for (Wide v : cur) big = std::fmax(big, std::fabs(v));
        if (n > 0 && big <= 1e-13L) return out;
        append_rounded(out, cur);
        mat_mul(k, step, cur, cur);
    }
    return std::vector<double>();
}

// What is left of the backward attack smoother's state after the right halo, relative to the state (gains
// are <= 1, so this bounds the error of gA at the core's last frame; it falls off by rho per frame before it).
// 1e-7 is a tenth of the RMS bound the limiter tests hold (1e-6) and leaves 224 of a chunk's 256 blocks to the
// core at the default timings (1e-8: 221; 1e-6: 228, measured 1 % faster: profiles/r03_f_ab_limiter_quiet_chunks.txt).
constexpr double MGX_ATTACK_FORGET = 1e-7;

struct LimiterParams {
    int attack, hold, hw, hb, ha;
    // hold / release filters of order > 1 (limiter_general.h): general = max order, 0 when both are first order
    int general = 0;
    double hold_b[LIMITER_MAX_ORDER + 1], hold_a[LIMITER_MAX_ORDER + 1];
    double rel_b[LIMITER_MAX_ORDER + 1], rel_a[LIMITER_MAX_ORDER + 1];
    int hold_order = 1, rel_order = 1;
    std::vector<double> pow_hold, pow_rel, wk_hold, wk_rel;    // [17][K][K], [17][K][K], [n][K][K], [n][K][K] (shifted basis)
    ShiftedBasis hold_sb, rel_sb;
    Iir1 att, hold_f, rel_f;
    int threads;                                   // blocks per chunk the limiter kernel runs with: 256 or 1024
    LimiterBlock<256>::Geometry geo;               // (the same struct for every T)
    std::vector<double> w_hold, w_rel, w_att;      // look-back weights (alpha^chunk)^m
};

inline Iir1f to_f32(const Iir1& f) {
    Iir1f r;
    r.b0 = (float)f.b0;
    r.alpha = (float)f.alpha;
    r.beta = (float)f.beta;
    return r;
}
inline double power16(double alpha) {
    double p = 1.0;
    for (int j = 0; j < 16; ++j) p *= alpha;
    return p;
}
// (alpha^chunk)^m until it drops below 1e-10 (at least one entry, at most `cap`)
inline std::vector<double> lookback_weights(double alpha, int chunk, int cap) {
    const double ac = std::pow(alpha, (double)chunk);
    std::vector<double> w;
    double p = 1.0;
    while ((int)w.size() < cap && (w.empty() || p > 1e-10)) { w.push_back(p); p *= ac; }
    return w;
}

// blocks per chunk asked for from outside the parameter rule (0: none); set by mgx.hip (MGX_LIMIT_THREADS)
inline int& limiter_threads_wish() {
    static int wish = 0;
    return wish;
}
// utils.py:50-55, hyrax.py:43-75.  Returns an error text, empty when fine.
inline std::string limiter_params(const mgx_config& c, LimiterParams& p) {
    const double sr = c.internal_sample_rate;
    p.attack = (int)(sr * c.attack_ms * 1e-3);
    p.hold = (int)(sr * c.hold_ms * 1e-3);
    if (p.attack < 1) return "limiter attack shorter than one sample";
    if (p.hold < 3) return "limiter hold shorter than three samples (the reference's sliding window is empty there)";
    if (c.hold_filter_order < 1 || c.release_filter_order < 1) return "hold/release filter orders must be positive";
    // Routed by conditioning, not by order (limiter_general.h): the rounding noise of the reference's own float64
    // recursion, ~1.1e-16 / d^(order - 1/2) with d = 2 pi fc / fs the poles' distance from z = 1, must stay under 1e-6
    // of full scale -- above that there is no well-defined output to be within 1e-5 of.
    {
        const double two_pi = 6.283185307179586;
        const struct { const char* name; int order; double fc; } filters[2] = {
            {"hold", c.hold_filter_order, c.hold_filter_coefficient},
            {"release", c.release_filter_order, c.release_filter_coefficient / c.release_ms}};
        for (const auto& f : filters) {
            if (f.order == 1) continue;
            const double d = two_pi * f.fc / sr;
            const double noise = d > 0.0 ? 1.1e-16 / std::pow(d, f.order - 0.5) : 1.0;
            if (!(noise <= 1e-6)) {
                char text[320];
                std::snprintf(text, sizeof text,
                              "%s filter of order %d at %.4g Hz is ill-conditioned in the reference's transfer-function form: its own "
                              "float64 recursion carries rounding noise of ~%.1e of full scale (limit 1e-6; limiter_general.h) -- "
                              "refused rather than approximated", f.name, f.order, f.fc, noise);
                return text;
            }
            if (f.order > LIMITER_MAX_ORDER) {
                char text[200];
                std::snprintf(text, sizeof text, "%s filter order %d: the chunked limiter is built for orders up to %d", f.name,
                              f.order, LIMITER_MAX_ORDER);
                return text;
            }
        }
    }
    p.hold_order = c.hold_filter_order;
    p.rel_order = c.release_filter_order;
    p.general = (p.hold_order > 1 || p.rel_order > 1) ? (p.hold_order > p.rel_order ? p.hold_order : p.rel_order) : 0;
    const int w = (p.attack & 1) ? p.attack : p.attack + 1;
    p.hw = w - 1;
    p.hb = p.hold - 1;
    const double rho = std::exp(c.attack_filter_coefficient / p.attack);
    if (!(rho > 0.0 && rho < 1.0)) return "attack_filter_coefficient must be negative";
    p.att = Iir1{1.0 - rho, rho, rho * (1.0 - rho)};
    // (the first-order sections also serve the general path: its attack phases read p.att only, and the
    // geometry does not depend on them)
    p.hold_f = butter1(c.hold_filter_coefficient, sr);
    p.rel_f = butter1(c.release_filter_coefficient / c.release_ms, sr);
    // frames after which the attack smoother has forgotten its state (rho^ha <= MGX_ATTACK_FORGET)
    p.ha = (int)std::ceil(std::log(MGX_ATTACK_FORGET) / std::log(rho));
    // 256 blocks per chunk while the halos leave at least a quarter of them to the core, else 1024;
    // `limiter_threads_wish` (256 / 1024, 0 = this rule): what mgx.hip asks for (measurement aid)
    p.threads = 256;
    p.geo = LimiterBlock<256>::geometry(p.hw, p.hb, p.ha);
    int wish = limiter_threads_wish();
    if (p.geo.core_blocks < 64 && wish < 1024) wish = 1024;
    if (wish == 1024) {
        p.threads = 1024;
        const LimiterBlock<1024>::Geometry g = LimiterBlock<1024>::geometry(p.hw, p.hb, p.ha);
        p.geo.gl = g.gl; p.geo.gr = g.gr; p.geo.gw = g.gw; p.geo.core_blocks = g.core_blocks; p.geo.chunk = g.chunk;
    }
    if (p.geo.core_blocks < 64)
        return "limiter attack/hold times too long for the chunked kernel (halos of more than 960 blocks of 16 frames)";
    if (!(p.hold_f.alpha > 0.0 && p.hold_f.alpha < 1.0 && p.rel_f.alpha > 0.0 && p.rel_f.alpha < 1.0))
        return "hold/release filter is not a stable low-pass";
    p.w_hold = lookback_weights(p.hold_f.alpha, p.geo.chunk, 1 << 16);
    p.w_rel = lookback_weights(p.rel_f.alpha, p.geo.chunk, 1 << 16);
    p.w_att = lookback_weights(p.att.alpha, p.geo.chunk, 1 << 16);
    if (p.general) {
        if (p.threads != 256)
            return "hold/release filter orders above 1 together with attack/hold times that need 1024-block chunks are not implemented";
        const int k = p.general;
        butter_tf(p.hold_order, c.hold_filter_coefficient, sr, p.hold_b, p.hold_a);
        butter_tf(p.rel_order, c.release_filter_coefficient / c.release_ms, sr, p.rel_b, p.rel_a);
        for (int i = p.hold_order + 1; i <= LIMITER_MAX_ORDER; ++i) p.hold_b[i] = p.hold_a[i] = 0.0;
        for (int i = p.rel_order + 1; i <= LIMITER_MAX_ORDER; ++i) p.rel_b[i] = p.rel_a[i] = 0.0;
        // block maps and chunk words live in the shifted basis (limiter_general.h): the powers are those of I + dm
        p.hold_sb = shifted_basis(p.hold_order, k, p.hold_b, p.hold_a, c.hold_filter_coefficient, sr);
        p.rel_sb = shifted_basis(p.rel_order, k, p.rel_b, p.rel_a, c.release_filter_coefficient / c.release_ms, sr);
        const std::vector<Wide>&mh = p.hold_sb.step, &mr = p.rel_sb.step;
        p.pow_hold = block_powers(k, mh);
        p.pow_rel = block_powers(k, mr);
        p.wk_hold = lookback_matrices(k, mh, p.geo.chunk, 1 << 14);
        p.wk_rel = lookback_matrices(k, mr, p.geo.chunk, 1 << 14);
        if (p.wk_hold.empty() || p.wk_rel.empty()) return "hold/release filter does not decay (unstable at this order and cut-off)";
    }
    return "";
}

// fills everything of Limiter2Args that derives from the parameters (pointers are the caller's)
inline void limiter_fill(const LimiterParams& p, float threshold, LimiterArgs& a) {
    a.threshold = threshold;
    a.hw = p.hw;
    a.hb = p.hb;
    a.gl = p.geo.gl;
    a.gr = p.geo.gr;
    a.gw = p.geo.gw;
    a.att = p.att;
    a.hold = p.hold_f;
    a.rel = p.rel_f;
    a.attf = to_f32(p.att);
    a.holdf = to_f32(p.hold_f);
    a.relf = to_f32(p.rel_f);
    a.pa16 = power16(p.att.alpha);
    a.ph16 = power16(p.hold_f.alpha);
    a.pr16 = power16(p.rel_f.alpha);
    a.n_hold = (int)p.w_hold.size();
    a.n_rel = (int)p.w_rel.size();
    a.n_att = (int)p.w_att.size();
    // quiet chunks: sum_{k<C} ar^(C-1-k) br ah^k = br (ar^C - ah^C) / (ar - ah), in extended precision
    const Wide ar = p.rel_f.alpha, ah = p.hold_f.alpha, gap = ar - ah;
    // (phase_quiet_store forms beta_r hc (ar^k - ah^k) / (ar - ah) in float32 from two v_exp_f32 results, 2e-7 relative
    // each: the difference is good to ~4e-7 beta_r / |ar - ah| of hc.  Poles closer than 4 beta_r -- an error of 1e-7
    // of the hold carry, a tenth of the limiter tests' bound -- take the ordinary path, which runs the recurrence
    // instead; the default filters are 25 beta_r apart (1.6e-8).  ADVICE round 3.)
    a.quiet_ok = std::fabs((double)gap) > std::max(1e-9, 4.0 * std::fabs(p.rel_f.beta)) ? 1 : 0;
    a.quiet_rel_gain = a.quiet_ok ? (double)((Wide)p.rel_f.beta * (std::pow(ar, (Wide)p.geo.chunk) - std::pow(ah, (Wide)p.geo.chunk)) / gap) : 0.0;
    a.log2_hold = (float)std::log2(p.hold_f.alpha);
    a.log2_rel = (float)std::log2(p.rel_f.alpha);
    a.log2_att = (float)std::log2(p.att.alpha);
}

// look-back words of a launch: [3][nchunks] of the first-order kernel (the general one uses the attack
// row only) followed by [2][K][nchunks] state words of the general kernel
inline long long limiter_words(const LimiterParams& p, long long nchunks) { return (3 + 2 * (long long)p.general) * nchunks; }

// the general kernel's argument block; tables = device (or emulation) copies of pow_hold | pow_rel | wk_hold | wk_rel
template <int K>
inline GeneralArgs<K> general_fill(const LimiterParams& p, const double* tables, unsigned long long* published, long long nchunks) {
    GeneralArgs<K> g;
    g.hold = section_of<K>(p.hold_order, p.hold_b, p.hold_a, p.hold_sb);
    g.rel = section_of<K>(p.rel_order, p.rel_b, p.rel_a, p.rel_sb);
    g.pow_hold = tables;
    g.pow_rel = g.pow_hold + p.pow_hold.size();
    g.w_hold = g.pow_rel + p.pow_rel.size();
    g.w_rel = g.w_hold + p.wk_hold.size();
    g.n_hold = (int)(p.wk_hold.size() / (K * K));
    g.n_rel = (int)(p.wk_rel.size() / (K * K));
    g.words = published + 3 * nchunks;
    return g;
}
inline std::vector<double> general_tables(const LimiterParams& p) {
    std::vector<double> t(p.pow_hold);
    t.insert(t.end(), p.pow_rel.begin(), p.pow_rel.end());
    t.insert(t.end(), p.wk_hold.begin(), p.wk_hold.end());
    t.insert(t.end(), p.wk_rel.begin(), p.wk_rel.end());
    return t;
}

inline int ilog2_exact(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return (1 << l) == v ? l : -1;
}

}  // namespace mgx
