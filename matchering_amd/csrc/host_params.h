// Host-side derivation of kernel parameters from mgx_config (plain C++; shared
// by the HIP library and the CPU emulation harness).
#pragma once

#include <cmath>
#include <string>

#include "../../include/mgx.h"
#include <vector>

#include "limiter_kernel.h"

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

struct LimiterParams {
    int attack, hold, hw, hb, ha;
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

// utils.py:50-55, hyrax.py:43-75.  Returns an error text, empty when fine.
inline std::string limiter_params(const mgx_config& c, LimiterParams& p) {
    const double sr = c.internal_sample_rate;
    p.attack = (int)(sr * c.attack_ms * 1e-3);
    p.hold = (int)(sr * c.hold_ms * 1e-3);
    if (p.attack < 1) return "limiter attack shorter than one sample";
    if (p.hold < 3) return "limiter hold shorter than three samples (the reference's sliding window is empty there)";
    if (c.hold_filter_order != 1 || c.release_filter_order != 1)
        return "hold/release filter orders other than 1 are not implemented";
    const int w = (p.attack & 1) ? p.attack : p.attack + 1;
    p.hw = w - 1;
    p.hb = p.hold - 1;
    const double rho = std::exp(c.attack_filter_coefficient / p.attack);
    if (!(rho > 0.0 && rho < 1.0)) return "attack_filter_coefficient must be negative";
    p.att = Iir1{1.0 - rho, rho, rho * (1.0 - rho)};
    p.hold_f = butter1(c.hold_filter_coefficient, sr);
    p.rel_f = butter1(c.release_filter_coefficient / c.release_ms, sr);
    // frames after which the attack smoother has forgotten its state (rho^ha <= 1e-8)
    p.ha = (int)std::ceil(std::log(1e-8) / std::log(rho));
    // 256 blocks per chunk while the halos leave at least a quarter of them to the core, else 1024
    p.threads = 256;
    p.geo = LimiterBlock<256>::geometry(p.hw, p.hb, p.ha);
    if (p.geo.core_blocks < 64) {
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
}

inline int ilog2_exact(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return (1 << l) == v ? l : -1;
}

}  // namespace mgx
