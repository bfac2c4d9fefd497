// Host-side derivation of kernel parameters from mgx_config (plain C++; shared
// by the HIP library and the CPU emulation harness).
#pragma once

#include <cmath>
#include <string>

#include "../../include/mgx.h"
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
    int attack, hold, hw, hb;
    Iir1 att, hold_f, rel_f;
};

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
    const LimiterBlock::Geometry g = LimiterBlock::geometry(p.hw, p.hb);
    if (g.core_blocks < 64) return "limiter attack/hold windows too long for the chunked kernel";
    const double rho = std::exp(c.attack_filter_coefficient / p.attack);
    if (!(rho > 0.0 && rho < 1.0)) return "attack_filter_coefficient must be negative";
    p.att = Iir1{1.0 - rho, rho, rho * (1.0 - rho)};
    p.hold_f = butter1(c.hold_filter_coefficient, sr);
    p.rel_f = butter1(c.release_filter_coefficient / c.release_ms, sr);
    return "";
}

inline int ilog2_exact(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return (1 << l) == v ? l : -1;
}

}  // namespace mgx
