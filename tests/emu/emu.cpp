// CPU emulation harness for the kernel phase functions -- TEST INFRASTRUCTURE.
//
// Compiled by tests/emu/build.py with a plain host compiler and -DMGX_HOST_EMU.
// It drives the SAME per-thread phase functions the HIP kernels inline
// (matchering_amd/csrc/*_kernel.h) with a loop over thread ids where the GPU has
// a workgroup and a plain sequence point where the GPU has a barrier.  The CPU
// test-suite uses it to check the index arithmetic of the kernels against the
// oracle without a GPU.  It is not part of the product and is never loaded by
// matchering_amd.
#include <cmath>
#include <cstring>
#include <functional>
#include <memory>
#include <vector>

#include "../../matchering_amd/csrc/analysis2_kernel.h"
#include "../../matchering_amd/csrc/conv2_kernel.h"
#include "../../matchering_amd/csrc/conv_delay_kernel.h"
#include "../../matchering_amd/csrc/conv_wide_kernel.h"
#include "../../matchering_amd/csrc/fir_design.h"
#include "../../matchering_amd/csrc/host_params.h"
#include "../../matchering_amd/csrc/fir_plan.h"

using namespace mgx;

static std::vector<float2> twiddles(int n) {
    std::vector<float2> tw(n);
    const double pi = 3.14159265358979323846;
    for (int k = 0; k < n; ++k)
        tw[k] = make_float2((float)std::cos(2.0 * pi * k / n), (float)-std::sin(2.0 * pi * k / n));
    return tw;
}

#define FOR_THREADS(T_) for (int tid = 0; tid < (T_); ++tid)

// What runs between pass 0 and the inverse of pass 0 touches, in every wave, only points that wave owns (fft2.h,
// WAVE_LOCAL), and the kernels rely on it: no workgroup barrier separates those phases.  The emulation therefore
// runs such a sequence of phases WAVE BY WAVE -- wave 0 through all of them, then wave 1, ... -- so that a phase
// which reads another wave's points sees stale data here as it could on the GPU.  Plans without the property
// (and their kernels) keep a barrier per phase: phase by phase over all threads.
using Phase = std::function<void(int)>;
template <class F>
static void local_phases(const std::vector<Phase>& phases) {
    if (F::WAVE_LOCAL) {
        for (int w = 0; w < F::T / 64; ++w)
            for (const Phase& ph : phases)
                for (int tid = 64 * w; tid < 64 * w + 64; ++tid) ph(tid);
    } else {
        for (const Phase& ph : phases) { FOR_THREADS(F::T) ph(tid); }
    }
}
// the middle passes of a transform as phases
template <class F>
static std::vector<Phase> mid_phases(bool inverse, float2* lds, const float2* table) {
    std::vector<Phase> out;
    if (F::P < 3) return out;
    if (inverse) {
        if (F::P == 4) out.push_back([=](int tid) { F::inv_mid2(tid, lds, table); });
        out.push_back([=](int tid) { F::inv_mid(tid, lds, table); });
    } else {
        out.push_back([=](int tid) { F::fwd_mid(tid, lds, table); });
        if (F::P == 4) out.push_back([=](int tid) { F::fwd_mid2(tid, lds, table); });
    }
    return out;
}
static std::vector<Phase> operator+(std::vector<Phase> a, const std::vector<Phase>& b) {
    a.insert(a.end(), b.begin(), b.end());
    return a;
}
template <class F>
static void mid_pass(bool inverse, float2* lds, const float2* table) { local_phases<F>(mid_phases<F>(inverse, lds, table)); }

// ---------------------------------------------------------------------------
template <int LOG2N>
static int conv_impl(const float* x, long long n, const double* fir_mid, const double* fir_side, int taps,
                     double gain, float* y, float* ymid, double* peak) {
    using CB = Conv2Block<LOG2N>;
    using F = typename CB::F;
    const int parts = 2 * taps / F::N;
    const std::vector<float2> tw = twiddles(F::N);
    std::vector<float2> lds(F::LDS_ELEMS), mid_table(F::MID_TABLE + 1), tables((size_t)2 * parts * F::N);
    std::vector<float> h(2 * taps);
    for (int i = 0; i < taps; ++i) { h[i] = (float)fir_mid[i]; h[taps + i] = (float)fir_side[i]; }
    std::vector<typename CB::Persist> ps(F::T);
    FOR_THREADS(F::T) CB::load_persist(tid, tw.data(), mid_table.data(), ps[tid]);
    for (int ch = 0; ch < 2; ++ch)
        for (int k = 0; k < parts; ++k) {
            FOR_THREADS(F::T) CB::phase_load_taps(tid, h.data() + ((size_t)ch * parts + k) * CB::TAPS, ps[tid], lds.data());
            float2* table = tables.data() + ((size_t)ch * parts + k) * F::N;
            local_phases<F>(mid_phases<F>(false, lds.data(), mid_table.data()) + std::vector<Phase>{[&](int tid) {
                                CB::phase_write_filter(tid, lds.data(), (float)(gain / F::N), table);
                            }});
        }
    Conv2Args a;
    a.x = reinterpret_cast<const float2*>(x);
    a.n = n;
    a.y = reinterpret_cast<float2*>(y);
    a.ymid = ymid;
    a.h_mid = tables.data();
    a.h_side = tables.data() + (size_t)parts * F::N;
    a.tw = tw.data();
    a.parts = parts;
    a.npairs = (n + 2 * CB::LOUT - 1) / (2 * CB::LOUT);
    a.pair_peak = nullptr;
    float pk = 0.f;
    std::vector<typename CB::Kept> kept(F::T);
    std::vector<typename CB::RowAcc> acc(F::T);
    std::vector<typename CB::Held> held(F::T);
    // one channel of one pair, exactly the phase sequence of conv_channel() in mgx_kernels.h
    auto channel = [&](long long pair, bool edge, bool side) {
        const float2* hh = side ? a.h_side : a.h_mid;
        if (parts == 1) {
            // as conv_pair() in mgx_kernels.h: the mid pass reads the frames and leaves the side samples
            if (side) { FOR_THREADS(F::T) CB::phase_pass0_side(tid, held[tid], ps[tid], lds.data()); }
            else {
                FOR_THREADS(F::T) {
                    typename CB::Raw raw;
                    CB::fetch_frames(tid, pair, a, 0, raw);
                    CB::phase_pass0_mid(tid, raw, ps[tid], lds.data(), held[tid]);
                }
            }
            // (k_conv: no barrier between the middle passes and the row, in either direction)
            local_phases<F>(mid_phases<F>(false, lds.data(), mid_table.data()) +
                            std::vector<Phase>{[&](int tid) {
                                typename CB::RowFilter rf;
                                CB::fetch_filter(tid, hh, rf);
                                CB::phase_filter(tid, rf, lds.data());
                            }} +
                            mid_phases<F>(true, lds.data(), mid_table.data()));
            return;
        } else {
            FOR_THREADS(F::T) CB::clear_acc(acc[tid]);
            for (int k = 0; k < parts; ++k) {
                if (side) { FOR_THREADS(F::T) CB::template phase_load<true>(tid, pair, edge, a, ps[tid], lds.data(), k); }
                else { FOR_THREADS(F::T) CB::template phase_load<false>(tid, pair, edge, a, ps[tid], lds.data(), k); }
                local_phases<F>(mid_phases<F>(false, lds.data(), mid_table.data()) + std::vector<Phase>{[&](int tid) {
                                    typename CB::RowFilter rf;
                                    CB::fetch_filter(tid, hh + (size_t)k * F::N, rf);
                                    CB::phase_accumulate(tid, rf, lds.data(), acc[tid]);
                                }});
            }
            local_phases<F>(std::vector<Phase>{[&](int tid) { CB::phase_finish_row(tid, acc[tid], lds.data()); }} +
                            mid_phases<F>(true, lds.data(), mid_table.data()));
        }
    };
    for (long long pair = 0; pair < a.npairs; ++pair) {
        const bool edge = !CB::interior(pair, n, parts);
        channel(pair, edge, false);
        FOR_THREADS(F::T) CB::phase_keep_mid(tid, ps[tid], lds.data(), kept[tid]);
        channel(pair, edge, true);
        FOR_THREADS(F::T) pk = std::fmax(pk, CB::phase_store(tid, pair, edge, a, ps[tid], lds.data(), kept[tid]));
    }
    if (peak) *peak = pk;
    return 0;
}

// block_log2 = 0: N = 2*taps (one partition); otherwise N = 2^block_log2 and 2*taps/N partitions
extern "C" int emu_convolve_blocked(const float* x, long long n, const double* fir_mid, const double* fir_side,
                                    int taps, double gain, float* y, float* ymid, double* peak, int block_log2) {
    const int l = ilog2_exact(taps);
    if (l < 0) return -1;
    const int log2b = block_log2 ? block_log2 : l + 1;
    if ((2 * taps) % (1 << log2b) != 0) return -1;
    switch (log2b) {
#define CASE(L) case L: return conv_impl<L>(x, n, fir_mid, fir_side, taps, gain, y, ymid, peak);
        CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13) CASE(14)
#undef CASE
        default: return -4;
    }
}
extern "C" int emu_convolve(const float* x, long long n, const double* fir_mid, const double* fir_side,
                            int taps, double gain, float* y, float* ymid, double* peak) {
    return emu_convolve_blocked(x, n, fir_mid, fir_side, taps, gain, y, ymid, peak, 0);
}

// ---------------------------------------------------------------------------
// taps = N in two partitions as a frequency-domain delay line (conv_delay_kernel.h): the phase sequence of
// k_conv_delay in mgx_kernels.h, workgroups of `run` consecutive blocks each
template <int LOG2N>
static int conv_delay_impl(const float* x, long long n, const double* fir_mid, const double* fir_side, int taps,
                           double gain, float* y, float* ymid, float* block_peak, int run) {
    using CD = ConvDelay<LOG2N>;
    using CB = Conv2Block<LOG2N>;
    using F = typename CB::F;
    if (taps != F::N || run < 1) return -1;
    const int parts = 2;
    const std::vector<float2> tw = twiddles(F::N);
    std::vector<float2> lds(F::LDS_ELEMS), mid_table(F::MID_TABLE + 1), tables((size_t)2 * parts * F::N);
    std::vector<float> h(2 * taps);
    for (int i = 0; i < taps; ++i) { h[i] = (float)fir_mid[i]; h[taps + i] = (float)fir_side[i]; }
    std::vector<typename CB::Persist> ps(F::T);
    FOR_THREADS(F::T) CB::load_persist(tid, tw.data(), mid_table.data(), ps[tid]);
    for (int ch = 0; ch < 2; ++ch)
        for (int k = 0; k < parts; ++k) {
            FOR_THREADS(F::T) CB::phase_load_taps(tid, h.data() + ((size_t)ch * parts + k) * CB::TAPS, ps[tid], lds.data());
            float2* table = tables.data() + ((size_t)ch * parts + k) * F::N;
            local_phases<F>(mid_phases<F>(false, lds.data(), mid_table.data()) + std::vector<Phase>{[&](int tid) {
                                CB::phase_write_filter(tid, lds.data(), (float)(gain / F::N), table);
                            }});
        }
    Conv2Args a;
    a.x = reinterpret_cast<const float2*>(x);
    a.n = n;
    a.y = reinterpret_cast<float2*>(y);
    a.ymid = ymid;
    a.h_mid = tables.data();
    a.h_side = tables.data() + (size_t)parts * F::N;
    a.tw = tw.data();
    a.parts = parts;
    a.npairs = (n + CD::HOP - 1) / CD::HOP;              // blocks
    a.pair_peak = nullptr;
    std::vector<typename CD::Carry> carry(F::T);
    std::vector<typename CD::HeldFrames> held(F::T);
    for (long long first = 0; first < a.npairs; first += run) {          // one workgroup
        const long long end = std::min<long long>(a.npairs, first + run);
        FOR_THREADS(F::T) CD::clear(carry[tid]);
        for (long long b = first - 1; b < end; ++b) {
            if (b == first - 1) { FOR_THREADS(F::T) CD::prime(tid, b, a, held[tid]); }
            FOR_THREADS(F::T) {
                typename CD::HalfFrames newer;
                CD::template fetch_half<CD::R0 / 2>(tid, b, a, newer);
                CD::phase_pass0_held(tid, ps[tid], held[tid], newer, lds.data());
            }
            local_phases<F>(mid_phases<F>(false, lds.data(), mid_table.data()) +
                            std::vector<Phase>{[&](int tid) { CD::phase_row(tid, lds.data()); }});
            FOR_THREADS(F::T) CD::phase_multiply(tid, a, carry[tid], lds.data());   // (mirror rows: a barrier either side)
            if (b < first) continue;                                       // the block in front of the run: carry only
            local_phases<F>(std::vector<Phase>{[&](int tid) { CD::phase_row_back(tid, lds.data()); }} +
                            mid_phases<F>(true, lds.data(), mid_table.data()));
            float pk = 0.f;
            FOR_THREADS(F::T) pk = std::fmax(pk, CD::phase_store(tid, b, a, ps[tid], lds.data()));
            if (block_peak) block_peak[b] = pk;
        }
    }
    return 0;
}
extern "C" int emu_convolve_delay(const float* x, long long n, const double* fir_mid, const double* fir_side, int taps,
                                  double gain, float* y, float* ymid, float* block_peak, int run) {
    switch (ilog2_exact(taps)) {
#define CASE(L) case L: return conv_delay_impl<L>(x, n, fir_mid, fir_side, taps, gain, y, ymid, block_peak, run);
        CASE(11) CASE(12) CASE(13) CASE(14)
#undef CASE
        default: return -4;
    }
}

// ---------------------------------------------------------------------------
// F taps on N = 4F blocks (conv_wide_kernel.h): the phase sequence of k_conv_wide in mgx_kernels.h
template <int LOG2N>
static int conv_wide_impl(const float* x, long long n, const double* fir_mid, const double* fir_side, int taps,
                          double gain, float* y, float* ymid, float* block_peak) {
    using CW = ConvWide<LOG2N>;
    using CB = Conv2Block<LOG2N>;
    using F = typename CB::F;
    if (taps != CW::TAPS) return -1;
    const std::vector<float2> tw = twiddles(F::N);
    std::vector<float2> lds(F::LDS_ELEMS), mid_table(F::MID_TABLE + 1), tables((size_t)2 * F::N);
    std::vector<float> h(2 * taps);
    for (int i = 0; i < taps; ++i) { h[i] = (float)fir_mid[i]; h[taps + i] = (float)fir_side[i]; }
    std::vector<typename CB::Persist> ps(F::T);
    FOR_THREADS(F::T) CB::load_persist(tid, tw.data(), mid_table.data(), ps[tid]);
    for (int ch = 0; ch < 2; ++ch) {
        FOR_THREADS(F::T) CW::phase_load_taps(tid, h.data() + (size_t)ch * taps, ps[tid], lds.data());
        float2* table = tables.data() + (size_t)ch * F::N;
        local_phases<F>(mid_phases<F>(false, lds.data(), mid_table.data()) + std::vector<Phase>{[&](int tid) {
                            CB::phase_write_filter(tid, lds.data(), (float)(gain / F::N), table);
                        }});
    }
    Conv2Args a;
    a.x = reinterpret_cast<const float2*>(x);
    a.n = n;
    a.y = reinterpret_cast<float2*>(y);
    a.ymid = ymid;
    a.h_mid = tables.data();
    a.h_side = tables.data() + F::N;
    a.tw = tw.data();
    a.parts = 1;
    a.npairs = (n + CW::HOP - 1) / CW::HOP;              // blocks
    a.pair_peak = nullptr;
    for (long long b = 0; b < a.npairs; ++b) {
        FOR_THREADS(F::T) {
            typename CW::Frames fr;
            CW::fetch(tid, b, a, fr);
            CW::phase_pass0(tid, ps[tid], fr, lds.data());
        }
        std::vector<typename CW::Filters> filt(F::T);
        local_phases<F>(mid_phases<F>(false, lds.data(), mid_table.data()) + std::vector<Phase>{[&](int tid) {
                            CW::fetch_filters(tid, a, filt[tid]);
                            CW::phase_row(tid, lds.data());
                        }});
        FOR_THREADS(F::T) CW::phase_multiply(tid, filt[tid], lds.data());   // (mirror rows: a barrier either side)
        local_phases<F>(std::vector<Phase>{[&](int tid) { CW::phase_row_back(tid, lds.data()); }} +
                        mid_phases<F>(true, lds.data(), mid_table.data()));
        float pk = 0.f;
        FOR_THREADS(F::T) pk = std::fmax(pk, CW::phase_store(tid, b, a, ps[tid], lds.data()));
        if (block_peak) block_peak[b] = pk;
    }
    return 0;
}
extern "C" int emu_convolve_wide(const float* x, long long n, const double* fir_mid, const double* fir_side, int taps,
                                 double gain, float* y, float* ymid, float* block_peak) {
    switch (ilog2_exact(taps)) {
#define CASE(L) case L - 2: return conv_wide_impl<L>(x, n, fir_mid, fir_side, taps, gain, y, ymid, block_peak);
        CASE(11) CASE(12) CASE(13) CASE(14)
#undef CASE
        default: return -4;
    }
}

// ---------------------------------------------------------------------------
template <int LOG2N>
static int analyze_impl(const float* x, long long n, const mgx_config* cfg, int is_reference, double* peak,
                        double* amplitude_c, double* match_rms, int* divisions_out, long long* piece_out,
                        double* piece_rms, int* loud, double* avg_mid, double* avg_side) {
    using AB = Analysis2Block<LOG2N>;
    using F = typename AB::F;
    const std::vector<float2> tw = twiddles(F::N);
    int divisions;
    long long piece;
    piece_geometry(n, cfg->max_piece_size, divisions, piece);
    AnalysisArgs a;
    a.x = reinterpret_cast<const float2*>(x);
    a.n = n;
    a.fft = F::N;
    a.piece = piece;
    a.divisions = divisions;
    a.segs_per_piece = (int)(piece / F::N);
    a.chunks_per_piece = std::max(1, (a.segs_per_piece + 4) / 5);
    const int nwg = divisions * a.chunks_per_piece, half = F::N / 2;
    std::vector<double> wg_sumsq(nwg);
    std::vector<float> wg_peak(nwg), wg_spec((size_t)nwg * 2 * (half + 1), 0.f);
    a.wg_sumsq = wg_sumsq.data();
    a.wg_peak = wg_peak.data();
    a.wg_spec = wg_spec.data();
    a.tw = tw.data();
    std::vector<float2> lds(F::LDS_ELEMS);
    std::vector<typename AB::Thread> th(F::T);
    std::vector<typename AB::Row> own(F::T);
    std::vector<typename AB::Persist> ps(F::T);
    std::vector<float2> mid_table(F::MID_TABLE + 1);
    FOR_THREADS(F::T) AB::load_persist(tid, a.tw, mid_table.data(), ps[tid]);
    for (int wg = 0; wg < nwg; ++wg) {
        const int d = wg / a.chunks_per_piece, ch = wg % a.chunks_per_piece;
        FOR_THREADS(F::T) AB::init(th[tid]);
        int s0, s1;
        Analysis2Block<LOG2N>::chunk_segments(a, ch, s0, s1);
        for (int s = s0; s < s1; ++s) {
            const long long start = d * piece + (long long)s * F::N;
            FOR_THREADS(F::T) { typename AB::Raw raw; AB::fetch(tid, start, a, raw); AB::phase_load(tid, raw, ps[tid], th[tid], lds.data()); }
            local_phases<F>(mid_phases<F>(false, lds.data(), mid_table.data()) +
                            std::vector<Phase>{[&](int tid) { AB::phase_row(tid, own[tid], lds.data()); }});
            FOR_THREADS(F::T) AB::phase_magnitudes(tid, own[tid], th[tid], lds.data());
        }
        if (ch == a.chunks_per_piece - 1) {
            FOR_THREADS(F::T)
            AB::phase_loose_frames(tid, d * piece + (long long)a.segs_per_piece * F::N, (d + 1) * piece, true, a, th[tid]);
            if (d == divisions - 1) {
                FOR_THREADS(F::T) AB::phase_loose_frames(tid, (long long)divisions * piece, n, false, a, th[tid]);
            }
        }
        FOR_THREADS(F::T) AB::phase_write_spectrum(tid, wg, a, th[tid]);
        double ss = 0.0;
        float pk = 0.f;
        FOR_THREADS(F::T) { ss += th[tid].sumsq; pk = std::fmax(pk, th[tid].peak); }
        wg_sumsq[wg] = ss;
        wg_peak[wg] = pk;
    }
    std::vector<double> rms(divisions);
    std::vector<int> ld(divisions);
    TrackStats st;
    finish_levels(wg_sumsq.data(), wg_peak.data(), a.chunks_per_piece, divisions, piece, is_reference != 0,
                  cfg->threshold, cfg->min_value, rms.data(), ld.data(), st);
    if (peak) *peak = st.peak;
    if (amplitude_c) *amplitude_c = st.amplitude_c;
    if (match_rms) *match_rms = st.match_rms;
    if (divisions_out) *divisions_out = divisions;
    if (piece_out) *piece_out = piece;
    for (int d = 0; d < divisions; ++d) {
        if (piece_rms) piece_rms[d] = rms[d];
        if (loud) loud[d] = ld[d];
    }
    const double scale = 1.0 / ((double)st.loud_count * a.segs_per_piece * (double)F::N * st.amplitude_c);
    for (int k = 0; k <= half; ++k) {
        double sm = 0.0, ssd = 0.0;
        for (int wg = 0; wg < nwg; ++wg) {
            if (!ld[wg / a.chunks_per_piece]) continue;
            sm += wg_spec[(size_t)wg * 2 * (half + 1) + k];
            ssd += wg_spec[(size_t)wg * 2 * (half + 1) + (half + 1) + k];
        }
        if (avg_mid) avg_mid[k] = sm * scale;
        if (avg_side) avg_side[k] = ssd * scale;
    }
    return 0;
}

// fft_size = 2 * Fft2<LOG2H>::N: k_analyze_double of mgx_kernels.h, phase by phase
template <int LOG2H>
static int analyze_double_impl(const float* x, long long n, const mgx_config* cfg, int is_reference, double* peak,
                               double* amplitude_c, double* match_rms, int* divisions_out, long long* piece_out,
                               double* piece_rms, int* loud, double* avg_mid, double* avg_side) {
    using AD = AnalysisDouble<LOG2H>;
    using F = typename AD::F;
    const int fft = 2 * F::N, half = F::N;
    const std::vector<float2> tw = twiddles(F::N);
    int divisions;
    long long piece;
    piece_geometry(n, cfg->max_piece_size, divisions, piece);
    AnalysisArgs a;
    a.x = reinterpret_cast<const float2*>(x);
    a.n = n;
    a.fft = fft;
    a.piece = piece;
    a.divisions = divisions;
    a.segs_per_piece = (int)(piece / fft);
    a.chunks_per_piece = std::max(1, (a.segs_per_piece + 4) / 5);
    const int nwg = divisions * a.chunks_per_piece;
    std::vector<double> wg_sumsq(nwg);
    std::vector<float> wg_peak(nwg), wg_spec((size_t)nwg * 2 * (half + 1), 0.f);
    a.wg_sumsq = wg_sumsq.data();
    a.wg_peak = wg_peak.data();
    a.wg_spec = wg_spec.data();
    a.tw = tw.data();
    std::vector<float2> lds(F::LDS_ELEMS);
    std::vector<typename AD::Thread> th(F::T);
    std::vector<typename AD::Persist> ps(F::T);
    std::vector<float2> mid_table(F::MID_TABLE + 1);
    FOR_THREADS(F::T) AD::AB::load_persist(tid, a.tw, mid_table.data(), ps[tid]);
    for (int wg = 0; wg < nwg; ++wg) {
        const int d = wg / a.chunks_per_piece, ch = wg % a.chunks_per_piece;
        FOR_THREADS(F::T) AD::init(th[tid]);
        int s0, s1;
        AD::AB::chunk_segments(a, ch, s0, s1);
        for (int s = s0; s < s1; ++s) {
            const long long start = d * piece + (long long)s * fft;
            FOR_THREADS(F::T) AD::template phase_load<false>(tid, start, a, ps[tid], th[tid], lds.data());
            local_phases<F>(mid_phases<F>(false, lds.data(), mid_table.data()) +
                            std::vector<Phase>{[&](int tid) { AD::phase_row(tid, lds.data()); }});
            FOR_THREADS(F::T) AD::template phase_magnitudes<false>(tid, th[tid], lds.data());
            FOR_THREADS(F::T) AD::template phase_load<true>(tid, start, a, ps[tid], th[tid], lds.data());
            local_phases<F>(mid_phases<F>(false, lds.data(), mid_table.data()) +
                            std::vector<Phase>{[&](int tid) { AD::phase_row(tid, lds.data()); }});
            FOR_THREADS(F::T) AD::template phase_magnitudes<true>(tid, th[tid], lds.data());
        }
        if (ch == a.chunks_per_piece - 1) {
            FOR_THREADS(F::T)
            AD::phase_loose_frames(tid, d * piece + (long long)a.segs_per_piece * fft, (d + 1) * piece, true, a, th[tid]);
            if (d == divisions - 1) {
                FOR_THREADS(F::T) AD::phase_loose_frames(tid, (long long)divisions * piece, n, false, a, th[tid]);
            }
        }
        FOR_THREADS(F::T) AD::phase_write_spectrum(tid, wg, a, th[tid]);
        double ss = 0.0;
        float pk = 0.f;
        FOR_THREADS(F::T) { ss += th[tid].sumsq; pk = std::fmax(pk, th[tid].peak); }
        wg_sumsq[wg] = ss;
        wg_peak[wg] = pk;
    }
    std::vector<double> rms(divisions);
    std::vector<int> ld(divisions);
    TrackStats st;
    finish_levels(wg_sumsq.data(), wg_peak.data(), a.chunks_per_piece, divisions, piece, is_reference != 0,
                  cfg->threshold, cfg->min_value, rms.data(), ld.data(), st);
    if (peak) *peak = st.peak;
    if (amplitude_c) *amplitude_c = st.amplitude_c;
    if (match_rms) *match_rms = st.match_rms;
    if (divisions_out) *divisions_out = divisions;
    if (piece_out) *piece_out = piece;
    for (int d = 0; d < divisions; ++d) {
        if (piece_rms) piece_rms[d] = rms[d];
        if (loud) loud[d] = ld[d];
    }
    const double scale = 1.0 / ((double)st.loud_count * a.segs_per_piece * (double)fft * st.amplitude_c);
    for (int k = 0; k <= half; ++k) {
        double sm = 0.0, ssd = 0.0;
        for (int wg = 0; wg < nwg; ++wg) {
            if (!ld[wg / a.chunks_per_piece]) continue;
            sm += wg_spec[(size_t)wg * 2 * (half + 1) + k];
            ssd += wg_spec[(size_t)wg * 2 * (half + 1) + (half + 1) + k];
        }
        if (avg_mid) avg_mid[k] = sm * scale;
        if (avg_side) avg_side[k] = ssd * scale;
    }
    return 0;
}

// fft_size = 4 * Fft2<LOG2H>::N: k_analyze_quad of mgx_kernels.h, phase by phase
template <int LOG2H>
static int analyze_quad_impl(const float* x, long long n, const mgx_config* cfg, int is_reference, double* peak,
                             double* amplitude_c, double* match_rms, int* divisions_out, long long* piece_out,
                             double* piece_rms, int* loud, double* avg_mid, double* avg_side) {
    using AQ = AnalysisQuad<LOG2H>;
    using F = typename AQ::F;
    const int fft = 4 * F::N, half = 2 * F::N;
    const std::vector<float2> tw = twiddles(F::N);
    int divisions;
    long long piece;
    piece_geometry(n, cfg->max_piece_size, divisions, piece);
    AnalysisArgs a;
    a.x = reinterpret_cast<const float2*>(x);
    a.n = n;
    a.fft = fft;
    a.piece = piece;
    a.divisions = divisions;
    a.segs_per_piece = (int)(piece / fft);
    if (a.segs_per_piece < 1) return -5;
    a.chunks_per_piece = std::max(1, (a.segs_per_piece + 4) / 5);
    const int nwg = divisions * a.chunks_per_piece;
    std::vector<double> wg_sumsq(nwg);
    std::vector<float> wg_peak(nwg), wg_spec((size_t)nwg * 2 * (half + 1), -1.f);      // (the kernel clears its rows)
    std::vector<float2> wg_pack((size_t)nwg * AQ::SCRATCH_FLOAT2);
    a.wg_sumsq = wg_sumsq.data();
    a.wg_peak = wg_peak.data();
    a.wg_spec = wg_spec.data();
    a.wg_pack = wg_pack.data();
    a.tw = tw.data();
    std::vector<float2> lds(F::LDS_ELEMS);
    std::vector<typename AQ::Thread> th(F::T);
    std::vector<typename AQ::Pairs> pairs(F::T);
    std::vector<typename AQ::Persist> ps(F::T);
    std::vector<float2> mid_table(F::MID_TABLE + 1);
    FOR_THREADS(F::T) AQ::AB::load_persist(tid, a.tw, mid_table.data(), ps[tid]);
    auto transform = [&]() {
        local_phases<F>(mid_phases<F>(false, lds.data(), mid_table.data()) +
                        std::vector<Phase>{[&](int tid) { AQ::phase_row(tid, lds.data()); }});
        FOR_THREADS(F::T) AQ::phase_unmix(tid, pairs[tid], lds.data());
    };
    for (int wg = 0; wg < nwg; ++wg) {
        const int d = wg / a.chunks_per_piece, ch = wg % a.chunks_per_piece;
        FOR_THREADS(F::T) AQ::init(th[tid]);
        FOR_THREADS(F::T) AQ::phase_clear(tid, wg, a);
        int s0, s1;
        AQ::AB::chunk_segments(a, ch, s0, s1);
        for (int s = s0; s < s1; ++s) {
            const long long start = d * piece + (long long)s * fft;
            FOR_THREADS(F::T) AQ::template phase_load<false, 0>(tid, start, a, ps[tid], th[tid], lds.data());
            transform();
            FOR_THREADS(F::T) AQ::phase_keep(tid, wg, a, pairs[tid]);
            FOR_THREADS(F::T) AQ::template phase_load<false, 1>(tid, start, a, ps[tid], th[tid], lds.data());
            transform();
            FOR_THREADS(F::T) AQ::template phase_magnitudes<false>(tid, wg, a, pairs[tid]);
            FOR_THREADS(F::T) AQ::template phase_load<true, 0>(tid, start, a, ps[tid], th[tid], lds.data());
            transform();
            FOR_THREADS(F::T) AQ::phase_keep(tid, wg, a, pairs[tid]);
            FOR_THREADS(F::T) AQ::template phase_load<true, 1>(tid, start, a, ps[tid], th[tid], lds.data());
            transform();
            FOR_THREADS(F::T) AQ::template phase_magnitudes<true>(tid, wg, a, pairs[tid]);
        }
        if (ch == a.chunks_per_piece - 1) {
            FOR_THREADS(F::T)
            AQ::phase_loose_frames(tid, d * piece + (long long)a.segs_per_piece * fft, (d + 1) * piece, true, a, th[tid]);
            if (d == divisions - 1) {
                FOR_THREADS(F::T) AQ::phase_loose_frames(tid, (long long)divisions * piece, n, false, a, th[tid]);
            }
        }
        FOR_THREADS(F::T) AQ::phase_write_spectrum(tid, wg, a);
        double ss = 0.0;
        float pk = 0.f;
        FOR_THREADS(F::T) { ss += th[tid].sumsq; pk = std::fmax(pk, th[tid].peak); }
        wg_sumsq[wg] = ss;
        wg_peak[wg] = pk;
    }
    std::vector<double> rms(divisions);
    std::vector<int> ld(divisions);
    TrackStats st;
    finish_levels(wg_sumsq.data(), wg_peak.data(), a.chunks_per_piece, divisions, piece, is_reference != 0,
                  cfg->threshold, cfg->min_value, rms.data(), ld.data(), st);
    if (peak) *peak = st.peak;
    if (amplitude_c) *amplitude_c = st.amplitude_c;
    if (match_rms) *match_rms = st.match_rms;
    if (divisions_out) *divisions_out = divisions;
    if (piece_out) *piece_out = piece;
    for (int d = 0; d < divisions; ++d) {
        if (piece_rms) piece_rms[d] = rms[d];
        if (loud) loud[d] = ld[d];
    }
    const double scale = 1.0 / ((double)st.loud_count * a.segs_per_piece * (double)fft * st.amplitude_c);
    for (int k = 0; k <= half; ++k) {
        double sm = 0.0, ssd = 0.0;
        for (int wg = 0; wg < nwg; ++wg) {
            if (!ld[wg / a.chunks_per_piece]) continue;
            sm += wg_spec[(size_t)wg * 2 * (half + 1) + k];
            ssd += wg_spec[(size_t)wg * 2 * (half + 1) + (half + 1) + k];
        }
        if (avg_mid) avg_mid[k] = sm * scale;
        if (avg_side) avg_side[k] = ssd * scale;
    }
    return 0;
}
// the four-transform form on a transform of 2^log2h points (fft_size = 4 * 2^log2h): small sizes test it quickly
extern "C" int emu_analyze_quad(const float* x, long long n, const mgx_config* cfg, int is_reference, double* peak,
                                double* amplitude_c, double* match_rms, int* divisions, long long* piece,
                                double* piece_rms, int* loud, double* avg_mid, double* avg_side, int log2h) {
    switch (log2h) {
#define CASE(L) case L: return analyze_quad_impl<L>(x, n, cfg, is_reference, peak, amplitude_c, match_rms, divisions, piece, piece_rms, loud, avg_mid, avg_side);
        CASE(8) CASE(10) CASE(12) CASE(14)
#undef CASE
        default: return -4;
    }
}

extern "C" int emu_analyze(const float* x, long long n, const mgx_config* cfg, int is_reference, double* peak,
                           double* amplitude_c, double* match_rms, int* divisions, long long* piece,
                           double* piece_rms, int* loud, double* avg_mid, double* avg_side) {
    const int l = ilog2_exact(cfg->fft_size);
    switch (l) {
#define CASE(L) case L: return analyze_impl<L>(x, n, cfg, is_reference, peak, amplitude_c, match_rms, divisions, piece, piece_rms, loud, avg_mid, avg_side);
        CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13) CASE(14)
#undef CASE
        case 15: return analyze_double_impl<14>(x, n, cfg, is_reference, peak, amplitude_c, match_rms, divisions, piece, piece_rms, loud, avg_mid, avg_side);
        case 16: return analyze_quad_impl<14>(x, n, cfg, is_reference, peak, amplitude_c, match_rms, divisions, piece, piece_rms, loud, avg_mid, avg_side);
        // (the emulation also runs the double form on small transforms, to test it quickly: fft_size = -2^l)
        default: return -4;
    }
}

// ---------------------------------------------------------------------------
template <int T>
static int limit_impl(const LimiterParams& lp, const float* x, long long n, const mgx_config* cfg, double gain,
                      double post_gain, float* out, float* dbg_sl, float* dbg_sh) {
    using LB = LimiterBlock<T>;
    LimiterArgs a;
    limiter_fill(lp, (float)cfg->threshold, a);
    a.y = reinterpret_cast<const float2*>(x);
    a.n = n;
    a.out = reinterpret_cast<float2*>(out);
    a.gain = &gain;
    a.post_gain = &post_gain;
    a.active = nullptr;
    a.nchunks = (n + lp.geo.chunk - 1) / lp.geo.chunk;
    std::vector<unsigned long long> published(3 * a.nchunks, LIMITER_UNPUBLISHED);
    a.published = published.data();
    a.w_hold = lp.w_hold.data();
    a.w_rel = lp.w_rel.data();
    a.w_att = lp.w_att.data();
    int ctrl[2] = {0, 0};
    a.ticket = &ctrl[0];
    a.error = &ctrl[1];
    std::vector<float> lds(LB::LDS_BYTES / 4 + 8);
    std::vector<typename LB::Thread> th(LB::T);
    std::vector<Affine> in0(LB::T), in1(LB::T), pre0(LB::T), pre1(LB::T);
    // the workgroup scans of the device kernel (wave shuffles there), as plain ordered loops
    auto scan = [&](const std::vector<Affine>& in, std::vector<Affine>& pre, bool reverse) {
        Affine run = affine_identity();
        for (int i = 0; i < LB::T; ++i) {
            const int t = reverse ? LB::T - 1 - i : i;
            pre[t] = run;
            run = affine_then(run, in[t]);
        }
        return run;
    };
    auto carry = [&](long long chunk, int slot) {
        double s = 0.0;
        for (int lane = 0; lane < 64; ++lane) {
            typename LB::Polls p;
            LB::lookback_ask(lane, chunk, slot, a, p);
            s += LB::lookback_take(lane, chunk, slot, a, p);
        }
        return s;
    };
    for (long long chunk = 0; chunk < a.nchunks; ++chunk) {
        FOR_THREADS(LB::T) {
            float pm[LB::E / 2];
            LB::phase_load(tid, chunk, a, lds.data(), pm);
        }
        // block maxima: the device folds eight lanes with DPP steps
        for (int b = 0; b < LB::T; ++b) {
            float m = 0.f;
            for (int j = 0; j < LB::E; ++j) m = std::fmax(m, LB::plane(lds.data())[b * LB::STRIDE + j]);
            LB::block_max(lds.data())[b] = m;
        }
        FOR_THREADS(LB::T) in1[tid] = LB::phase_hold_window(tid, chunk, a, th[tid], lds.data());
        const Affine whole_hold = scan(in1, pre1, false);
        FOR_THREADS(LB::T) th[tid].hold_pre = pre1[tid];
        LB::lookback_publish(chunk, 0, a, whole_hold.b);
        FOR_THREADS(LB::T) in0[tid] = LB::phase_attack_window(tid, a, th[tid], lds.data());
        if (dbg_sl || dbg_sh) {
            FOR_THREADS(LB::T) {
                if (!th[tid].core) continue;
                for (int j = 0; j < th[tid].valid; ++j) {
                    if (dbg_sl) dbg_sl[th[tid].base + j] = th[tid].sl[j];
                    if (dbg_sh) dbg_sh[th[tid].base + j] = th[tid].sh[j];
                }
            }
        }
        scan(in0, pre0, false);
        LB::lookback_publish(chunk, 2, a, pre0[LB::T - a.gr].b);
        const bool tail = LB::tail_chunk(chunk, a);
        const double att_carry = carry(chunk, 2), hold_carry = carry(chunk, 0);
        FOR_THREADS(LB::T)
            in0[tid] = LB::phase_attack_forward(tid, a, th[tid], pre0[tid], tail ? att_carry : 0.0, lds.data());
        scan(in0, pre0, true);
        FOR_THREADS(LB::T) LB::phase_attack_backward(tid, a, th[tid], pre0[tid]);
        FOR_THREADS(LB::T) in1[tid] = LB::phase_hold(tid, a, th[tid], hold_carry, tail ? 0.0 : att_carry);
        const Affine whole_rel = scan(in1, pre1, false);
        LB::lookback_publish(chunk, 1, a, whole_rel.b);
        const double rel_carry = carry(chunk, 1);
        FOR_THREADS(LB::T) LB::phase_gain(tid, a, th[tid], pre1[tid], rel_carry, lds.data());
        if (LB::full_chunk(chunk, a)) {       // as limit_chunk<T, true>: the frames are reloaded ahead of the gains
            std::vector<typename LB::Reload> again(LB::T);
            FOR_THREADS(LB::T) LB::phase_reload(tid, chunk, a, again[tid]);
            FOR_THREADS(LB::T) LB::phase_store_reloaded(tid, chunk, a, again[tid], lds.data());
        } else {
            FOR_THREADS(LB::T) LB::phase_store(tid, chunk, a, true, lds.data());
        }
    }
    return ctrl[1] ? -2 : 0;
}

// hold / release filters of order up to K: limit_chunk_general of mgx_kernels.h, phase by phase
template <int K>
static int limit_general_impl(const LimiterParams& lp, const float* x, long long n, const mgx_config* cfg, double gain,
                              double post_gain, float* out) {
    using LB = LimiterBlock<256>;
    using LG = LimiterGeneral<K>;
    LimiterArgs a;
    limiter_fill(lp, (float)cfg->threshold, a);
    a.y = reinterpret_cast<const float2*>(x);
    a.n = n;
    a.out = reinterpret_cast<float2*>(out);
    a.gain = &gain;
    a.post_gain = &post_gain;
    a.active = nullptr;
    a.nchunks = (n + lp.geo.chunk - 1) / lp.geo.chunk;
    std::vector<unsigned long long> published(limiter_words(lp, a.nchunks), LIMITER_UNPUBLISHED);
    a.published = published.data();
    a.w_hold = lp.w_hold.data();
    a.w_rel = lp.w_rel.data();
    a.w_att = lp.w_att.data();
    int ctrl[2] = {0, 0};
    a.ticket = &ctrl[0];
    a.error = &ctrl[1];
    const std::vector<double> tables = general_tables(lp);
    const GeneralArgs<K> g = general_fill<K>(lp, tables.data(), a.published, a.nchunks);
    std::vector<float> lds(LG::LDS_BYTES / 4 + 8);
    std::vector<typename LB::Thread> th(LB::T);
    std::vector<Affine> in0(LB::T), pre0(LB::T);
    std::vector<StateMap<K>> pre(LB::T), rel_pre(LB::T), mr(LB::T);
    auto scan = [&](const std::vector<Affine>& in, std::vector<Affine>& out_pre, bool reverse) {
        Affine run = affine_identity();
        for (int i = 0; i < LB::T; ++i) {
            const int t = reverse ? LB::T - 1 - i : i;
            out_pre[t] = run;
            run = affine_then(run, in[t]);
        }
        return run;
    };
    auto attack_carry = [&](long long chunk) {
        double s = 0.0;
        for (int lane = 0; lane < 64; ++lane) {
            typename LB::Polls p;
            LB::lookback_ask(lane, chunk, 2, a, p);
            s += LB::lookback_take(lane, chunk, 2, a, p);
        }
        return s;
    };
    auto state_carry = [&](long long chunk, int filter, double (&c)[K]) {
        for (int k = 0; k < K; ++k) c[k] = 0.0;
        for (int lane = 0; lane < 64; ++lane) {
            double acc[K];
            LG::take(lane, chunk, filter, g, a, acc);
            for (int k = 0; k < K; ++k) c[k] += acc[k];
        }
    };
    auto lds_scan = [&]() {          // the barrier-separated phases of the device scan
        FOR_THREADS(LB::T) LG::scan_groups(lds.data(), tid);
        FOR_THREADS(LB::T) LG::scan_top(lds.data(), tid);
    };
    for (long long chunk = 0; chunk < a.nchunks; ++chunk) {
        FOR_THREADS(LB::T) {
            float pm[LB::E / 2];
            LB::phase_load(tid, chunk, a, lds.data(), pm);
        }
        for (int b = 0; b < LB::T; ++b) {
            float m = 0.f;
            for (int j = 0; j < LB::E; ++j) m = std::fmax(m, LB::plane(lds.data())[b * LB::STRIDE + j]);
            LB::block_max(lds.data())[b] = m;
        }
        FOR_THREADS(LB::T) {
            LB::phase_hold_window(tid, chunk, a, th[tid], lds.data());
            LG::scan_put(lds.data(), tid, th[tid].core && th[tid].valid > 0
                                              ? LG::block_map(g.hold, th[tid].sh, th[tid].valid, g.pow_hold)
                                              : LG::identity());
        }
        lds_scan();
        FOR_THREADS(LB::T) pre[tid] = LG::scan_prefix(lds.data(), tid);
        LG::publish(g, a.nchunks, 0, chunk, LG::scan_whole(lds.data()).v);
        FOR_THREADS(LB::T) in0[tid] = LB::phase_attack_window(tid, a, th[tid], lds.data());
        scan(in0, pre0, false);
        LB::lookback_publish(chunk, 2, a, pre0[LB::T - a.gr].b);
        const bool tail = LB::tail_chunk(chunk, a);
        const double att_carry = attack_carry(chunk);
        double hold_carry[K], rel_carry[K];
        state_carry(chunk, 0, hold_carry);
        FOR_THREADS(LB::T)
            in0[tid] = LB::phase_attack_forward(tid, a, th[tid], pre0[tid], tail ? att_carry : 0.0, lds.data());
        scan(in0, pre0, true);
        FOR_THREADS(LB::T) LB::phase_attack_backward(tid, a, th[tid], pre0[tid]);
        FOR_THREADS(LB::T) mr[tid] = LG::phase_hold(tid, a, g, th[tid], pre[tid], hold_carry, tail ? 0.0 : att_carry);
        FOR_THREADS(LB::T) LG::scan_put(lds.data(), tid, mr[tid]);
        lds_scan();
        FOR_THREADS(LB::T) rel_pre[tid] = LG::scan_prefix(lds.data(), tid);
        LG::publish(g, a.nchunks, 1, chunk, LG::scan_whole(lds.data()).v);
        state_carry(chunk, 1, rel_carry);
        FOR_THREADS(LB::T) LG::phase_gain(tid, g, th[tid], rel_pre[tid], rel_carry, lds.data());
        FOR_THREADS(LB::T) LB::phase_store(tid, chunk, a, true, lds.data());
    }
    return ctrl[1] ? -2 : 0;
}

extern "C" int emu_butter(int order, double fc, double fs, double* b, double* a) {
    butter_tf(order, fc, fs, b, a);
    return 0;
}

extern "C" int emu_limit(const float* x, long long n, const mgx_config* cfg, double gain, double post_gain,
                         float* out, float* dbg_sl, float* dbg_sh) {
    LimiterParams lp;
    if (!limiter_params(*cfg, lp).empty()) return -1;
    if (lp.general == 2) return limit_general_impl<2>(lp, x, n, cfg, gain, post_gain, out);
    if (lp.general == 3) return limit_general_impl<3>(lp, x, n, cfg, gain, post_gain, out);
    if (lp.threads == 1024) return limit_impl<1024>(lp, x, n, cfg, gain, post_gain, out, dbg_sl, dbg_sh);
    return limit_impl<256>(lp, x, n, cfg, gain, post_gain, out, dbg_sl, dbg_sh);
}

// host FIR design (product code, re-exported here so the CPU tests need no HIP runtime)
extern "C" int emu_design_fir(const mgx_config* cfg, const double* avg_target, const double* avg_reference,
                              double* taps, double* curve_raw, double* curve_smooth) {
    FirDesignParams p{cfg->fft_size, cfg->internal_sample_rate, cfg->lin_log_oversampling, cfg->lowess_frac,
                      cfg->lowess_it, cfg->lowess_delta, cfg->min_value};
    design_fir(avg_target, avg_reference, p, taps, curve_raw, curve_smooth);
    return 0;
}
extern "C" int emu_design_fir_direct(const mgx_config* cfg, const double* avg_target, const double* avg_reference,
                                     double* taps, double* curve_raw, double* curve_smooth) {
    FirDesignParams p{cfg->fft_size, cfg->internal_sample_rate, cfg->lin_log_oversampling, cfg->lowess_frac,
                      cfg->lowess_it, cfg->lowess_delta, cfg->min_value};
    design_fir_direct(avg_target, avg_reference, p, taps, curve_raw, curve_smooth);
    return 0;
}
// The raw -> smooth operator in its two factors through the LOWESS anchors (mgx.hip build_fir_factors, mgx_kernels.h
// k_fir_apply_a / k_fir_apply_b), with the phase functions the device kernels run: unit raw curves -> the anchors' fits
// (A), unit fits -> the smooth curve (B), rows cut to the window above 1e-18 of their largest entry (k_fir_band), then
// smooth = B (A raw) with bins 0 and 1 pinned.  Returns the windows' total sizes (in doubles) for the two factors.
extern "C" int emu_fir_factored(const mgx_config* cfg, const double* raw_in, double* smooth_out, long long* a_doubles,
                                long long* b_doubles) {
    using FD = FirDesign;
    FirDesignParams p{cfg->fft_size, cfg->internal_sample_rate, cfg->lin_log_oversampling, cfg->lowess_frac,
                      0, cfg->lowess_delta, cfg->min_value};
    std::shared_ptr<FirPlanHost> plan = FirPlanHost::get(p);
    const FirPlanView pl = plan->view(plan->blob());
    const int bins = pl.bins, anchors = pl.lw.anchors;
    std::vector<double> raw(bins), m1(bins), on_log(pl.nlog), fit(anchors), log_s(pl.nlog), m2(pl.nlog), smooth(bins);
    FirScratch s{raw.data(), m1.data(), on_log.data(), fit.data(), log_s.data(), m2.data(), smooth.data()};
    std::vector<Affine> sc(FD::Scan::SCRATCH);
#define ALL(stmt) for (int tid = 0; tid < FD::T; ++tid) { stmt; }
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
    std::vector<double> A((size_t)anchors * bins), B((size_t)bins * anchors);
    for (int col = 0; col < bins; ++col) {                       // k_fir_unit_a + k_fir_lowess + k_fir_gather_plane
        for (int k = 0; k < bins; ++k) raw[k] = k == col ? 1.0 : 0.0;
        solve(pl.s1, s.raw, s.m1);
        ALL(FD::phase_eval(tid, pl.s1, s.raw, s.m1, s.on_log))
        ALL(FD::phase_lowess_fit(tid, pl.lw, s.on_log, s.fit))
        for (int a = 0; a < anchors; ++a) A[(size_t)a * bins + col] = fit[a];
    }
    for (int col = 0; col < anchors; ++col) {                    // k_fir_unit_fit + k_fir_b + k_fir_gather_plane
        for (int a = 0; a < anchors; ++a) fit[a] = a == col ? 1.0 : 0.0;
        raw[1] = 0.0;
        ALL(FD::phase_lowess_fill(tid, pl.lw, s.fit, s.log_s))
        solve(pl.s2, s.log_s, s.m2);
        ALL(FD::phase_eval(tid, pl.s2, s.log_s, s.m2, s.smooth))
        ALL(FD::phase_pin(tid, s))
        for (int i = 0; i < bins; ++i) B[(size_t)i * anchors + col] = smooth[i];
    }
#undef ALL
    auto window = [](const double* row, int n, int& first, int& last) {          // k_fir_band
        double mx = 0.0;
        for (int j = 0; j < n; ++j) mx = std::fmax(mx, std::fabs(row[j]));
        const double cut = mx * 1e-18;
        first = n, last = 0;
        for (int j = 0; j < n; ++j)
            if (std::fabs(row[j]) > cut) { first = std::min(first, j); last = std::max(last, j + 1); }
        if (first >= last) first = last = 0;
    };
    long long asz = 0, bsz = 0;
    std::vector<double> f0(anchors);
    for (int a = 0; a < anchors; ++a) {                           // k_fir_apply_a
        int first, last;
        window(A.data() + (size_t)a * bins, bins, first, last);
        asz += last - first;
        double acc = 0.0;
        for (int j = first; j < last; ++j) acc = std::fma(A[(size_t)a * bins + j], raw_in[j], acc);
        f0[a] = acc;
    }
    for (int i = 0; i < bins; ++i) {                              // k_fir_apply_b
        int first, last;
        window(B.data() + (size_t)i * anchors, anchors, first, last);
        bsz += last - first;
        double acc = 0.0;
        for (int a = first; a < last; ++a) acc = std::fma(B[(size_t)i * anchors + a], f0[a], acc);
        smooth_out[i] = i == 0 ? 0.0 : i == 1 ? raw_in[1] : acc;
    }
    if (a_doubles) *a_doubles = asz;
    if (b_doubles) *b_doubles = bsz;
    return 0;
}

extern "C" int emu_fir_anchors(const mgx_config* cfg) {
    FirDesignParams p{cfg->fft_size, cfg->internal_sample_rate, cfg->lin_log_oversampling, cfg->lowess_frac,
                      0, cfg->lowess_delta, cfg->min_value};
    return FirPlanHost::get(p)->anchors();
}
extern "C" int emu_lowess_robust(const double* y, int n, double frac, double delta, int it, double* fit) {
    lowess(y, n, frac, delta, it, fit);
    return 0;
}
extern "C" int emu_lowess(const double* y, int n, double frac, double delta, double* fit) {
    lowess_it0(y, n, frac, delta, fit);
    return 0;
}
extern "C" int emu_spline(const double* x, const double* y, int n, const double* xq, int nq, double* out) {
    cubic_spline_nak(x, y, n, xq, nq, out);
    return 0;
}

// 1 when the closed-form quiet-chunk path of the limiter applies to these parameters, 0 when the pole gap is too small
// for its float32 difference of exponentials, < 0 when the parameters are refused (host_params.h: limiter_fill)
extern "C" int emu_limiter_quiet_ok(const mgx_config* cfg) {
    LimiterParams lp;
    if (!limiter_params(*cfg, lp).empty()) return -1;
    LimiterArgs a;
    limiter_fill(lp, (float)cfg->threshold, a);
    return a.quiet_ok;
}
