// Hyrax brick-wall limiter (matchering/limiter/hyrax.py:78-99) in ONE streaming pass.
//
// Reference data flow (every array is length n, float64):
//   rect  = max(|L|,|R|) floored at thr, / thr                    dsp.py:117-121
//   g0    = 1 - 1/rect                                            hyrax.py:87
//   sl    = centred sliding max of g0, window 2w-1, w = odd(att)  hyrax.py:35-37
//   gA    = filtfilt(1-pole rho = exp(coef/att), sl)              hyrax.py:48-51
//   sh    = trailing sliding max of sl over `hold` samples        hyrax.py:38-40
//   ho    = lfilter(butter(hold order, hold Hz), sh)              hyrax.py:61-66
//   ro    = lfilter(butter(rel order, rel Hz), max(sh, ho))       hyrax.py:68-73
//   gain  = 1 - max(g0, gA, max(ho, ro));  out = x * gain         hyrax.py:75,97,99
//
// GPU formulation.  The track is cut into chunks of C = CB*16 frames; a chunk is one T-thread
// workgroup whose thread t owns "block" t = 16 consecutive frames, GL halo blocks before the chunk's
// CB core blocks and GR after them (T = 256: four workgroups per CU; T = 1024 for attack / hold times
// whose halos would not leave a 256-block chunk enough core).
//
//  * Frames are loaded coalesced (16 B per lane); only g0 travels through LDS to the owning thread.
//    The final gain travels back the same way and the frames are re-read (L2 / Infinity Cache) for
//    the coalesced store: 8 B/frame read + 8 B/frame written reach HBM.
//  * ONE LDS plane.  The raw hard-clip gains g0 stay in the plane; both sliding maxima are windows
//    of g0 itself (sh[n] = max g0[n-hw-hb .. n+hw]) and a thread forms its 2 x 16 window maxima from
//    the plane directly: a suffix run on the left edge, a prefix run on the right edge, whole-block
//    maxima in between (those come out of the load phase: eight neighbouring lanes hold one block,
//    three DPP steps).
//  * Every recurrence is first order (state z: y[n] = b0 x[n] + z[n-1], z[n] = alpha z[n-1] +
//    beta x[n], scipy's transposed direct form II).  A thread runs its 16 frames in float32 from
//    a zero state; its block acts on the carried state as an affine map, the maps are composed
//    across the workgroup by an ordered float64 scan (wave shuffles + one LDS hop), and the exact
//    outputs are the local run plus alpha^j times the carry.  Rounding never accumulates beyond
//    16 frames.
//  * The attack smoother's pole rho = exp(coef/attack) forgets quickly: rho^HA <= 1e-7 after
//    HA ~ 8*attack frames.  The right halo is HA (+ window) frames long, so the backward run of
//    scipy.signal.filtfilt started from zero at the end of the halo is exact (to 1e-7) inside the
//    core and never needs a later chunk.  filtfilt's edge handling (odd extension by 6,
//    steady-state initial conditions) is applied by the chunks that contain frame 0 / frame n-1.
//  * The forward attack smoother, the hold and the release low-passes carry state from chunk to
//    chunk (the latter two for seconds).  Each chunk publishes the state its core frames produce
//    from a zero carry (one float64 per filter, written once); a chunk's carry is
//    sum_m (alpha^C)^m * published[chunk-1-m], truncated where (alpha^C)^m <= 1e-10 (1 chunk for
//    the attack pole, a handful for the 7 Hz hold filter, ~170 for the 0.27 Hz release filter).
//    No chunk ever waits for another chunk's look-back of the same filter, so the dependency
//    depth is two (release aggregates need the exact hold output) however long the track is.
//    A chunk is its workgroup's number (k_limit in mgx_kernels.h: the dispatcher hands workgroups out in
//    the order of their numbers, so every chunk a workgroup waits for has already started -- an
//    assumption about ONE launch that is alone on the chip, observed on this firmware and partition
//    mode, documented nowhere) or, on a handle that has ever seen a wait expire and under
//    MGX_LIMIT_TICKETS=1, a number drawn from an atomic ticket (true whatever the dispatcher does).
//    Every wait is bounded in time (wait_on below); an expired one raises the handle's error word.
//  * Order of work inside a chunk: the hold path first (its aggregate is what successors wait for
//    longest), then the attack path to completion -- so that few 16-frame arrays are live at a time --
//    and the look-back words are asked for as soon as the chunk's own aggregates are published and
//    taken as late as possible.  The forward attack carry enters linearly, so the attack path runs
//    on a zero carry while the words are in flight and the carry is added at the end:
//        gA[n] += carry * kappa * rho^(n - n0),   kappa = b0 + beta*rho / (1 - rho^2)
//    (the backward smoother's response to the decaying state, summed to infinity: the right halo is
//    >= 9 time constants long).  Only the chunk that holds the last frame of the track -- filtfilt's
//    odd extension makes its backward start depend on the forward end state -- waits for its carry
//    first.
//  * Chunks that lie strictly inside the track run a branch-free instantiation (FULL) of every phase.
//
// Published words are 8-byte granules whose value is the flag: the array is preset to all-ones
// (not a finite double) before each launch and written with one relaxed agent-scope atomic store
// (MI355X_MICROARCH.md, inter-workgroup visibility: a single naturally aligned 8-byte sc1 store,
// polled with relaxed sc1 loads, needs no fence).  Every poll loop is bounded.
#pragma once

#include "scan_util.h"

namespace mgx {

struct Iir1 {
    double b0, alpha, beta;        // y = b0*x + z_prev ; z = alpha*z_prev + beta*x
};
struct Iir1f {
    float b0, alpha, beta;
};

struct LimiterArgs {
    const float2* y;               // (n,2) level-corrected result before the final gains
    long long n;
    float2* out;                   // (n,2) limited output
    const double* gain;            // device scalar: accumulated level-correction gain
    const double* post_gain;       // device scalar: final amplitude coefficient (stages.py:203)
    const int* active;             // device flag: 0 => limiter early-out (hyrax.py:83-85)
    float threshold;
    int hw;                        // attack half window = odd(attack) - 1
    int hb;                        // hold look-back     = hold - 1
    int gl, gr, gw;                // halo blocks left / right; blocks without a full sl window
    Iir1 att, hold, rel;           // float64 coefficients (edge states, aggregates)
    Iir1f attf, holdf, relf;       // float32 copies for the per-frame arithmetic
    double pa16, ph16, pr16;       // alpha^16 in float64: the decay of a full block
    long long nchunks;
    unsigned long long* published; // [3][nchunks]: hold, release, attack chunk aggregates (bit patterns)
    const double* w_hold;          // (alpha_hold^C)^m, m = 0..n_hold-1
    const double* w_rel;
    const double* w_att;
    int n_hold, n_rel, n_att;
    int* ticket;                   // chunk dispenser (zeroed before the launch)
    int* error;                    // set to 1 if a bounded wait expired (page-locked HOST memory: written, never polled)
    int* gave_up;                  // the same fact in device memory: a waiter that runs out of time sets it, the others look
                                   // at it (zeroed with the ticket before the launch)
    // quiet chunks (limit_chunk_quiet): release state a core of zero input leaves per unit of hold carry,
    // log2 of the three poles, 0 when the closed forms do not apply (equal hold and release poles)
    double quiet_rel_gain;
    float log2_hold, log2_rel, log2_att;
    int quiet_ok;
};

constexpr unsigned long long LIMITER_UNPUBLISHED = ~0ull;

// ---- inter-workgroup words ---------------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MGX_HOST_EMU)
__device__ __forceinline__ void publish_word(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long poll_word(unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void backoff(int spins) {             // ~0.03 us at first, ~0.5 us when it drags on
    if (spins < 8) __builtin_amdgcn_s_sleep(1);
    else if (spins < 64) __builtin_amdgcn_s_sleep(4);
    else __builtin_amdgcn_s_sleep(16);
}
#else
inline void publish_word(unsigned long long* p, unsigned long long v) { *p = v; }
inline unsigned long long poll_word(unsigned long long* p) { return *p; }
inline void backoff(int) {}
#endif
// How long a wait for another workgroup's word may last.  Nothing a chunk waits for takes longer than the kernel
// itself (hundreds of microseconds); the bound exists so that a word that never comes -- the failure the handle
// recovers from, mgx.hip check_device_error -- costs a bounded time and never hangs the GPU.  It is a TIME (the
// constant 100 MHz counter), not a poll count: 50 ms, looked at every 256 polls (~0.2 ms: a wait that ends as waits
// do never gets there), and a waiter also gives up as soon as any other has (`gave_up`, a word in DEVICE memory:
// the launch is lost whatever this chunk does), so a lost launch ends about 50 ms after its first waiter started --
// the figure INTEGRATION.md quotes.  Nothing on this path reads host memory (a first version polled the handle's
// page-locked error word: every waiter's read crossed PCIe and the limiter took 0.96 ms instead of 0.13).
// `t0` = 0 on entry; `gave_up` may be null (then only the time counts).
#if defined(__HIPCC__) && !defined(MGX_HOST_EMU)
constexpr long long WAIT_BUDGET_TICKS = 5000000;                 // 50 ms of the 100 MHz counter
__device__ __forceinline__ bool wait_on(int spins, long long& t0, int* gave_up, int max_spins) {
    if (max_spins > 0) return spins < max_spins;                 // test builds: a poll count
    if ((spins & 255) != 255) return true;
    if (gave_up && __hip_atomic_load(gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
    const long long now = wall_clock64();
    if (t0 == 0) t0 = now;
    if (now - t0 < WAIT_BUDGET_TICKS) return true;
    if (gave_up) __hip_atomic_store(gave_up, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return false;
}
#else
inline bool wait_on(int spins, long long&, int*, int max_spins) { return spins < (max_spins > 0 ? max_spins : 1); }
#endif
MGX_HD unsigned long long double_bits(double v) {
    union { double d; unsigned long long u; } c;
    c.d = v;
    return c.u;
}
MGX_HD double bits_double(unsigned long long u) {
    union { double d; unsigned long long u; } c;
    c.u = u;
    return c.d;
}

template <int T_>
struct LimiterBlock {
    static constexpr int T = T_;
    static constexpr int E = 16;
    static constexpr int STRIDE = E + 1;               // LDS row stride (floats): conflict-free columns
    static constexpr int WAVES = T / 64;
    static constexpr int FRAMES = T * E;
    static constexpr int PLANE = T * STRIDE;
#ifdef MGX_TEST_LIMITER_MAX_SPINS                         // tests/test_device_errors.py: a look-back that gives up quickly
    static constexpr int MAX_SPINS = MGX_TEST_LIMITER_MAX_SPINS;
#else
    static constexpr int MAX_SPINS = 0;                // product: bounded by time (wait_on), not by a poll count
#endif
    static constexpr int POLL_SLOTS = 4;               // look-back words a lane keeps in flight (x64 lanes)

    // LDS carve (floats): [0, PLANE) g0, later the gain | [PLANE, +T) block maxima | misc
    static constexpr int BM_OFF = PLANE;
    static constexpr int MISC_OFF = BM_OFF + T;
    static constexpr int TOTALS_FLOATS = 4 * WAVES * 4;                 // four scans x WAVES Affine
    static constexpr int MISC_FLOATS = 16 + TOTALS_FLOATS + 16;        // edge sl[14] | totals | scalars
    static constexpr size_t LDS_BYTES = (size_t)(MISC_OFF + MISC_FLOATS) * 4 + 16;

    static MGX_HD float* plane(float* lds) { return lds; }
    static MGX_HD float* block_max(float* lds) { return lds + BM_OFF; }
    static MGX_HD float* edge_sl(float* lds) { return lds + MISC_OFF; }
    // scan 0 = forward attack, 1 = hold, 2 = backward attack, 3 = release
    static MGX_HD Affine* wave_totals(float* lds, int scan) {
        return reinterpret_cast<Affine*>(lds + MISC_OFF + 16) + scan * WAVES;
    }
    //   scalars: [0] hold carry, [1] release carry, [2] attack carry, [4] ticket
    static MGX_HD double* scalars(float* lds) { return reinterpret_cast<double*>(lds + MISC_OFF + 16 + TOTALS_FLOATS); }
    static MGX_HD int gidx(int i) { return (i >> 4) * STRIDE + (i & 15); }

    struct Geometry {
        int gl, gr, gw, core_blocks, chunk;
    };
    // ha = frames after which the attack pole has decayed to 1e-8
    static MGX_HD Geometry geometry(int hw, int hb, int ha) {
        Geometry g;
        g.gw = (hw + E - 1) / E;
        const int hab = (ha + E - 1) / E;
        g.gl = (hw + hb + E - 1) / E;          // left: only the sh window (the attack state is carried in)
        g.gr = hab + g.gw;                      // right: backward warm-up + sl window
        g.core_blocks = T - g.gl - g.gr;
        g.chunk = g.core_blocks * E;
        return g;
    }
    static MGX_HD long long region_start(long long chunk, const LimiterArgs& a) {
        return chunk * (long long)((T - a.gl - a.gr) * E) - (long long)a.gl * E;
    }
    // the chunk whose region reaches the end of the track: its attack carry is taken up front
    static MGX_HD bool tail_chunk(long long chunk, const LimiterArgs& a) {
        return region_start(chunk, a) + FRAMES >= a.n;
    }

    // a chunk whose region lies strictly inside the track: every block has 16 frames, no filtfilt edge,
    // plain loads.  All but the first and the last one or two chunks of a track: the phases are compiled
    // twice, FULL = true without any of the per-frame validity tests.
    static MGX_HD bool full_chunk(long long chunk, const LimiterArgs& a) {
        const long long r0 = region_start(chunk, a);
        return r0 >= 0 && r0 + FRAMES < a.n;
    }

    struct Thread {
        long long base;
        int valid;
        bool core, has_sl;
        bool inject_left, inject_right;
        double edge_state;
        double att_decay;                 // decay of the forward attack state from the chunk's first sl frame to this block
        Affine hold_pre;                  // composition of the hold maps of the blocks before this one
        float inner;                      // max g0 over the part of the attack window every frame of the block shares
        float sl[E], sh[E], yf[E], yb[E], x2[E], mx[E];
    };

    // ---- hard-clip gain (dsp.py:117-121, hyrax.py:87): 1 - thr/amax above the threshold -----------
    // amax - thr is exact for amax < 2 thr (Sterbenz); the reciprocal is good to 1 ulp.
    static MGX_HD float gain_of(float2 v, float thr) {
        const float amax = fmaxf(fabsf(v.x), fabsf(v.y));
        return amax > thr ? (amax - thr) * fast_rcp(amax) : 0.f;
    }
    static MGX_HD float2 scaled(float2 y, float g) { return make_float2(y.x * g, y.y * g); }
    static MGX_HD float own_gain(float2 v, float k, bool with_gain, float thr) {
        return with_gain ? fminf(k, 1.0f - gain_of(v, thr)) : 1.f;
    }

    // ---- P1: coalesced load, g0 -> plane; pm[j] = max of this lane's two frames of iteration j ----
    // (lanes 8q .. 8q+7 of iteration j hold block q + 32 j: the kernel folds them into the block maxima)
    template <bool FULL = false>
    static MGX_HD void phase_load(int tid, long long chunk, const LimiterArgs& a, float* lds, float (&pm)[E / 2]) {
        const long long r0 = region_start(chunk, a);
        const bool interior = FULL || (r0 >= 0 && r0 + FRAMES <= a.n);
        const float g = (float)*a.gain;
        float* gp = plane(lds);
        MGX_UNROLL
        for (int j = 0; j < E / 2; ++j) {
            const int i = 2 * tid + 2 * T * j;
            float2 v0 = make_float2(0.f, 0.f), v1 = v0;
            if (interior) {
                const float4 q = *reinterpret_cast<const float4*>(a.y + (r0 + i));
                v0 = make_float2(q.x, q.y);
                v1 = make_float2(q.z, q.w);
            } else {
                const long long f = r0 + i;
                if (f >= 0 && f < a.n) v0 = a.y[f];
                if (f + 1 >= 0 && f + 1 < a.n) v1 = a.y[f + 1];
            }
            const float g0 = gain_of(scaled(v0, g), a.threshold), g1 = gain_of(scaled(v1, g), a.threshold);
            gp[gidx(i)] = g0;
            gp[gidx(i + 1)] = g1;
            pm[j] = pmax(g0, g1);
        }
    }
    static MGX_HD int block_of(int tid, int j) { return (tid >> 3) + (T / 8) * j; }
    // the same for a chunk inside the track, keeping the frames: a quiet chunk (below) stores from them
    struct Reload { float4 q[E / 2]; };
    static MGX_HD void phase_load_full(int tid, long long chunk, const LimiterArgs& a, float* lds, float (&pm)[E / 2],
                                       Reload& kept) {
        const float2* y = a.y + region_start(chunk, a) + 2 * tid;
        const float g = (float)*a.gain;
        float* gp = plane(lds);
        MGX_UNROLL
        for (int j = 0; j < E / 2; ++j) kept.q[j] = *reinterpret_cast<const float4*>(y + 2 * T * j);
        MGX_UNROLL
        for (int j = 0; j < E / 2; ++j) {
            const int i = 2 * tid + 2 * T * j;
            const float4 q = kept.q[j];
            const float g0 = gain_of(scaled(make_float2(q.x, q.y), g), a.threshold);
            const float g1 = gain_of(scaled(make_float2(q.z, q.w), g), a.threshold);
            gp[gidx(i)] = g0;
            gp[gidx(i + 1)] = g1;
            pm[j] = pmax(g0, g1);
        }
    }

    // ---- window maxima from the raw plane -----------------------------------------------------------
    // value at frame 16*tid + d of the region, d uniform
    static MGX_HD int rel(int d) { return (d >> 4) * STRIDE + (d & 15); }
    // max over frames 16*tid + [a, b] (uniform, may be empty): ragged ends frame by frame (at most 15
    // each, masked reads), whole blocks from the block maxima
    static MGX_HD float range_max(int tid, int a, int b, const float* lds) {
        const float* row = plane(const_cast<float*>(lds)) + tid * STRIDE;
        const float* bm = block_max(const_cast<float*>(lds)) + tid;
        const int len = b - a + 1;
        if (len <= 0) return 0.f;
        const int to_edge = (16 - (a & 15)) & 15;                 // frames up to the next block boundary
        const int nl = to_edge < len ? to_edge : len;
        const int a2 = a + nl, rest = len - nl;
        const int nb = rest >> 4, nr = rest & 15;
        float m = 0.f;
        const float* pl = row + rel(a);                           // nl <= 15 frames inside one row
        MGX_UNROLL
        for (int k = 0; k < 15; ++k) {
            const float v = pl[k < nl ? k : 0];
            m = pmax(m, k < nl ? v : 0.f);
        }
        for (int k = 0; k < nb; ++k) m = pmax(m, bm[(a2 >> 4) + k]);
        const float* pr = row + rel(a2 + 16 * nb);                // a block boundary: nr <= 15 frames of one row
        MGX_UNROLL
        for (int k = 0; k < 15; ++k) {
            const float v = pr[k < nr ? k : 0];
            m = pmax(m, k < nr ? v : 0.f);
        }
        // (with nl == 0 and nr == 0 the two rows above are read at offset 0 only: rel(a), rel(a2) lie inside the region)
        return m;
    }
    // 15 consecutive frames starting at 16*tid + d0 (they cross at most one row boundary)
    static MGX_HD void run15(int tid, int d0, const float* lds, float (&v)[15]) {
        const float* row = plane(const_cast<float*>(lds)) + tid * STRIDE;
        const float* p = row + rel(d0);
        const int cross = 16 - (d0 & 15);                          // frames before the pad slot
        MGX_UNROLL
        for (int k = 0; k < 15; ++k) {
            const float* q = k < cross ? p : p + 1;
            v[k] = q[k];
        }
    }
    // sl[j] = max g0[c - hw .. c + hw], sh[j] = max g0[c - hw - hb .. c + hw] for c = 16 tid + j:
    //   window = [j - lw, 14 - lw] (suffix run)  u  [15 - lw, hw] (always inside)  u  [hw + 1, hw + j] (prefix run)
    // One 15-frame run is in registers at a time (the register budget is the point of this kernel): the
    // left run right-to-left into the output, then the right run left-to-right.  `inner` = the maximum
    // over the always-inside part.
    static MGX_HD void window16(int tid, int lw, int hw, float inner, const float* lds, float (&out)[E]) {
        float r[15];
        run15(tid, -lw, lds, r);
        float s = 0.f;
        out[15] = inner;
        MGX_UNROLL
        for (int j = 14; j >= 0; --j) {
            s = pmax(s, r[j]);
            out[j] = pmax(inner, s);
        }
        run15(tid, hw + 1, lds, r);
        s = 0.f;
        MGX_UNROLL
        for (int j = 1; j < E; ++j) {
            s = pmax(s, r[j - 1]);
            out[j] = pmax(out[j], s);
        }
    }
    // windows shorter than 15 frames (attack times of a few samples): the three parts above would
    // overlap, so every frame's window is read out directly
    static MGX_HD bool short_window(int lw, int hw) { return lw + hw < 14; }
    static MGX_HD void window_direct(int tid, int lw, int hw, const float* lds, float (&out)[E]) {
        const float* row = plane(const_cast<float*>(lds)) + tid * STRIDE;
        MGX_UNROLL
        for (int j = 0; j < E; ++j) {
            float m = 0.f;
            for (int d = j - lw; d <= j + hw; ++d) m = pmax(m, row[rel(d)]);
            out[j] = m;
        }
    }

    // ---- first-order recurrences over a thread's 16 frames, float32 ---------------------------------
    // state after `count` frames starting from `z0` (frames >= count do not advance the state)
    static MGX_HD float run_forward(const Iir1f& f, const float (&x)[E], int count, float z0) {
        float s = z0;
        MGX_UNROLL
        for (int j = 0; j < E; ++j)
            if (j < count) s = fmaf(f.alpha, s, f.beta * x[j]);
        return s;
    }
    static MGX_HD float run_backward(const Iir1f& f, const float (&x)[E], int count, float z0) {
        float s = z0;
        MGX_UNROLL
        for (int j = E - 1; j >= 0; --j)
            if (j < count) s = fmaf(f.alpha, s, f.beta * x[j]);
        return s;
    }
    // outputs y[j] = b0 x[j] + z[j-1] with the true state z0 entering the block; returns the state
    // after the block.  Rounding accumulates over at most 16 frames (the carry is exact float64
    // rounded once).
    static MGX_HD float out_forward(const Iir1f& f, const float (&x)[E], int count, float z0, float (&y)[E]) {
        float s = z0;
        MGX_UNROLL
        for (int j = 0; j < E; ++j) {
            if (j < count) {
                y[j] = fmaf(f.b0, x[j], s);
                s = fmaf(f.alpha, s, f.beta * x[j]);
            } else {
                y[j] = 0.f;
            }
        }
        return s;
    }
    static MGX_HD void out_backward(const Iir1f& f, const float (&x)[E], int count, float z0, float (&y)[E]) {
        float s = z0;
        MGX_UNROLL
        for (int j = E - 1; j >= 0; --j) {
            if (j < count) {
                y[j] = fmaf(f.b0, x[j], s);
                s = fmaf(f.alpha, s, f.beta * x[j]);
            } else {
                y[j] = 0.f;
            }
        }
    }
    // alpha^count for the aggregate of a block with `count` valid frames: table[E] for a full block
    static MGX_HD double block_decay(double full, double alpha, int count) {
        if (count == E) return full;
        double r = 1.0;
        for (int i = 0; i < count; ++i) r *= alpha;
        return r;
    }

    // scipy.signal.filtfilt edges (padtype 'odd', padlen 6, lfilter_zi), float64
    static MGX_HD double filtfilt_left_state(const Iir1& f, const float* sl0 /* sl[0..6] */) {
        const double x0 = (double)sl0[0];
        const double zi = f.beta / (1.0 - f.alpha);
        double z = 0.0;
        for (int i = 0; i < 6; ++i) {
            const double e = 2.0 * x0 - (double)sl0[6 - i];
            if (i == 0) z = zi * e;
            z = fma(f.alpha, z, f.beta * e);
        }
        return z;
    }
    // sl_end = sl[n-7 .. n-1]; z_end = forward state after frame n-1.  Returns the backward state
    // entering frame n-1.
    static MGX_HD double filtfilt_right_state(const Iir1& f, const float* sl_end, double z_end) {
        const double xl = (double)sl_end[6];
        const double zi = f.beta / (1.0 - f.alpha);
        double yfe[6];
        double z = z_end;
        for (int i = 0; i < 6; ++i) {
            const double e = 2.0 * xl - (double)sl_end[5 - i];
            yfe[i] = fma(f.b0, e, z);
            z = fma(f.alpha, z, f.beta * e);
        }
        double zb = zi * yfe[5];
        for (int i = 5; i >= 0; --i) zb = fma(f.alpha, zb, f.beta * yfe[i]);
        return zb;
    }


    // ---- quiet neighbourhoods ------------------------------------------------------------------------
    // Everything a thread's two windows read lies in the blocks tid - ceil((hw + hb)/16) .. tid + (hw + 15)/16;
    // when their maxima are all zero (no frame above the threshold: most of a track that is not mastered
    // "hot"), sl and sh of the block are zero without reading the plane.  The kernel votes per wave
    // (`busy` = any lane has a non-zero neighbourhood) so that the branch is uniform; the results are the
    // same either way, and the host emulation always takes the long way.
    static MGX_HD float neighbourhood_max(int tid, const LimiterArgs& a, const float* lds) {
        if (!(tid >= a.gl && tid < T - a.gw)) return 0.f;           // blocks without windows (has_sl is false)
        const float* bm = block_max(const_cast<float*>(lds)) + tid;
        const int back = (a.hw + a.hb + E - 1) / E, ahead = (a.hw + E - 1) / E;   // = gl, gw: inside the region
        float m = 0.f;
        for (int d = -back; d <= ahead; ++d) m = pmax(m, bm[d]);
        return m;
    }

    // ---- P2: block geometry; sh; block map of the hold filter ----------------------------------------
    // The hold path goes first: its aggregate is what successors wait for longest.
    template <bool FULL = false>
    static MGX_HD Affine phase_hold_window(int tid, long long chunk, const LimiterArgs& a, Thread& th, const float* lds,
                                           bool busy = true) {
        th.base = region_start(chunk, a) + (long long)tid * E;
        th.core = tid >= a.gl && tid < T - a.gr;
        th.has_sl = tid >= a.gl && tid < T - a.gw;
        const long long left = a.n - th.base;
        th.valid = FULL ? E : (th.base < 0 ? 0 : (left >= E ? E : (left > 0 ? (int)left : 0)));
        const int valid = FULL ? E : th.valid;
        th.inject_left = false;
        th.inject_right = false;
        th.edge_state = 0.0;
        th.inner = 0.f;
        Affine r = affine_identity();
        MGX_UNROLL
        for (int j = 0; j < E; ++j) th.sh[j] = 0.f;
        const bool short_sl = short_window(a.hw, a.hw);                       // uniform
        if (busy && th.has_sl && !short_sl) th.inner = range_max(tid, 15 - a.hw, a.hw, lds);
        if (th.core) {
            if (busy) {
                const int lw = a.hw + a.hb;
                if (short_window(lw, a.hw)) window_direct(tid, lw, a.hw, lds, th.sh);
                else if (short_sl) window16(tid, lw, a.hw, range_max(tid, 15 - lw, a.hw, lds), lds, th.sh);
                else window16(tid, lw, a.hw, pmax(th.inner, range_max(tid, 15 - lw, 14 - a.hw, lds)), lds, th.sh);
                if (!FULL) {
                    MGX_UNROLL
                    for (int j = 0; j < E; ++j)
                        if (j >= valid) th.sh[j] = 0.f;                       // windows are truncated at the array ends
                }
            }
            if (valid > 0)
                r = Affine{block_decay(a.ph16, a.hold.alpha, valid),
                           busy ? (double)run_forward(a.holdf, th.sh, valid, 0.f) : 0.0};
        }
        return r;
    }
    // ---- P3: sl; block map of the forward attack smoother ------------------------------------------
    template <bool FULL = false>
    static MGX_HD Affine phase_attack_window(int tid, const LimiterArgs& a, Thread& th, float* lds, bool busy = true) {
        const int valid = FULL ? E : th.valid;
        Affine r = affine_identity();
        MGX_UNROLL
        for (int j = 0; j < E; ++j) th.sl[j] = 0.f;
        if (th.has_sl) {
            if (busy) {
                if (short_window(a.hw, a.hw)) window_direct(tid, a.hw, a.hw, lds, th.sl);
                else window16(tid, a.hw, a.hw, th.inner, lds, th.sl);
            }
            if (!FULL) {
                MGX_UNROLL
                for (int j = 0; j < E; ++j)
                    if (j >= valid) th.sl[j] = 0.f;
                MGX_UNROLL
                for (int j = 0; j < E; ++j) {
                    const long long f = th.base + j;
                    if (j < valid && f >= a.n - 7) edge_sl(lds)[7 + (int)(f - (a.n - 7))] = th.sl[j];
                }
                th.inject_left = th.base == 0;
                th.inject_right = valid > 0 && th.base + valid == a.n;
            }
            if (valid > 0) {
                const double decay = block_decay(a.pa16, a.att.alpha, valid);
                const double zend = busy ? (double)run_forward(a.attf, th.sl, valid, 0.f) : 0.0;
                r = Affine{decay, zend};
                if (!FULL && th.inject_left) {
                    th.edge_state = filtfilt_left_state(a.att, th.sl);
                    r = Affine{0.0, fma(decay, th.edge_state, zend)};
                }
            }
        }
        return r;
    }

    // ---- look-back words, split into "ask" and "take" ----------------------------------------------
    struct Polls {
        unsigned long long v[POLL_SLOTS];
    };
    static MGX_HD void lookback_publish(long long chunk, int slot, const LimiterArgs& a, double b) {
#ifdef MGX_TEST_LOSE_WORD                                  // the same test build: chunk 1 never publishes its hold word
        if (slot == 0 && chunk == 1) return;
#endif
        publish_word(a.published + (size_t)slot * a.nchunks + chunk, double_bits(b));
    }
    static MGX_HD void lookback_ask(int lane, long long chunk, int slot, const LimiterArgs& a, Polls& p) {
        const int count = slot == 0 ? a.n_hold : (slot == 1 ? a.n_rel : a.n_att);
        MGX_UNROLL
        for (int k = 0; k < POLL_SLOTS; ++k) {
            const int m = lane + 64 * k;
            const long long c = chunk - 1 - m;
            p.v[k] = 0ull;                                          // +0.0: a word that does not exist adds nothing
            if (m < count && c >= 0) p.v[k] = poll_word(a.published + (size_t)slot * a.nchunks + c);
        }
    }
    // this lane's share of sum_m w[m] * published[chunk-1-m] (the caller adds the 64 shares)
    static MGX_HD double lookback_take(int lane, long long chunk, int slot, const LimiterArgs& a, Polls& p) {
        const double* w = slot == 0 ? a.w_hold : (slot == 1 ? a.w_rel : a.w_att);
        const int count = slot == 0 ? a.n_hold : (slot == 1 ? a.n_rel : a.n_att);
        double acc = 0.0;
        MGX_UNROLL
        for (int k = 0; k < POLL_SLOTS; ++k) {
            const int m = lane + 64 * k;
            const long long c = chunk - 1 - m;
            if (m < count && c >= 0) {
                unsigned long long* q = a.published + (size_t)slot * a.nchunks + c;
                unsigned long long v = p.v[k];
                int spins = 0;
                long long t0 = 0;
                while (v == LIMITER_UNPUBLISHED && wait_on(spins, t0, a.gave_up, MAX_SPINS)) {
                    backoff(spins);
                    v = poll_word(q);
                    ++spins;
                }
                if (v == LIMITER_UNPUBLISHED) {
                    *a.error = 1;
                    v = 0;
                }
                acc = fma(w[m], bits_double(v), acc);
            }
        }
        // filters with a longer memory than POLL_SLOTS * 64 chunks: the remaining words, one at a time
        for (int m = lane + 64 * POLL_SLOTS; m < count; m += 64) {
            const long long c = chunk - 1 - m;
            if (c < 0) break;
            unsigned long long* q = a.published + (size_t)slot * a.nchunks + c;
            unsigned long long v = poll_word(q);
            int spins = 0;
            long long t0 = 0;
            while (v == LIMITER_UNPUBLISHED && wait_on(spins, t0, a.gave_up, MAX_SPINS)) {
                backoff(spins);
                v = poll_word(q);
                ++spins;
            }
            if (v == LIMITER_UNPUBLISHED) {
                *a.error = 1;
                v = 0;
            }
            acc = fma(w[m], bits_double(v), acc);
        }
        return acc;
    }

    // ---- P3: forward attack output from carry `att_now` (zero unless tail chunk); block map of the
    //          backward smoother (right-to-left scan).  `p0` = forward attack prefix of this thread.
    template <bool FULL = false>
    static MGX_HD Affine phase_attack_forward(int tid, const LimiterArgs& a, Thread& th, Affine p0, double att_now,
                                              const float* lds) {
        Affine r = affine_identity();
        const int valid = FULL ? E : th.valid;
        th.att_decay = p0.a;
        MGX_UNROLL
        for (int j = 0; j < E; ++j) th.yf[j] = 0.f;
        if (th.has_sl) {
            double c = affine_apply(p0, att_now);
            if (!FULL && th.inject_left) c = th.edge_state;
            const float zend = out_forward(a.attf, th.sl, valid, (float)c, th.yf);
            if (valid > 0) {
                const double decay = block_decay(a.pa16, a.att.alpha, valid);
                const double zb = (double)run_backward(a.attf, th.yf, valid, 0.f);
                r = Affine{decay, zb};
                if (!FULL && th.inject_right) {
                    th.edge_state = filtfilt_right_state(a.att, edge_sl(const_cast<float*>(lds)) + 7, (double)zend);
                    r = Affine{0.0, fma(decay, th.edge_state, zb)};
                }
            }
        }
        return r;
    }
    // ---- P4: backward attack output (carry-free part).  `pb` = composition of the blocks to the right
    template <bool FULL = false>
    static MGX_HD void phase_attack_backward(int tid, const LimiterArgs& a, Thread& th, Affine pb) {
        MGX_UNROLL
        for (int j = 0; j < E; ++j) th.yb[j] = 0.f;
        if (!th.core) return;
        double cb = affine_apply(pb, 0.0);
        if (!FULL && th.inject_right) cb = th.edge_state;
        out_backward(a.attf, th.yf, FULL ? E : th.valid, (float)cb, th.yb);
    }
    // kappa of the file header
    static MGX_HD double attack_kappa(const Iir1& f) { return f.b0 + f.beta * f.alpha / (1.0 - f.alpha * f.alpha); }

    // ---- P5: carries have arrived.  Exact hold output, attack carry term, max(sh, ho) -> block map of
    //          the release filter.  att_deferred = the attack carry not yet applied (0 in a tail chunk)
    template <bool FULL = false>
    static MGX_HD Affine phase_hold(int tid, const LimiterArgs& a, Thread& th, double hold_carry, double att_deferred) {
        Affine r = affine_identity();
        const int valid = FULL ? E : th.valid;
        MGX_UNROLL
        for (int j = 0; j < E; ++j) { th.x2[j] = 0.f; th.mx[j] = 0.f; }
        if (!th.core) return r;
        float pw = (float)(att_deferred * attack_kappa(a.att) * th.att_decay);
        // hold output (out_forward inlined so that sh[j], yb[j] die as x2[j], mx[j] are born)
        float z = (float)affine_apply(th.hold_pre, hold_carry);
        MGX_UNROLL
        for (int j = 0; j < E; ++j) {
            const bool in = j < valid;
            const float ho = in ? fmaf(a.holdf.b0, th.sh[j], z) : 0.f;
            if (in) z = fmaf(a.holdf.alpha, z, a.holdf.beta * th.sh[j]);
            const float ga = in ? th.yb[j] + pw : 0.f;
            pw *= a.attf.alpha;
            th.x2[j] = fmaxf(th.sh[j], ho);                      // hyrax.py:73
            th.mx[j] = fmaxf(ho, ga);
        }
        if (valid > 0)
            r = Affine{block_decay(a.pr16, a.rel.alpha, valid), (double)run_forward(a.relf, th.x2, valid, 0.f)};
        return r;
    }

    // ---- P6: release output -> gain -> plane ---------------------------------------------------------
    template <bool FULL = false>
    static MGX_HD void phase_gain(int tid, const LimiterArgs& a, Thread& th, Affine pr, double rel_carry, float* lds) {
        if (!th.core) return;
        const int valid = FULL ? E : th.valid;
        float z = (float)affine_apply(pr, rel_carry);
        float* gn = plane(lds) + tid * STRIDE;
        MGX_UNROLL
        for (int j = 0; j < E; ++j) {
            const bool in = j < valid;
            const float ro = in ? fmaf(a.relf.b0, th.x2[j], z) : 0.f;
            if (in) z = fmaf(a.relf.alpha, z, a.relf.beta * th.x2[j]);
            gn[j] = 1.0f - fmaxf(th.mx[j], ro);                  // hyrax.py:75,97 without g0 (phase_store)
        }
    }

    // ---- quiet chunks -------------------------------------------------------------------------------------
    // A chunk whose whole region (halos included) holds no frame above the threshold has g0 = sl = sh = 0:
    // its hold and attack aggregates are zero, and with the carries hc (hold), ac (attack), zr (release)
    // entering its first core frame the envelopes are geometric sequences in the core offset k:
    //   ho[k] = hc ah^k                      (hyrax.py:66 on zero input: the filter state decays)
    //   x2[k] = max(0, ho[k]) = ho[k]        (hyrax.py:73; carries are sums of non-negative terms)
    //   ro[k] = b0r ho[k] + ar^k zr + br hc (ar^k - ah^k) / (ar - ah)
    //   gA[k] = ac kappa rho^k               (the deferred attack carry term of phase_hold)
    // and the release state it hands on is hc * quiet_rel_gain (host_params.h).  No windows, no scans, no
    // reload: the frames wait in registers from the load.  Powers by v_exp_f32 (2e-7 relative).
    static MGX_HD void phase_quiet_store(int tid, long long chunk, const LimiterArgs& a, const Reload& r, double hold_carry,
                                         double att_carry, double rel_carry) {
        const long long r0 = region_start(chunk, a);
        const int c0 = a.gl * E, c1 = (T - a.gr) * E;
        const float g = (float)*a.gain, post = (float)*a.post_gain;
        const float hc = (float)hold_carry, zr0 = (float)rel_carry;
        const float ak = (float)(att_carry * attack_kappa(a.att));
        const float d = (float)(a.rel.beta * hold_carry / (a.rel.alpha - a.hold.alpha));
        MGX_UNROLL
        for (int j = 0; j < E / 2; ++j) {
            const int i = 2 * tid + 2 * T * j;
            if (i < c0 || i >= c1) continue;
            const float k = (float)(i - c0);
            float eh = exp2f(k * a.log2_hold), er = exp2f(k * a.log2_rel), ea = exp2f(k * a.log2_att);
            float s[2];
            MGX_UNROLL
            for (int u = 0; u < 2; ++u) {
                const float ho = hc * eh;
                const float ro = fmaf(a.relf.b0, ho, fmaf(er, zr0, d * (er - eh)));
                s[u] = (1.0f - fmaxf(fmaxf(ak * ea, ho), ro)) * post;
                eh *= a.holdf.alpha;
                er *= a.relf.alpha;
                ea *= a.attf.alpha;
            }
            const float4 q = r.q[j];
            const float2 v0 = scaled(make_float2(q.x, q.y), g), v1 = scaled(make_float2(q.z, q.w), g);
            st_stream(reinterpret_cast<float4*>(a.out + (r0 + i)), make_float4(v0.x * s[0], v0.y * s[0], v1.x * s[1], v1.y * s[1]));
        }
    }

    // ---- P7: coalesced reload, apply gain, store ---------------------------------------------------
    template <bool FULL = false>
    static MGX_HD void phase_store(int tid, long long chunk, const LimiterArgs& a, bool with_gain, const float* lds) {
        const long long r0 = region_start(chunk, a);
        const long long c0 = r0 + (long long)a.gl * E, c1 = r0 + (long long)(T - a.gr) * E;
        const bool interior = FULL || (r0 >= 0 && r0 + FRAMES <= a.n);
        const float g = (float)*a.gain, post = (float)*a.post_gain;
        const float* gn = plane(const_cast<float*>(lds));
        MGX_UNROLL
        for (int j = 0; j < E / 2; ++j) {
            const int i = 2 * tid + 2 * T * j;
            const long long f = r0 + i;
            if (f < c0 || f >= c1) continue;
            const float k0 = with_gain ? gn[gidx(i)] : 1.f, k1 = with_gain ? gn[gidx(i + 1)] : 1.f;
            if (interior) {
                const float4 q = *reinterpret_cast<const float4*>(a.y + f);
                const float2 v0 = scaled(make_float2(q.x, q.y), g), v1 = scaled(make_float2(q.z, q.w), g);
                const float s0 = own_gain(v0, k0, with_gain, a.threshold) * post;
                const float s1 = own_gain(v1, k1, with_gain, a.threshold) * post;
                st_stream(reinterpret_cast<float4*>(a.out + f), make_float4(v0.x * s0, v0.y * s0, v1.x * s1, v1.y * s1));
            } else {
                if (f < a.n) {
                    const float2 v = scaled(a.y[f], g);
                    const float s = own_gain(v, k0, with_gain, a.threshold) * post;
                    a.out[f] = make_float2(v.x * s, v.y * s);
                }
                if (f + 1 < a.n) {
                    const float2 v = scaled(a.y[f + 1], g);
                    const float s = own_gain(v, k1, with_gain, a.threshold) * post;
                    a.out[f + 1] = make_float2(v.x * s, v.y * s);
                }
            }
        }
    }

    // The same for a chunk inside the track, in two halves: the reload is issued while the release
    // look-back is in flight (the kernel has nothing else to do there), the rest follows the gains.
    static MGX_HD void phase_reload(int tid, long long chunk, const LimiterArgs& a, Reload& r) {
        // (every frame of the region exists; only the core is stored, so only the core is fetched again:
        // the halos are 13 % of a 256-block region)
        const float2* y = a.y + region_start(chunk, a) + 2 * tid;
        const int c0 = a.gl * E, c1 = (T - a.gr) * E;
        MGX_UNROLL
        for (int j = 0; j < E / 2; ++j) {
            const int i = 2 * tid + 2 * T * j;
            r.q[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i >= c0 && i < c1) r.q[j] = *reinterpret_cast<const float4*>(y + 2 * T * j);
        }
    }
    static MGX_HD void phase_store_reloaded(int tid, long long chunk, const LimiterArgs& a, const Reload& r, const float* lds) {
        const long long r0 = region_start(chunk, a);
        const long long c0 = r0 + (long long)a.gl * E, c1 = r0 + (long long)(T - a.gr) * E;
        const float g = (float)*a.gain, post = (float)*a.post_gain;
        const float* gn = plane(const_cast<float*>(lds));
        MGX_UNROLL
        for (int j = 0; j < E / 2; ++j) {
            const int i = 2 * tid + 2 * T * j;
            const long long f = r0 + i;
            if (f < c0 || f >= c1) continue;
            const float4 q = r.q[j];
            const float2 v0 = scaled(make_float2(q.x, q.y), g), v1 = scaled(make_float2(q.z, q.w), g);
            const float s0 = own_gain(v0, gn[gidx(i)], true, a.threshold) * post;
            const float s1 = own_gain(v1, gn[gidx(i + 1)], true, a.threshold) * post;
            st_stream(reinterpret_cast<float4*>(a.out + f), make_float4(v0.x * s0, v0.y * s0, v1.x * s1, v1.y * s1));
        }
    }
};

}  // namespace mgx
