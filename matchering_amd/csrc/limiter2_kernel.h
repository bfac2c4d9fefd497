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
// GPU formulation.  The track is cut into chunks of C = CB*16 frames; a chunk is one 256-thread
// workgroup (four fit a CU) whose thread t owns "block" t = 16 consecutive frames, GL halo blocks
// before the chunk's CB core blocks and GR after them.
//
//  * Frames are loaded coalesced (16 B per lane); only g0 travels through LDS to the owning
//    thread.  The final gain travels back the same way and the frames are re-read (L2) for the
//    coalesced store: 8 B/frame read + 8 B/frame written reach HBM.
//  * Both sliding maxima are windows of g0 itself (sh[n] = max g0[n-hw-hb .. n+hw]): per-block
//    prefix/suffix maxima in LDS give each output with two LDS reads (van Herk).
//  * Every recurrence is first order (state z: y[n] = b0 x[n] + z[n-1], z[n] = alpha z[n-1] +
//    beta x[n], scipy's transposed direct form II).  A thread runs its 16 frames in float32 from
//    a zero state; its block acts on the carried state as an affine map, the maps are composed
//    across the workgroup by an ordered float64 scan (scan_util.h), and the exact outputs are the
//    local run plus alpha^j times the carry.  Rounding never accumulates beyond 16 frames.
//  * The attack smoother's pole rho = exp(coef/attack) forgets quickly: rho^HA <= 1e-8 after
//    HA ~ 9*attack frames.  The right halo is HA (+ window) frames long, so the backward run of
//    scipy.signal.filtfilt started from zero at the end of the halo is exact (to 1e-8) inside the
//    core and never needs a later chunk.  filtfilt's edge handling (odd extension by 6,
//    steady-state initial conditions) is applied by the chunks that contain frame 0 / frame n-1.
//  * The forward attack smoother, the hold and the release low-passes carry state from chunk to
//    chunk (the latter two for seconds).  Each chunk publishes the state its core frames produce
//    from a zero carry (one float64 per filter, written once); a chunk's carry is
//    sum_m (alpha^C)^m * published[chunk-1-m], truncated where (alpha^C)^m <= 1e-10 (1 chunk for
//    the attack pole, a handful for the 7 Hz hold filter, ~170 for the 0.27 Hz release filter).
//    No chunk ever waits for another chunk's look-back of the same filter, so the dependency
//    depth is two (release aggregates need the exact hold output) however long the track is.
//    Chunk numbers are drawn from an atomic ticket, so every chunk a workgroup waits for has
//    already started.
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

struct Limiter2Args {
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
    int* error;                    // set to 1 if a bounded wait expired
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

struct Limiter2Block {
    static constexpr int T = 256;
    static constexpr int E = 16;
    static constexpr int STRIDE = E + 1;               // LDS row stride (floats): conflict-free columns
    static constexpr int WAVES = T / 64;
    static constexpr int FRAMES = T * E;               // frames a workgroup touches
    static constexpr int PLANE = T * STRIDE;           // floats
    static constexpr int MAX_SPINS = 1 << 20;            // x ~0.2 us of s_sleep: a fifth of a second

    // LDS carve (floats).  The window planes are dead once sl/sh are in registers, so the gain
    // plane reuses plane 0.
    //   [0, 2*PLANE)              planes: g0 -> prefix maxima | suffix maxima ; later [0, PLANE) = gain
    //   [2*PLANE, +T)             per-block maxima
    //   [MISC_OFF, ...)           edge samples, wave totals of the scans, broadcast scalars (never aliased)
    static constexpr int BM_OFF = 2 * PLANE;
    static constexpr int MISC_OFF = BM_OFF + T;
    static constexpr int MISC_FLOATS = 16 + 4 * 2 * WAVES * 2 + 16;   // edge sl[14] | totals | scalars
    static constexpr size_t LDS_BYTES = (size_t)(MISC_OFF + MISC_FLOATS) * 4 + 16;

    static MGX_HD float* plane(float* lds, int i) { return lds + i * PLANE; }
    static MGX_HD float* block_max(float* lds) { return lds + BM_OFF; }
    static MGX_HD float* gain_plane(float* lds) { return lds; }
    static MGX_HD float* edge_sl(float* lds) { return lds + MISC_OFF; }                 // [14]
    // wave totals of scan k (0/1) of round r (0/1): WAVES Affine each
    static MGX_HD Affine* wave_totals(float* lds, int round, int k) {
        return reinterpret_cast<Affine*>(lds + MISC_OFF + 16) + (round * 2 + k) * WAVES;
    }
    static MGX_HD double* scalars(float* lds) { return reinterpret_cast<double*>(lds + MISC_OFF + 16 + 4 * 2 * WAVES * 2); }
    //   scalars: [0] hold chunk carry, [1] release chunk carry, [2] attack chunk carry, [4] ticket
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

    struct Thread {
        long long base;           // first frame of this thread's block
        int valid;                // frames of the block inside [0, n)
        bool core, has_sl;
        float sl[E], sh[E];
        float yf[E], x2[E], m[E];         // attack forward output; max(sh,ho); max(ho,ro)
        bool inject_left, inject_right;   // this block starts at frame 0 / holds frame n-1
        double edge_state;                // filtfilt state to inject (left: entering frame 0; right: entering n-1)
    };

    static MGX_HD long long region_start(long long chunk, const Limiter2Args& a) {
        return chunk * (long long)((T - a.gl - a.gr) * E) - (long long)a.gl * E;
    }

    // ---- P1: coalesced load, g0 -> LDS (natural order) --------------------------------------
    static MGX_HD float gain_of(float2 v, float thr) {
        const float amax = fmaxf(fabsf(v.x), fabsf(v.y));
        // 1 - 1/(amax/thr) = (amax - thr)/amax, zero at or below the threshold (dsp.py:117-121, hyrax.py:87)
        return amax > thr ? (amax - thr) / amax : 0.f;
    }
    static MGX_HD float2 scaled(float2 y, float g) { return make_float2(y.x * g, y.y * g); }
    // gain = 1 - max(g0, envelopes) = min(1 - g0, k) with k = 1 - max(envelopes) from the gain plane:
    // the frame's own hard-clip gain g0 is re-derived from the re-read frame instead of being held
    // in registers through the whole kernel
    static MGX_HD float own_gain(float2 v, float k, bool with_gain, float thr) {
        return with_gain ? fminf(k, 1.0f - gain_of(v, thr)) : 1.f;
    }
    static MGX_HD void phase_load(int tid, long long chunk, const Limiter2Args& a, float* lds) {
        const long long r0 = region_start(chunk, a);
        const bool interior = r0 >= 0 && r0 + FRAMES <= a.n;
        const float g = (float)*a.gain;
        float* gp = plane(lds, 0);
        MGX_UNROLL
        for (int j = 0; j < E / 2; ++j) {
            const int i = 2 * tid + 2 * T * j;               // two consecutive frames per lane
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
            gp[gidx(i)] = gain_of(scaled(v0, g), a.threshold);
            gp[gidx(i + 1)] = gain_of(scaled(v1, g), a.threshold);
        }
    }

    // ---- P2: own block of g0 -> registers; prefix/suffix maxima -> LDS -------------------------
    static MGX_HD void phase_planes(int tid, long long chunk, const Limiter2Args& a, Thread& th, float* lds) {
        th.base = region_start(chunk, a) + (long long)tid * E;
        th.core = tid >= a.gl && tid < T - a.gr;
        th.has_sl = tid >= a.gl && tid < T - a.gw;
        const long long left = a.n - th.base;
        th.valid = th.base < 0 ? 0 : (left >= E ? E : (left > 0 ? (int)left : 0));
        float* gp = plane(lds, 0) + tid * STRIDE;
        float* gs = plane(lds, 1) + tid * STRIDE;
        float g0[E];
        MGX_UNROLL
        for (int j = 0; j < E; ++j) g0[j] = gp[j];
        float run = 0.f;
        MGX_UNROLL
        for (int j = 0; j < E; ++j) { run = fmaxf(run, g0[j]); gp[j] = run; }
        block_max(lds)[tid] = run;
        run = 0.f;
        MGX_UNROLL
        for (int j = E - 1; j >= 0; --j) { run = fmaxf(run, g0[j]); gs[j] = run; }
    }

    // max of g0 over [c - lw, c + rw] for the 16 frames c of block tid (van Herk: suffix maximum of
    // the block holding the window's first frame, prefix maximum of the block holding its last
    // frame, whole-block maxima in between).  Needs lw + rw >= 16, so that the two ends never
    // share a block, and blocks tid-ceil(lw/16) .. tid+ceil(rw/16) inside the workgroup.
    static MGX_HD void window_max(int tid, int lw, int rw, const float* lds, float (&out)[E]) {
        const float* gp = plane(const_cast<float*>(lds), 0);
        const float* gs = plane(const_cast<float*>(lds), 1);
        const float* bm = block_max(const_cast<float*>(lds));
        const int la = lw >> 4, lb = lw & 15, ra = rw >> 4, rb = rw & 15;
        // blocks tid-la+1 .. tid+ra-1 lie strictly inside every window of this block
        float core = 0.f;
        for (int k = tid - la + 1; k <= tid + ra - 1; ++k) core = fmaxf(core, bm[k]);
        const float ml = bm[tid - la], mr = bm[tid + ra];
        const bool apart = la + ra > 0;                   // tid-la and tid+ra are different blocks
        // Rows are STRIDE = 17 floats apart, so stepping from a block into its neighbour skips exactly
        // one pad slot: the window's first frame c - lw sits at  bl + j - (j < lb),  its last frame
        // c + rw at  br + j + (j + rb >= 16).  lb, rb are uniform: two base registers per side and an
        // immediate offset per frame.
        const float* pl = gs + (tid - la) * STRIDE - lb;
        const float* pr = gp + (tid + ra) * STRIDE + rb;
        MGX_UNROLL
        for (int j = 0; j < E; ++j) {
            const bool far_l = j < lb, far_r = j + rb >= E;
            const float lft = far_l ? pl[j - 1] : pl[j];
            const float rgt = far_r ? pr[j + 1] : pr[j];
            float m = fmaxf(core, fmaxf(lft, rgt));
            // block tid-la is whole inside the window when the window starts left of it and ends
            // right of it; likewise block tid+ra
            if (far_l && (apart || far_r)) m = fmaxf(m, ml);
            if (far_r && (apart || far_l)) m = fmaxf(m, mr);
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

    // What a thread hands to the two workgroup scans that follow a phase, and what it gets back:
    // scan 0 = forward attack smoother (round 1) / backward attack smoother, right to left (round 2);
    // scan 1 = hold filter (round 1) / release filter (round 2).  `p0`, `p1` = composition of the
    // maps of all blocks before this one in the scan's direction (the device composes them with
    // wave shuffles, mgx_kernels.h; the CPU emulation with a plain loop).
    struct ScanIn {
        Affine m0, m1;
    };
    struct ScanOut {
        Affine p0, p1;
    };

    // ---- P3: sl, sh, block maps of the forward attack smoother and the hold filter -----------------
    static MGX_HD ScanIn phase_windows(int tid, const Limiter2Args& a, Thread& th, float* lds) {
        ScanIn r;
        r.m0 = affine_identity();
        r.m1 = affine_identity();
        th.inject_left = false;
        th.inject_right = false;
        th.edge_state = 0.0;
        MGX_UNROLL
        for (int j = 0; j < E; ++j) { th.sl[j] = 0.f; th.sh[j] = 0.f; }
        if (th.has_sl) {
            window_max(tid, a.hw, a.hw, lds, th.sl);
            MGX_UNROLL
            for (int j = 0; j < E; ++j)
                if (j >= th.valid) th.sl[j] = 0.f;            // windows are truncated at the array ends
            // filtfilt edge samples sl[n-7 .. n-1] for whoever holds frame n-1
            MGX_UNROLL
            for (int j = 0; j < E; ++j) {
                const long long f = th.base + j;
                if (j < th.valid && f >= a.n - 7) edge_sl(lds)[7 + (int)(f - (a.n - 7))] = th.sl[j];
            }
            th.inject_left = th.base == 0;
            th.inject_right = th.valid > 0 && th.base + th.valid == a.n;
            if (th.valid > 0) {
                const double decay = block_decay(a.pa16, a.att.alpha, th.valid);
                const double zend = (double)run_forward(a.attf, th.sl, th.valid, 0.f);
                r.m0 = Affine{decay, zend};
                if (th.inject_left) {
                    // the state entering frame 0 is filtfilt's steady-state start, whatever precedes it
                    th.edge_state = filtfilt_left_state(a.att, th.sl);
                    r.m0 = Affine{0.0, fma(decay, th.edge_state, zend)};
                }
            }
        }
        if (th.core) {
            window_max(tid, a.hw + a.hb, a.hw, lds, th.sh);
            MGX_UNROLL
            for (int j = 0; j < E; ++j)
                if (j >= th.valid) th.sh[j] = 0.f;
            if (th.valid > 0)
                r.m1 = Affine{block_decay(a.ph16, a.hold.alpha, th.valid), (double)run_forward(a.holdf, th.sh, th.valid, 0.f)};
        }
        return r;
    }

    // ---- chunk carries ----------------------------------------------------------------------------
    // slot 0 = hold, 1 = release, 2 = forward attack.  A chunk publishes the state its core frames
    // produce from a zero carry; lane `lane` of the fetching wave returns its share of
    // sum_m w[m] * published[chunk-1-m] (the caller adds the 64 shares).
    static MGX_HD void lookback_publish(long long chunk, int slot, const Limiter2Args& a, double b) {
        publish_word(a.published + (size_t)slot * a.nchunks + chunk, double_bits(b));
    }
    static MGX_HD double lookback_share(int lane, long long chunk, int slot, const Limiter2Args& a) {
        const double* w = slot == 0 ? a.w_hold : (slot == 1 ? a.w_rel : a.w_att);
        const int count = slot == 0 ? a.n_hold : (slot == 1 ? a.n_rel : a.n_att);
        double acc = 0.0;
        for (int m = lane; m < count; m += 64) {
            const long long c = chunk - 1 - m;
            if (c < 0) break;
            unsigned long long* p = a.published + (size_t)slot * a.nchunks + c;
            unsigned long long v = poll_word(p);
            int spins = 0;
            while (v == LIMITER_UNPUBLISHED && spins < MAX_SPINS) {
                backoff(spins);
                v = poll_word(p);
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

    // ---- P5: exact forward attack output and hold output; block maps of the backward attack
    //          smoother (input yf) and of the release filter (input max(sh, ho)) --------------------
    static MGX_HD ScanIn phase_exact_first(int tid, const Limiter2Args& a, Thread& th, const ScanOut& pre,
                                           double att_carry, double hold_carry, const float* lds) {
        ScanIn r;
        r.m0 = affine_identity();
        r.m1 = affine_identity();
        MGX_UNROLL
        for (int j = 0; j < E; ++j) { th.yf[j] = 0.f; th.x2[j] = 0.f; th.m[j] = 0.f; }
        if (th.has_sl) {
            double c = affine_apply(pre.p0, att_carry);
            if (th.inject_left) c = th.edge_state;
            const float zend = out_forward(a.attf, th.sl, th.valid, (float)c, th.yf);
            if (th.valid > 0) {
                const double decay = block_decay(a.pa16, a.att.alpha, th.valid);
                const double zb = (double)run_backward(a.attf, th.yf, th.valid, 0.f);
                r.m0 = Affine{decay, zb};
                if (th.inject_right) {
                    // forward state after frame n-1 -> filtfilt's backward start
                    th.edge_state = filtfilt_right_state(a.att, edge_sl(const_cast<float*>(lds)) + 7, (double)zend);
                    r.m0 = Affine{0.0, fma(decay, th.edge_state, zb)};
                }
            }
        }
        if (th.core) {
            const double c = affine_apply(pre.p1, hold_carry);
            float ho[E];
            out_forward(a.holdf, th.sh, th.valid, (float)c, ho);
            MGX_UNROLL
            for (int j = 0; j < E; ++j) {
                th.x2[j] = fmaxf(th.sh[j], ho[j]);                 // hyrax.py:73
                th.m[j] = ho[j];
            }
            if (th.valid > 0)
                r.m1 = Affine{block_decay(a.pr16, a.rel.alpha, th.valid), (double)run_forward(a.relf, th.x2, th.valid, 0.f)};
        }
        return r;
    }

    // ---- P7: exact backward attack output and release output -> gain -> LDS gain plane -------------
    static MGX_HD void phase_gain(int tid, const Limiter2Args& a, Thread& th, const ScanOut& pre, double rel_carry,
                                  float* lds) {
        if (!th.core) return;
        double cb = affine_apply(pre.p0, 0.0);
        if (th.inject_right) cb = th.edge_state;
        const double cr = affine_apply(pre.p1, rel_carry);
        float ro[E], yb[E];
        out_forward(a.relf, th.x2, th.valid, (float)cr, ro);
        MGX_UNROLL
        for (int j = 0; j < E; ++j) th.m[j] = fmaxf(th.m[j], ro[j]);           // max(ho, ro), hyrax.py:75
        out_backward(a.attf, th.yf, th.valid, (float)cb, yb);
        float* gn = gain_plane(lds) + tid * STRIDE;
        MGX_UNROLL
        for (int j = 0; j < E; ++j) gn[j] = 1.0f - fmaxf(th.m[j], yb[j]);      // hyrax.py:97 without g0 (phase_store)
    }

    // ---- P8: coalesced reload, apply gain, store ------------------------------------------------
    // with_gain = false: limiter early-out, the array only gets the final amplitude coefficient
    static MGX_HD void phase_store(int tid, long long chunk, const Limiter2Args& a, bool with_gain, const float* lds) {
        const long long r0 = region_start(chunk, a);
        const long long c0 = r0 + (long long)a.gl * E, c1 = r0 + (long long)(T - a.gr) * E;   // core frames
        const bool interior = r0 >= 0 && r0 + FRAMES <= a.n;
        const float g = (float)*a.gain, post = (float)*a.post_gain;
        const float* gn = gain_plane(const_cast<float*>(lds));
        MGX_UNROLL
        for (int j = 0; j < E / 2; ++j) {
            const int i = 2 * tid + 2 * T * j;
            const long long f = r0 + i;
            if (f < c0 || f >= c1) continue;                       // halo frames belong to the neighbours
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
};

}  // namespace mgx
