// fft_size 8, 16 and 32: the sizes the reference still runs (defaults.py:110-112 asks only for a power of two
// above 1; at 2 and 4 its own cubic interpolation fails, match_frequencies.py:45-58) and the LDS transforms of
// fft2.h do not reach (a transform there is at least 64 points wide: one wavefront of rows).  Nobody masters with
// a 32-tap matching filter, so these kernels are written to be obviously right, not fast: a whole segment's
// transform fits one thread's registers, and a 32-tap filter is applied in the time domain.
//
//   k_analyze_small<LOG2F>  match_levels.py:134-161 + match_frequencies.py:30-42, outputs as k_analyze
//   k_fir_taps_direct       match_frequencies.py:98-99 (irfft, ifftshift, Hann) as a direct cosine sum (mgx_kernels.h)
//   k_conv_direct           match_frequencies.py:104-119 (fftconvolve "same" on mid and side, ms_to_lr)
#pragma once

#include "analysis2_kernel.h"
#include "fir_plan.h"

#if defined(__clang__)
#pragma clang fp contract(fast)        // the float32 butterflies only, as in fft2.h
#endif

namespace mgx {

// One thread = one segment at a time: z = mid + j side, F-point transform in registers, |M_k| and |S_k| from a bin
// and its mirror exactly as analysis2_kernel.h forms them.  Workgroup (piece, chunk) layout, leftover frames and the
// per-workgroup outputs are those of k_analyze, so everything downstream is shared.
template <int LOG2F>
__global__ __launch_bounds__(256) void k_analyze_small(AnalysisArgs a0, AnalysisArgs a1, int nwg0) {
    constexpr int F = 1 << LOG2F, HALF = F / 2;
    __shared__ double dscratch[8];
    __shared__ float fscratch[8];
    __shared__ float spec[4][2][HALF + 1];
    const bool second = (int)blockIdx.x >= nwg0;                 // uniform
    const AnalysisArgs& a = second ? a1 : a0;
    const int tid = threadIdx.x, wg = second ? blockIdx.x - nwg0 : blockIdx.x;
    const int d = wg / a.chunks_per_piece, ch = wg % a.chunks_per_piece;
    const int s0 = (int)((long long)ch * a.segs_per_piece / a.chunks_per_piece);
    const int s1 = (int)((long long)(ch + 1) * a.segs_per_piece / a.chunks_per_piece);
    double sumsq = 0.0;
    float peak = 0.f, acc_mid[HALF + 1], acc_side[HALF + 1];
#pragma unroll
    for (int k = 0; k <= HALF; ++k) acc_mid[k] = acc_side[k] = 0.f;
    for (int s = s0 + tid; s < s1; s += 256) {
        const float2* x = a.x + (long long)d * a.piece + (long long)s * F;
        float2 v[F];
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < F; ++j) {
            const float2 lr = x[j];
            const float m = (lr.x + lr.y) * 0.5f;                // dsp.py:59-60
            v[j] = make_float2(m, m - lr.y);                     // dsp.py:62
            ss = fmaf(m, m, ss);
            peak = fmaxf(peak, fmaxf(fabsf(lr.x), fabsf(lr.y)));
        }
        sumsq += (double)ss;
        dft_regs<F, false>(v);                                   // X[q] sits at v[bitrev(q)]
#pragma unroll
        for (int k = 0; k <= HALF; ++k) {
            const float2 z = v[bitrev(k, LOG2F)], zm = v[bitrev((F - k) & (F - 1), LOG2F)];
            const float mx = z.x + zm.x, my = z.y - zm.y;        // M = (Z + conj Zm) / 2
            const float sx = z.x - zm.x, sy = z.y + zm.y;        // S = (Z - conj Zm) / 2j
            acc_mid[k] += 0.5f * fast_sqrt(fmaf(mx, mx, my * my));
            acc_side[k] += 0.5f * fast_sqrt(fmaf(sx, sx, sy * sy));
        }
    }
    // frames outside whole segments: RMS (the piece's leftover) and peak (the ignored tail of the track)
    auto loose = [&](long long begin, long long end, bool count_rms) {
        for (long long f = begin + tid; f < end; f += 256) {
            const float2 lr = a.x[f];
            const float m = (lr.x + lr.y) * 0.5f;
            if (count_rms) sumsq += (double)(m * m);
            peak = fmaxf(peak, fmaxf(fabsf(lr.x), fabsf(lr.y)));
        }
    };
    if (ch == a.chunks_per_piece - 1) {
        loose((long long)d * a.piece + (long long)a.segs_per_piece * F, (long long)(d + 1) * a.piece, true);
        if (d == a.divisions - 1) loose((long long)a.divisions * a.piece, a.n, false);
    }
    // spectrum sums of the workgroup: lanes by shuffles, the four waves through LDS
    const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int k = 0; k <= HALF; ++k) {
        float m = acc_mid[k], s = acc_side[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            m += __shfl_xor(m, o, 64);
            s += __shfl_xor(s, o, 64);
        }
        if (lane == 0) {
            spec[wave][0][k] = m;
            spec[wave][1][k] = s;
        }
    }
    __syncthreads();
    if (tid < 2 * (HALF + 1)) {
        const int c = tid / (HALF + 1), k = tid % (HALF + 1);
        a.wg_spec[(size_t)wg * 2 * (HALF + 1) + (size_t)c * (HALF + 1) + k] =
            (spec[0][c][k] + spec[1][c][k]) + (spec[2][c][k] + spec[3][c][k]);
    }
    const double ss = block_sum<256>(sumsq, dscratch);
    const float pk = block_max<256>(peak, fscratch);
    if (tid == 0) {
        a.wg_sumsq[wg] = ss;
        a.wg_peak[wg] = pk;
    }
}

// y[n] = sum_k h[k] x[n + (F-1)/2 - k] on mid and side (scipy.signal.fftconvolve(..., "same"): the output is the
// full convolution from index (F-1)//2 on), frames outside the track count as zeros; L = mid + side, R = mid - side
// (dsp.py:67-68).  A workgroup of 256 threads takes CONV_DIRECT_TILE output frames; `peaks[workgroup]` = its
// max(|L|,|R|), what the level correction reads as the convolution's block peaks.
constexpr int CONV_DIRECT_TILE = 1024, CONV_DIRECT_MAX_TAPS = 32;
__global__ __launch_bounds__(256) void k_conv_direct(const float2* x, long long n, const float* taps /* [2][F] */, int f,
                                                     const double* gain_ptr, double gain, float2* y, float* ymid,
                                                     float* peaks) {
    __shared__ float mid[CONV_DIRECT_TILE + CONV_DIRECT_MAX_TAPS], side[CONV_DIRECT_TILE + CONV_DIRECT_MAX_TAPS];
    __shared__ float h[2][CONV_DIRECT_MAX_TAPS];
    __shared__ float fscratch[8];
    const int tid = threadIdx.x;
    const long long o0 = (long long)blockIdx.x * CONV_DIRECT_TILE;
    const float g = (float)((gain_ptr ? *gain_ptr : 1.0) * gain);       // stages.py:80-88, folded into the filter
    if (tid < 2 * f) h[tid / f][tid % f] = taps[tid] * g;
    const long long first = o0 - f / 2;                                  // input frame of window position 0
    for (int i = tid; i < CONV_DIRECT_TILE + f; i += 256) {
        const long long q = first + i;
        float2 lr = make_float2(0.f, 0.f);
        if (q >= 0 && q < n) lr = x[q];
        const float m = (lr.x + lr.y) * 0.5f;
        mid[i] = m;
        side[i] = m - lr.y;
    }
    __syncthreads();
    float pk = 0.f;
    for (int r = tid; r < CONV_DIRECT_TILE; r += 256) {
        const long long out = o0 + r;
        if (out >= n) break;
        // x[out + F/2 - 1 - k] is window position r + F - 1 - k
        float am = 0.f, as = 0.f;
        for (int k = 0; k < f; ++k) {
            am = fmaf(h[0][k], mid[r + f - 1 - k], am);
            as = fmaf(h[1][k], side[r + f - 1 - k], as);
        }
        const float2 lr = make_float2(am + as, am - as);
        y[out] = lr;
        if (ymid) ymid[out] = am;
        pk = fmaxf(pk, fmaxf(fabsf(lr.x), fabsf(lr.y)));
    }
    const float bp = block_max<256>(pk, fscratch);
    if (tid == 0 && peaks) peaks[blockIdx.x] = bp;
}

}  // namespace mgx

#if defined(__clang__)
#pragma clang fp contract(off)
#endif
