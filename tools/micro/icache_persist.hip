// Three questions about the instruction cache and kernels of two streams (round 5):
//   1. Does a kernel's code stay in the instruction cache from one launch to the next (is the cache kept over a
//      dispatch)?  walk<K KiB> twice in a row; then with 1 GB streamed through the L2s in between.
//   2. Do kernels of two streams run side by side?  `hold` (129 workgroups, each busy for ~30 us) on stream A,
//      `walk` on stream B: start and end stamps of both (100 MHz wall clock).
//   3. Does a "shadow" launch of a kernel on a second stream, while a small kernel holds the main stream, leave its
//      code warm for the real launch that follows on the main stream?
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/icache_persist tools/micro/icache_persist.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define CODE_16K(lit) asm volatile(".rept 2048\n v_add_f32 %0, " lit ", %0\n .endr" : "+v"(v))

// 32 KiB of straight-line code, one wave per workgroup; ticks[wg] = {start, end}
__global__ void k_walk32(long long* stamps, float* sink, float seed) {
    float v = seed;
    const long long t0 = wall_clock64();
    CODE_16K("0x3f800001");
    CODE_16K("0x3f800002");
    const long long t1 = wall_clock64();
    if (threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = t0;
        stamps[2 * blockIdx.x + 1] = t1;
    }
    if (v == 123.456f) sink[0] = v;
}
// other code of the same size (evicts nothing of walk32 unless the cache is smaller than both)
__global__ void k_other32(long long* stamps, float* sink, float seed) {
    float v = seed;
    const long long t0 = wall_clock64();
    CODE_16K("0x3f800011");
    CODE_16K("0x3f800012");
    const long long t1 = wall_clock64();
    if (threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = t0;
        stamps[2 * blockIdx.x + 1] = t1;
    }
    if (v == 123.456f) sink[0] = v;
}
__global__ __launch_bounds__(256) void k_stream(const float4* x, long long n, float* sink) {
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float4 q = x[i];
        acc += q.x + q.y + q.z + q.w;
    }
    if (acc == 123.456f) sink[0] = acc;
}
// a small resident grid that holds its stream for `ticks` of the wall clock (the level-correction tail's shape)
__global__ __launch_bounds__(256) void k_hold(long long ticks, long long* stamps) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = t0;
        stamps[2 * blockIdx.x + 1] = wall_clock64();
    }
}

static double median_ns(const std::vector<long long>& s, int wgs, double ns_per_tick, int instructions) {
    std::vector<double> d(wgs);
    for (int i = 0; i < wgs; ++i) d[i] = (double)(s[2 * i + 1] - s[2 * i]) * ns_per_tick / instructions;
    std::sort(d.begin(), d.end());
    return d[wgs / 2];
}
static void span(const std::vector<long long>& s, int wgs, long long& lo, long long& hi) {
    lo = s[0]; hi = s[1];
    for (int i = 0; i < wgs; ++i) { lo = std::min(lo, s[2 * i]); hi = std::max(hi, s[2 * i + 1]); }
}

int main() {
    int wc_khz = 0;
    hipDeviceGetAttribute(&wc_khz, hipDeviceAttributeWallClockRate, 0);
    const double ns = 1e6 / wc_khz;
    const int G = 1024, I = 4096;
    long long *sa, *sb, *sh;
    float* sink;
    hipMalloc(&sa, G * 16); hipMalloc(&sb, G * 16); hipMalloc(&sh, G * 16); hipMalloc(&sink, 64);
    const long long nbig = (1ll << 30) / 16;
    float4* big; hipMalloc(&big, nbig * 16); hipMemset(big, 0, nbig * 16);
    hipStream_t A, B;
    hipStreamCreateWithFlags(&A, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&B, hipStreamNonBlocking);
    std::vector<long long> ha(2 * G), hb(2 * G), hh(2 * G);
    auto fetch = [&](long long* d, std::vector<long long>& h) { hipMemcpy(h.data(), d, G * 16, hipMemcpyDeviceToHost); };

    printf("wall clock %d kHz; walk32 = 4096 instructions (32 KiB) per wave, %d workgroups of one wave; ns per instruction, median over workgroups\n", wc_khz, G);
    // 1a. twice in a row
    for (int rep = 0; rep < 2; ++rep) {
        k_stream<<<2048, 256, 0, A>>>(big, nbig, sink);                  // push everything out of the L2s first
        k_other32<<<G, 64, 0, A>>>(sb, sink, 1.f);                       // ... and other code into the instruction caches
        k_walk32<<<G, 64, 0, A>>>(sa, sink, 1.f);
        k_walk32<<<G, 64, 0, A>>>(sb, sink, 1.f);
        hipStreamSynchronize(A);
        fetch(sa, ha); fetch(sb, hb);
        printf("1a. stream 1 GB, other32, then walk32 twice in a row:      first %.2f   second %.2f\n", median_ns(ha, G, ns, I), median_ns(hb, G, ns, I));
    }
    // 1b. the L2s flushed in between, the instruction caches untouched
    for (int rep = 0; rep < 2; ++rep) {
        k_walk32<<<G, 64, 0, A>>>(sa, sink, 1.f);
        k_stream<<<2048, 256, 0, A>>>(big, nbig, sink);
        k_walk32<<<G, 64, 0, A>>>(sb, sink, 1.f);
        hipStreamSynchronize(A);
        fetch(sb, hb);
        printf("1b. walk32, stream 1 GB, walk32:                           second %.2f\n", median_ns(hb, G, ns, I));
    }
    // 1c. other code of the same size in between (64 KiB together)
    for (int rep = 0; rep < 2; ++rep) {
        k_walk32<<<G, 64, 0, A>>>(sa, sink, 1.f);
        k_other32<<<G, 64, 0, A>>>(sa, sink, 1.f);
        k_walk32<<<G, 64, 0, A>>>(sb, sink, 1.f);
        hipStreamSynchronize(A);
        fetch(sb, hb);
        printf("1c. walk32, other32, walk32:                               second %.2f\n", median_ns(hb, G, ns, I));
    }
    // 1d. with a host synchronisation between the launches (what a fence at the end of a dispatch does)
    k_walk32<<<G, 64, 0, A>>>(sa, sink, 1.f);
    hipStreamSynchronize(A);
    k_walk32<<<G, 64, 0, A>>>(sb, sink, 1.f);
    hipStreamSynchronize(A);
    fetch(sb, hb);
    printf("1d. walk32, host sync, walk32:                                 second %.2f\n", median_ns(hb, G, ns, I));

    // 2. two streams side by side: hold (129 workgroups x 30 us) on A, walk32 on B behind an event of A
    hipEvent_t ev;
    hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    for (int rep = 0; rep < 3; ++rep) {
        k_stream<<<2048, 256, 0, A>>>(big, nbig / 4, sink);
        hipEventRecord(ev, A);
        k_hold<<<129, 256, 0, A>>>((long long)(30e3 / ns), sh);
        hipStreamWaitEvent(B, ev, 0);
        k_walk32<<<G, 64, 0, B>>>(sa, sink, 1.f);
        hipStreamSynchronize(A); hipStreamSynchronize(B);
        fetch(sh, hh); fetch(sa, ha);
        long long h0, h1, w0, w1;
        span(hh, 129, h0, h1); span(ha, G, w0, w1);
        printf("2.  hold on A: 0 .. %.1f us;  walk32 on B (behind an event): %.1f .. %.1f us  (overlap %s)\n", (h1 - h0) * ns * 1e-3,
               (w0 - h0) * ns * 1e-3, (w1 - h0) * ns * 1e-3, w0 < h1 ? "YES" : "no");
    }
    // 3. shadow launch: A = stream (evicts the L2s), other32 (evicts the instruction caches), hold 30 us, walk32 (timed);
    //    B = walk32 behind the event recorded before the hold.  Against the same without B.
    for (int shadow = 0; shadow < 2; ++shadow)
        for (int rep = 0; rep < 3; ++rep) {
            k_stream<<<2048, 256, 0, A>>>(big, nbig, sink);
            k_other32<<<G, 64, 0, A>>>(sb, sink, 1.f);
            hipEventRecord(ev, A);
            k_hold<<<129, 256, 0, A>>>((long long)(30e3 / ns), sh);
            if (shadow) {
                hipStreamWaitEvent(B, ev, 0);
                k_walk32<<<G / 4, 64, 0, B>>>(sb, sink, 1.f);          // one wave per compute unit
            }
            k_walk32<<<G, 64, 0, A>>>(sa, sink, 1.f);
            hipStreamSynchronize(A); hipStreamSynchronize(B);
            fetch(sa, ha); fetch(sh, hh);
            long long h0, h1, w0, w1;
            span(hh, 129, h0, h1); span(ha, G, w0, w1);
            printf("3.  %s: hold 0 .. %.1f us, the timed walk32 %.1f .. %.1f us, %.2f ns per instruction\n",
                   shadow ? "with a shadow walk32 on B " : "without                   ", (h1 - h0) * ns * 1e-3, (w0 - h0) * ns * 1e-3,
                   (w1 - h0) * ns * 1e-3, median_ns(ha, G, ns, I));
        }
    return 0;
}
