// libmgx_probe.so -- what a kernel can measure about the box it runs on (tools/gpu_state.py, bench.py's `gpu_state`).
//
// Measurement code, NOT part of the product: it lived in libmgx.so until round 3 and moved here so that
// include/mgx.h is the drop-in boundary and nothing else.  No reference counterpart:
// the boxes of a pool differ in the clocks they sustain and in what an instruction-cache miss costs, and a
// throughput figure means little without them.  Standalone: own stream, own events, own scratch; nothing of
// libmgx is linked or needed.
//
//   mgx_probe_clock(device, workgroups, iterations, out[4])
//       `workgroups` x 256 threads each issue `iterations` x 4 dependent FMAs; one wave reads the shader cycle
//       counter and the constant 100 MHz counter before and after.  out = {shader cycles, 100 MHz ticks,
//       shader MHz, kernel ms by HIP events}.  workgroups = 1 probes a nearly idle chip, a few thousand a chip
//       whose every SIMD issues VALU.
//   mgx_probe_memory(device, out[14])
//       out[0..2] ns per dependent load in working sets of 1 GiB (HBM), 2 MiB (L2), 8 KiB (first level);
//       out[3] GB/s of a streaming read of 1 GiB; out[4] us per launch of 200 empty kernels back to back;
//       out[5], out[6] ns per instruction of one wave walking 112 KiB of straight-line code, cold and again;
//       out[7] ns per dependent LDS read; out[8] ns per workgroup barrier (256 threads); out[9] ns per
//       returning atomic on one word; out[10..13] ns per instruction of a wave looping over 16 / 32 / 48 / 64 KiB
//       of code.  Allocates 1 GiB and frees it on every path.
//   Both return 0, or a negative number with the message in mgx_probe_last_error().
#include <hip/hip_runtime.h>

#include <algorithm>
#include <string>

static thread_local std::string g_error;
static int fail(const std::string& msg) {
    g_error = msg;
    return -2;
}
#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return fail(std::string(#expr) + ": " + hipGetErrorName(e_));            \
    } while (0)

// stream, two events and a few words of scratch, released whichever way a probe returns
struct Bench {
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    unsigned long long* ticks = nullptr;       // 64 words
    void* big = nullptr;
    int open(int device) {
        HIP_TRY(hipSetDevice(device));
        HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreate(&ev0));
        HIP_TRY(hipEventCreate(&ev1));
        HIP_TRY(hipMalloc((void**)&ticks, 512));
        HIP_TRY(hipMemsetAsync(ticks, 0, 512, stream));
        return 0;
    }
    ~Bench() {
        if (stream) hipStreamSynchronize(stream);
        if (big) hipFree(big);
        if (ticks) hipFree(ticks);
        if (ev0) hipEventDestroy(ev0);
        if (ev1) hipEventDestroy(ev1);
        if (stream) hipStreamDestroy(stream);
    }
};

// every thread runs a dependent FMA chain, wave 0 of workgroup 0 reads both counters around it
__global__ void __launch_bounds__(256) k_clock_probe(int iterations, unsigned long long* ticks, float* sink) {
    const bool reporter = blockIdx.x == 0 && threadIdx.x < 64;
    unsigned long long c0 = 0, r0 = 0;
    if (reporter) {
        c0 = __builtin_readcyclecounter();
        r0 = wall_clock64();
    }
    float a = 1.0f + threadIdx.x * 1e-7f, b = 0.999999f;
    for (int i = 0; i < iterations; ++i) {
        a = fmaf(a, b, 1e-9f);
        a = fmaf(a, b, 1e-9f);
        a = fmaf(a, b, 1e-9f);
        a = fmaf(a, b, 1e-9f);
    }
    if (reporter && threadIdx.x == 0) {
        ticks[0] = __builtin_readcyclecounter() - c0;
        ticks[1] = wall_clock64() - r0;
    }
    if (a == 123.456f) sink[0] = a;
}
extern "C" const char* mgx_probe_last_error(void) { return g_error.c_str(); }

extern "C" int mgx_probe_clock(int device, int workgroups, int iterations, double* out) {
    if (!out || workgroups < 1 || iterations < 1) return fail("bad clock probe arguments");
    Bench bench;
    if (int rc = bench.open(device)) return rc;
    HIP_TRY(hipEventRecord(bench.ev0, bench.stream));
    hipLaunchKernelGGL(k_clock_probe, dim3((unsigned)workgroups), dim3(256), 0, bench.stream, iterations, bench.ticks,
                       (float*)(bench.ticks + 4));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(bench.ev1, bench.stream));
    unsigned long long host[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(host, bench.ticks, sizeof(host), hipMemcpyDeviceToHost, bench.stream));
    HIP_TRY(hipStreamSynchronize(bench.stream));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, bench.ev0, bench.ev1));
    out[0] = (double)host[0];
    out[1] = (double)host[1];
    out[2] = host[1] ? 100.0 * (double)host[0] / (double)host[1] : 0.0;
    out[3] = ms;
    return 0;
}

// memory probe (mgx_memory_probe): one lane chases a chain of dependent loads through a table
template <bool BYPASS_L1>
__global__ void k_chase(const unsigned* table, unsigned start, int hops, unsigned long long* ticks, unsigned* sink) {
    unsigned i = start;
    const unsigned long long t0 = wall_clock64();
    for (int k = 0; k < hops; ++k) i = BYPASS_L1 ? __builtin_nontemporal_load(table + i) : table[i];
    ticks[0] = wall_clock64() - t0;
    sink[0] = i;
}
__global__ void k_chase_fill(unsigned* table, unsigned entries, unsigned stride) {
    // entry j of the cycle: j -> (j + stride) mod entries, with entries and stride coprime (one cycle through all)
    for (unsigned j = blockIdx.x * 256 + threadIdx.x; j < entries; j += gridDim.x * 256) table[j] = (j + stride) % entries;
}
__global__ __launch_bounds__(256) void k_stream_read(const float4* x, long long n, float* sink) {
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float4 v = x[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 123.456f) sink[0] = acc;
}
// latency probes for what the kernels of this library lean on besides HBM: instruction fetch (a wave walks
// 112 KiB of straight-line code, more than the instruction cache holds, cold and again), LDS (a chain of dependent
// ds_read_b32), the workgroup barrier (256 threads), and a returning atomic on one L2 word
// (.rept inside ONE asm statement: the assembler unrolls, the compiler sees a single instruction)
#define MGX_CODE_16K(lit) asm volatile(".rept 2048\n v_add_f32 %0, " lit ", %0\n .endr" : "+v"(v))
__global__ void k_ifetch(unsigned long long* ticks, float* sink, float seed) {
    float v = seed;
    const unsigned long long t0 = wall_clock64();
    // 7 x 2048 VOP2 instructions with a 32-bit literal: 8 bytes each = 112 KiB of straight-line code
    MGX_CODE_16K("0x3f800001"); MGX_CODE_16K("0x3f800002"); MGX_CODE_16K("0x3f800003"); MGX_CODE_16K("0x3f800004");
    MGX_CODE_16K("0x3f800005"); MGX_CODE_16K("0x3f800006"); MGX_CODE_16K("0x3f800007");
    ticks[0] = wall_clock64() - t0;
    sink[0] = v;
}
// the same walk over a footprint of `blocks` x 16 KiB, sixteen times: nanoseconds per instruction of the last
// eight rounds tell which footprints stay in the instruction cache
__global__ void k_iloop(int blocks, unsigned long long* ticks, float* sink, float seed) {
    float v = seed;
    unsigned long long t0 = 0;
#pragma unroll 1                        // (the compiler takes the .rept blocks for three lines each)
    for (int rep = 0; rep < 16; ++rep) {
        if (rep == 8) t0 = wall_clock64();
        MGX_CODE_16K("0x3f800001");
        if (blocks > 1) MGX_CODE_16K("0x3f800002");
        if (blocks > 2) MGX_CODE_16K("0x3f800003");
        if (blocks > 3) MGX_CODE_16K("0x3f800004");
    }
    ticks[0] = wall_clock64() - t0;
    sink[0] = v;
}
__global__ __launch_bounds__(256) void k_onchip(unsigned long long* ticks, unsigned* word, unsigned* sink) {
    __shared__ unsigned chain[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) chain[i] = (i + 331) & 1023;
    __syncthreads();
    unsigned i = 0;
    unsigned long long t0 = wall_clock64();
    for (int k = 0; k < 4096; ++k) i = chain[i];
    if (threadIdx.x == 0) ticks[2] = wall_clock64() - t0;
    __syncthreads();
    t0 = wall_clock64();
    for (int k = 0; k < 2048; ++k) __syncthreads();
    if (threadIdx.x == 0) ticks[3] = wall_clock64() - t0;
    unsigned a = i;
    if (threadIdx.x == 0) {
        t0 = wall_clock64();
        for (int k = 0; k < 2048; ++k) a = atomicAdd(word, a & 1u) + 1u;
        ticks[4] = wall_clock64() - t0;
    }
    sink[0] = i + a;
}
__global__ void k_empty() {}
extern "C" int mgx_probe_memory(int device, double* out) {
    if (!out) return fail("null argument");
    Bench bench;
    if (int rc = bench.open(device)) return rc;
    const size_t big = (size_t)1 << 30;                         // 1 GiB: four times the Infinity Cache
    HIP_TRY(hipMalloc(&bench.big, big));              // (freed by ~Bench on every return)
    unsigned* table = (unsigned*)bench.big;
    unsigned long long* ticks = bench.ticks;
    // dependent-load latency in three working sets: HBM (1 GiB, 4 MiB + 64 B steps), L2 (2 MiB), first level (8 KiB)
    const struct { size_t bytes; unsigned stride; int hops; } sets[3] = {
        {big, (4u << 20) / 4 + 16, 4096}, {(size_t)2 << 20, 4099, 8192}, {(size_t)8 << 10, 67, 8192}};
    for (int s = 0; s < 3; ++s) {
        const unsigned entries = (unsigned)(sets[s].bytes / 4);
        hipLaunchKernelGGL(k_chase_fill, dim3(2048), dim3(256), 0, bench.stream, table, entries, sets[s].stride);
        // a first walk warms the TLB (and, for the two small sets, the cache under test); the HBM walk is then
        // repeated from an entry 256 bytes further on: the same pages, lines nobody has touched
        if (s == 2) {
            hipLaunchKernelGGL(k_chase<false>, dim3(1), dim3(1), 0, bench.stream, table, 0u, sets[s].hops, ticks, (unsigned*)(ticks + 4));
            hipLaunchKernelGGL(k_chase<false>, dim3(1), dim3(1), 0, bench.stream, table, 0u, sets[s].hops, ticks, (unsigned*)(ticks + 4));
        } else {
            hipLaunchKernelGGL(k_chase<true>, dim3(1), dim3(1), 0, bench.stream, table, 0u, sets[s].hops, ticks, (unsigned*)(ticks + 4));
            hipLaunchKernelGGL(k_chase<true>, dim3(1), dim3(1), 0, bench.stream, table, s == 0 ? 64u : 0u, sets[s].hops, ticks,
                               (unsigned*)(ticks + 4));
        }
        unsigned long long t = 0;
        HIP_TRY(hipMemcpyAsync(&t, ticks, 8, hipMemcpyDeviceToHost, bench.stream));
        HIP_TRY(hipStreamSynchronize(bench.stream));
        out[s] = (double)t * 10.0 / sets[s].hops;               // ns per hop (100 MHz ticks)
    }
    // streaming read of the 1 GiB, GB/s (best of three)
    double best = 0.0;
    for (int rep = 0; rep < 4; ++rep) {
        HIP_TRY(hipEventRecord(bench.ev0, bench.stream));
        hipLaunchKernelGGL(k_stream_read, dim3(4096), dim3(256), 0, bench.stream, (const float4*)table, (long long)(big / 16),
                           (float*)(ticks + 4));
        HIP_TRY(hipEventRecord(bench.ev1, bench.stream));
        HIP_TRY(hipEventSynchronize(bench.ev1));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, bench.ev0, bench.ev1));
        if (rep > 0) best = std::max(best, (double)big / (ms * 1e-3) / 1e9);
    }
    out[3] = best;
    // 200 empty kernels back to back: microseconds per launch as the device sees them (events on the stream)
    HIP_TRY(hipEventRecord(bench.ev0, bench.stream));
    for (int k = 0; k < 200; ++k) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, bench.stream);
    HIP_TRY(hipEventRecord(bench.ev1, bench.stream));
    HIP_TRY(hipEventSynchronize(bench.ev1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, bench.ev0, bench.ev1));
    out[4] = ms * 1e3 / 200.0;
    // instruction fetch, LDS, barrier, atomic (nanoseconds per instruction / hop / barrier / atomic)
    HIP_TRY(hipMemsetAsync(ticks, 0, 64, bench.stream));
    hipLaunchKernelGGL(k_ifetch, dim3(1), dim3(64), 0, bench.stream, ticks, (float*)(ticks + 6), 1.0f);        // cold
    hipLaunchKernelGGL(k_ifetch, dim3(1), dim3(64), 0, bench.stream, ticks + 1, (float*)(ticks + 6), 1.0f);    // again: from the L2
    hipLaunchKernelGGL(k_onchip, dim3(1), dim3(256), 0, bench.stream, ticks, (unsigned*)(ticks + 7), (unsigned*)(ticks + 6));
    unsigned long long t5[5] = {0, 0, 0, 0, 0};
    HIP_TRY(hipMemcpyAsync(t5, ticks, sizeof(t5), hipMemcpyDeviceToHost, bench.stream));
    HIP_TRY(hipStreamSynchronize(bench.stream));
    out[5] = (double)t5[0] * 10.0 / 14336.0;                    // cold walk of 112 KiB of code
    out[6] = (double)t5[1] * 10.0 / 14336.0;                    // second walk
    out[7] = (double)t5[2] * 10.0 / 4096.0;                     // dependent LDS read
    out[8] = (double)t5[3] * 10.0 / 2048.0;                     // workgroup barrier, 256 threads
    out[9] = (double)t5[4] * 10.0 / 2048.0;                     // returning atomic on one word
    // footprints of 16, 32, 48 and 64 KiB of code walked over and over
    for (int b = 1; b <= 4; ++b) {
        hipLaunchKernelGGL(k_iloop, dim3(1), dim3(64), 0, bench.stream, b, ticks, (float*)(ticks + 6), 1.0f);
        unsigned long long t = 0;
        HIP_TRY(hipMemcpyAsync(&t, ticks, 8, hipMemcpyDeviceToHost, bench.stream));
        HIP_TRY(hipStreamSynchronize(bench.stream));
        out[9 + b] = (double)t * 10.0 / (8.0 * 2048.0 * b);
    }
    return 0;
}
