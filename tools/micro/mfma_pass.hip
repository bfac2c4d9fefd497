// Step A of VERDICT round 5, item 1: do the radix-8 middle passes of Fft2<14> run faster when a share of their
// butterflies goes through the (idle) matrix pipe as exact-f32 MFMA, and does a co-resident MFMA stream cost the VALU
// stream anything?
//
// One workgroup of 1024 threads per CU, one 16384-point transform in LDS (the layout of the product kernels),
// `iters` times: forward pass 1, forward pass 2, inverse pass 2, inverse pass 1, rescale by 1/64 (the four passes
// are the identity times 64).  Modes:
//   0  VALU only (fft2.h's butterflies)                 1  MFMA only (every butterfly as a 16x16x4 tile)
//   2  shared: per wave half the butterflies each way    3  rescale only (the loop's fixed cost)
//   4  waves 0-3, 8-11 VALU, the others idle             5  waves 4-7, 12-15 MFMA, the others idle
//   6  waves 0-3, 8-11 VALU and waves 4-7, 12-15 MFMA   (every SIMD holds two waves of each kind)
//  13  VALU butterflies, the two passes of a direction fused: the exchange between them through v_permlane*_swap
// Prints us per iteration and block, the shader clock during the run (s_memtime against the 100 MHz counter), and the
// deviation of the result from the input after the identity (and between modes).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -I matchering_amd/csrc -I tools/micro -o tools/micro/mfma_pass tools/micro/mfma_pass.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "fft2_mfma.h"
#include "fft2_lanes.h"

using namespace mgx;
using F = Fft2<14>;
using FM = Fft2Mfma<14>;
using FL = Fft2Lanes<14>;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(1024) void k(float2* data, const float2* tw, int iters, long long* clocks) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2* lds = reinterpret_cast<float2*>(smem);
    float2* table = lds + F::LDS_ELEMS;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    float2* blk = data + (size_t)blockIdx.x * F::N;
    for (int i = tid; i < F::N; i += 1024) lds[F::pad(i)] = blk[i];
    F::fill_mid_table(tid, tw, table);
    FM::Matrix wf, wi;
    FM::load_matrix<false>(tid, wf);
    FM::load_matrix<true>(tid, wi);
    __syncthreads();
    const long long c0 = clock64(), r0 = wall_clock64();
    const bool valu_wave = ((wave >> 2) & 1) == 0;
    for (int it = 0; it < iters; ++it) {
        asm volatile("" : "+v"(wf.a[0]), "+v"(wf.a[1]), "+v"(wf.a[2]), "+v"(wf.a[3]));
        asm volatile("" : "+v"(wi.a[0]), "+v"(wi.a[1]), "+v"(wi.a[2]), "+v"(wi.a[3]));
        int t = tid;
        asm volatile("" : "+v"(t));
        if (MODE == 0 || (MODE == 4 && valu_wave) || (MODE == 6 && valu_wave)) {
            FM::pass_shared<1, false, 2>(t, lds, table, wf);
            asm volatile("" ::: "memory");
            FM::pass_shared<2, false, 2>(t, lds, table + F::MID_TABLE1, wf);
            asm volatile("" ::: "memory");
            FM::pass_shared<2, true, 2>(t, lds, table + F::MID_TABLE1, wi);
            asm volatile("" ::: "memory");
            FM::pass_shared<1, true, 2>(t, lds, table, wi);
        } else if (MODE == 1 || (MODE == 5 && !valu_wave) || (MODE == 6 && !valu_wave)) {
            FM::pass_shared<1, false, 0>(t, lds, table, wf);
            asm volatile("" ::: "memory");
            FM::pass_shared<2, false, 0>(t, lds, table + F::MID_TABLE1, wf);
            asm volatile("" ::: "memory");
            FM::pass_shared<2, true, 0>(t, lds, table + F::MID_TABLE1, wi);
            asm volatile("" ::: "memory");
            FM::pass_shared<1, true, 0>(t, lds, table, wi);
        } else if (MODE == 13) {           // the two middle passes as one phase, exchange through the lanes (fft2_lanes.h)
            FL::fwd_mid_fused(t, lds, table);
            asm volatile("" ::: "memory");
            FL::inv_mid_fused(t, lds, table);
        } else if (MODE == 2) {
            FM::pass_shared<1, false, 1>(t, lds, table, wf);
            asm volatile("" ::: "memory");
            FM::pass_shared<2, false, 1>(t, lds, table + F::MID_TABLE1, wf);
            asm volatile("" ::: "memory");
            FM::pass_shared<2, true, 1>(t, lds, table + F::MID_TABLE1, wi);
            asm volatile("" ::: "memory");
            FM::pass_shared<1, true, 1>(t, lds, table, wi);
        }
        asm volatile("" ::: "memory");
        if (MODE <= 3 || MODE == 13) {
            // the wave's own 1024 points, times 1/64
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float2* p = lds + F::pad(wave * 1024 + lane + 64 * e);
                float2 v = *p;
                *p = make_float2(v.x * (1.f / 64), v.y * (1.f / 64));
            }
        } else if ((valu_wave && MODE != 5) || (!valu_wave && MODE != 4)) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float2* p = lds + F::pad(wave * 1024 + lane + 64 * e);
                float2 v = *p;
                *p = make_float2(v.x * (1.f / 64), v.y * (1.f / 64));
            }
        }
        asm volatile("" ::: "memory");
    }
    const long long c1 = clock64(), r1 = wall_clock64();
    __syncthreads();
    for (int i = tid; i < F::N; i += 1024) blk[i] = lds[F::pad(i)];
    if (tid == 0 && blockIdx.x == 0) {
        clocks[0] = c1 - c0;
        clocks[1] = r1 - r0;
    }
    if (tid == 256 && blockIdx.x == 0) clocks[2] = r1 - r0;       // wave 4: an MFMA wave of modes 5 and 6
}

// The same question without the LDS: both streams on registers only.  KIND per wave as above ((wave >> 2) & 1), or every
// wave both streams (BOTH_IN_ONE).  Per iteration: VALU stream = two radix-8 butterflies with their seven twiddles
// (the arithmetic of one middle pass), matrix stream = 16 MFMAs (four tiles' worth: half a middle pass).
template <bool VALU_ON, bool MFMA_ON, bool BOTH_IN_ONE>
__global__ __launch_bounds__(1024) void kreg(float* out, int iters, float seed, long long* clocks) {
    const int tid = threadIdx.x, wave = tid >> 6;
    const bool valu_wave = BOTH_IN_ONE || ((wave >> 2) & 1) == 0, mfma_wave = BOTH_IN_ONE || !valu_wave;
    float2 v[2][8], w[7];
    for (int j = 0; j < 8; ++j) v[0][j] = make_float2(seed * (tid + j), seed * j), v[1][j] = make_float2(seed * j, seed * (tid - j));
    for (int j = 0; j < 7; ++j) w[j] = make_float2(cosf(seed * j), sinf(seed * j));
    FM::Matrix wm;
    FM::load_matrix<false>(tid, wm);
    mfma_f4 d[4];
    float b[4][4];
    for (int t = 0; t < 4; ++t) {
        d[t] = mfma_f4{0.f, 0.f, 0.f, 0.f};
        for (int m = 0; m < 4; ++m) b[t][m] = seed * (t + m + tid);
    }
    __syncthreads();
    const long long c0 = clock64(), r0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        if (MFMA_ON && mfma_wave) {
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int t = 0; t < 4; ++t) d[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wm.a[m], b[t][m], d[t], 0, 0, 0);
        }
        if (VALU_ON && valu_wave) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                dft_regs<8, false>(v[c]);
#pragma unroll
                for (int q = 1; q < 8; ++q) v[c][q] = cmul(v[c][q], w[q - 1]);
#pragma unroll
                for (int q = 0; q < 8; ++q) v[c][q] = make_float2(v[c][q].x * 0.125f, v[c][q].y * 0.125f);
            }
        }
    }
    const long long c1 = clock64(), r1 = wall_clock64();
    float acc = 0.f;
    for (int c = 0; c < 2; ++c)
        for (int j = 0; j < 8; ++j) acc += v[c][j].x + v[c][j].y;
    for (int t = 0; t < 4; ++t) acc += d[t][0] + d[t][1] + d[t][2] + d[t][3];
    out[blockIdx.x * 1024 + tid] = acc;
    if (blockIdx.x == 0 && tid == 0) clocks[0] = c1 - c0, clocks[1] = r1 - r0;      // wave 0: a VALU wave
    if (blockIdx.x == 0 && tid == 256) clocks[2] = r1 - r0;                          // wave 4: a matrix wave
}
template <bool VALU_ON, bool MFMA_ON, bool BOTH_IN_ONE>
static void run_reg(const char* name, float* d_out, int iters, long long* d_clocks) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f;
    long long clk[3] = {0, 0, 0};
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        kreg<VALU_ON, MFMA_ON, BOTH_IN_ONE><<<256, 1024>>>(d_out, iters, 1e-3f, d_clocks);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) {
            best = ms;
            CK(hipMemcpy(clk, d_clocks, sizeof clk, hipMemcpyDeviceToHost));
        }
    }
    printf("%-52s kernel %8.1f ns / iteration   VALU wave %8.1f ns   matrix wave %8.1f ns   sclk %.0f MHz\n", name, best * 1e6 / iters,
           clk[1] * 10.0 / iters, clk[2] * 10.0 / iters, clk[1] ? 100.0 * clk[0] / clk[1] : 0.0);
}

template <int MODE>
static void run(const char* name, float2* d_data, const float2* d_tw, const std::vector<float2>& input, int blocks, int iters,
                long long* d_clocks, std::vector<float2>* keep) {
    const size_t lds_bytes = ((size_t)F::LDS_ELEMS + F::MID_TABLE) * sizeof(float2);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    // identity check: one iteration
    CK(hipMemcpy(d_data, input.data(), input.size() * sizeof(float2), hipMemcpyHostToDevice));
    k<MODE><<<blocks, 1024, lds_bytes>>>(d_data, d_tw, 1, d_clocks);
    CK(hipDeviceSynchronize());
    std::vector<float2> out(input.size());
    CK(hipMemcpy(out.data(), d_data, out.size() * sizeof(float2), hipMemcpyDeviceToHost));
    double err = 0, ref = 0, dev = 0;
    for (size_t i = 0; i < out.size(); ++i) {
        const double dx = out[i].x - input[i].x, dy = out[i].y - input[i].y;
        err += dx * dx + dy * dy;
        ref += (double)input[i].x * input[i].x + (double)input[i].y * input[i].y;
        if (keep && !keep->empty()) {
            const double ex = out[i].x - (*keep)[i].x, ey = out[i].y - (*keep)[i].y;
            dev += ex * ex + ey * ey;
        }
    }
    if (keep && keep->empty()) *keep = out;
    // timing: warm-up, then three launches
    float best = 1e30f;
    long long clk[3] = {0, 0, 0};
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        k<MODE><<<blocks, 1024, lds_bytes>>>(d_data, d_tw, iters, d_clocks);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) {
            best = ms;
            CK(hipMemcpy(clk, d_clocks, sizeof clk, hipMemcpyDeviceToHost));
        }
    }
    const double rounds = (double)((blocks + 255) / 256);
    printf("%-44s %8.3f us / iteration and block (wave 0: %6.3f, wave 4: %6.3f)  sclk %.0f MHz   identity rel.rms %.2e   vs mode 0 rel.rms %.2e\n",
           name, best * 1e3 / iters / rounds, clk[1] * 0.01 / iters, clk[2] * 0.01 / iters, clk[1] ? 100.0 * clk[0] / clk[1] : 0.0,
           std::sqrt(err / ref), std::sqrt(dev / ref));
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 400;
    const int blocks = argc > 2 ? atoi(argv[2]) : 256;
    std::vector<float2> input((size_t)blocks * F::N), tw(F::N);
    srand(7);
    for (auto& v : input) v = make_float2(rand() / (float)RAND_MAX - 0.5f, rand() / (float)RAND_MAX - 0.5f);
    for (int i = 0; i < F::N; ++i) tw[i] = make_float2((float)cos(2 * M_PI * i / F::N), (float)-sin(2 * M_PI * i / F::N));
    float2 *d_data, *d_tw;
    long long* d_clocks;
    CK(hipMalloc(&d_data, input.size() * sizeof(float2)));
    CK(hipMalloc(&d_tw, tw.size() * sizeof(float2)));
    CK(hipMalloc(&d_clocks, 64));
    CK(hipMemset(d_clocks, 0, 64));
    CK(hipMemcpy(d_tw, tw.data(), tw.size() * sizeof(float2), hipMemcpyHostToDevice));
    printf("Fft2<14> middle passes (2 forward + 2 inverse radix-8 passes over 16384 points per iteration), %d blocks, %d iterations\n", blocks, iters);
    std::vector<float2> keep;
    run<0>("0 VALU only", d_data, d_tw, input, blocks, iters, d_clocks, &keep);
    run<1>("1 MFMA only", d_data, d_tw, input, blocks, iters, d_clocks, &keep);
    run<2>("2 shared (half the butterflies on each pipe)", d_data, d_tw, input, blocks, iters, d_clocks, &keep);
    run<3>("3 rescale only (fixed cost of the loop)", d_data, d_tw, input, blocks, iters, d_clocks, nullptr);
    run<4>("4 half the waves VALU, the others idle", d_data, d_tw, input, blocks, iters, d_clocks, nullptr);
    run<5>("5 half the waves MFMA, the others idle", d_data, d_tw, input, blocks, iters, d_clocks, nullptr);
    run<6>("6 half the waves VALU beside half MFMA", d_data, d_tw, input, blocks, iters, d_clocks, nullptr);
    run<13>("13 fused middle passes (lane exchange)", d_data, d_tw, input, blocks, iters, d_clocks, &keep);
    printf("registers only (no LDS): per iteration a VALU wave issues two radix-8 butterflies + twiddles, a matrix wave 16 MFMAs\n");
    float* d_out;
    CK(hipMalloc(&d_out, 256 * 1024 * sizeof(float)));
    const int ri = iters * 10;
    run_reg<true, false, false>("7 eight VALU waves, eight idle", d_out, ri, d_clocks);
    run_reg<false, true, false>("8 eight matrix waves, eight idle", d_out, ri, d_clocks);
    run_reg<true, true, false>("9 eight VALU waves beside eight matrix waves", d_out, ri, d_clocks);
    run_reg<true, false, true>("10 sixteen VALU waves", d_out, ri, d_clocks);
    run_reg<false, true, true>("11 sixteen matrix waves", d_out, ri, d_clocks);
    run_reg<true, true, true>("12 sixteen waves, both streams in every wave", d_out, ri, d_clocks);
    return 0;
}
