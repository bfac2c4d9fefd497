// How fast does one CU move a 96 KB tile between L2/HBM and registers, as a function of the width of
// the per-lane access?  Two 256-thread workgroups per CU (as k_conv), every wave moves 24 KB per
// round: 48 x dwordx2, 24 x dwordx4 or 96 x dword.  Reports bytes per cycle per CU (s_memtime).
// Build: hipcc --offload-arch=gfx950 -O3 -o vmem_issue vmem_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2 __attribute__((ext_vector_type(2)));
typedef float v4 __attribute__((ext_vector_type(4)));

template <int W, bool STORE>
__global__ __launch_bounds__(256, 2) void k(float* buf, size_t tile_stride_floats, int rounds, long long* cyc, float* sink) {
    extern __shared__ float occupancy_limiter[];      // dynamic LDS sets the workgroups per CU
    if (rounds < 0) occupancy_limiter[threadIdx.x] = 0.f;
    // tile of this workgroup: 24576 floats (96 KB); wave w, lane l
    float* base = buf + (size_t)blockIdx.x * tile_stride_floats;
    const int t = threadIdx.x;
    float acc = 0.f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < rounds; ++r) {
        float* p = base + (size_t)(r & 3) * 0;       // same tile every round (L2-resident after round 0)
        if (W == 2) {
            v2 v[48];
#pragma unroll
            for (int j = 0; j < 48; ++j) {
                if (STORE) { v[j] = v2{(float)j, acc}; reinterpret_cast<v2*>(p)[t + j * 256] = v[j]; }
                else v[j] = reinterpret_cast<const v2*>(p)[t + j * 256];
            }
            if (!STORE) {
#pragma unroll
                for (int j = 0; j < 48; ++j) acc += v[j].x + v[j].y;
            }
        } else if (W == 4) {
            v4 v[24];
#pragma unroll
            for (int j = 0; j < 24; ++j) {
                if (STORE) { v[j] = v4{(float)j, acc, 1.f, 2.f}; reinterpret_cast<v4*>(p)[t + j * 256] = v[j]; }
                else v[j] = reinterpret_cast<const v4*>(p)[t + j * 256];
            }
            if (!STORE) {
#pragma unroll
                for (int j = 0; j < 24; ++j) acc += v[j].x + v[j].y + v[j].z + v[j].w;
            }
        } else {
            float v[96];
#pragma unroll
            for (int j = 0; j < 96; ++j) {
                if (STORE) { v[j] = (float)j + acc; p[t + j * 256] = v[j]; }
                else v[j] = p[t + j * 256];
            }
            if (!STORE) {
#pragma unroll
                for (int j = 0; j < 96; ++j) acc += v[j];
            }
        }
        __syncthreads();
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (t == 0) cyc[blockIdx.x] = t1 - t0;
    if (acc == 123.456f) sink[0] = acc;
}

template <int W, bool STORE>
void run(const char* name, float* buf, size_t stride, long long* cyc, float* sink, int wgs, int per_cu) {
    const size_t lds = (size_t)160 * 1024 / per_cu - 1024;
    hipFuncSetAttribute((const void*)k<W, STORE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int rounds = 64;
    for (int rep = 0; rep < 2; ++rep) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        k<W, STORE><<<wgs, 256, lds>>>(buf, stride, rounds, cyc, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> h(wgs);
        hipMemcpy(h.data(), cyc, wgs * 8, hipMemcpyDeviceToHost);
        double mean = 0; for (auto c : h) mean += c; mean /= wgs;
        const double bytes = 98304.0 * rounds;
        printf("%-26s %d WG/CU rep %d: %.3f ms, %.0f cycles/round/WG, %.1f B/cycle/WG, chip %.2f TB/s\n", name, per_cu, rep, ms,
               mean / rounds, bytes / mean, bytes * wgs / ms * 1e-9);
    }
}

int main() {
    const int wgs = 1024;
    float* buf; long long* cyc; float* sink;
    const size_t resident = 24576, streaming = 24576;   // floats between tiles (tiles are private either way)
    hipMalloc(&buf, (size_t)wgs * 24576 * 4);
    hipMemset(buf, 0, (size_t)wgs * 24576 * 4);
    hipMalloc(&cyc, wgs * 8); hipMalloc(&sink, 4);
    (void)streaming;
    for (int per_cu : {1, 2, 4}) {
        const int n = 256 * per_cu;
        run<2, false>("load dwordx2 (48/wave)", buf, resident, cyc, sink, n, per_cu);
        run<4, false>("load dwordx4 (24/wave)", buf, resident, cyc, sink, n, per_cu);
        run<2, true>("store dwordx2 (48/wave)", buf, resident, cyc, sink, n, per_cu);
        run<4, true>("store dwordx4 (24/wave)", buf, resident, cyc, sink, n, per_cu);
    }
    return 0;
}
