// Do LDS traffic and VALU work of ONE workgroup of 16 waves overlap on a gfx950 CU?  (round 6: the 16384-point kernels take
// as long as their LDS cycles PLUS their VALU cycles, profiles/r06_a_*.)
//
// Per iteration every wave issues R ds_read_b64 + W ds_write_b64 (its own 1024-point slice of a 128 KB array, the
// product kernels' padded layout) and V plain f32 VALU instructions.  DEP = 0: the VALU chain works on registers that have
// nothing to do with the LDS data (perfect overlap is possible inside one wave); DEP = 1: read -> arithmetic on what was
// read -> write, as a butterfly pass does (overlap only between waves).  BAR = 1: an s_barrier per iteration (all waves
// enter each iteration together, as the kernels' phases do).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/micro/lds_valu_overlap tools/micro/lds_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ int pad(int i) { return i + ((i >> 4) << 1); }

template <int R, int W, int V, int DEP, int BAR>
__global__ __launch_bounds__(1024) void k(float* out, int iters, float seed) {
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < 16384 + 2048 + 2; i += 1024) lds[i] = make_float2(seed * i, seed);
    __syncthreads();
    float2 d[16];
    float a[8];
    for (int j = 0; j < 16; ++j) d[j] = make_float2(seed * j, seed * (j + tid));
    for (int j = 0; j < 8; ++j) a[j] = seed * (j + 1);
    float2* p = lds + pad(wave * 1024 + lane);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const float2 t = p[(j & 15) * 72];            // 64 points further on, padded: 64 + 8
            if (DEP) d[j & 15] = t;
            else d[j & 15].x += t.y * 0.f;               // (consumed, but off the VALU chain's critical path)
        }
        if (DEP) {
            // V instructions on the loaded values: chains of fma across the sixteen points
#pragma unroll
            for (int v = 0; v < V / 32; ++v)
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    d[j].x = fmaf(d[j].x, 0.999f, d[(j + 1) & 15].y);
                    d[j].y = fmaf(d[j].y, 0.999f, d[(j + 5) & 15].x);
                }
        } else {
#pragma unroll
            for (int v = 0; v < V / 8; ++v)
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] = fmaf(a[j], 0.999f, a[(j + 1) & 7]);
        }
#pragma unroll
        for (int j = 0; j < W; ++j) p[(j & 15) * 72] = d[j & 15];
        if (BAR) __syncthreads();
        else asm volatile("" ::: "memory");
    }
    float acc = 0.f;
    for (int j = 0; j < 16; ++j) acc += d[j].x + d[j].y;
    for (int j = 0; j < 8; ++j) acc += a[j];
    out[blockIdx.x * 1024 + tid] = acc;
}

template <int R, int W, int V, int DEP, int BAR>
static float run(float* d_out, int iters) {
    const size_t lds_bytes = (16384 + 2048 + 2) * sizeof(float2);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k<R, W, V, DEP, BAR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        k<R, W, V, DEP, BAR><<<256, 1024, lds_bytes>>>(d_out, iters, 1e-3f);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    return best * 1e6f / iters;          // ns per iteration
}

template <int DEP, int BAR>
static void table(float* d_out, int iters) {
    printf("%s, %s: ns per iteration (16 waves on a CU; per wave and iteration R ds_read_b64, W ds_write_b64, V VALU)\n",
           DEP ? "arithmetic ON the loaded values" : "arithmetic independent of the LDS data", BAR ? "barrier per iteration" : "no barrier");
    const float r = run<16, 0, 0, DEP, BAR>(d_out, iters), w = run<0, 16, 0, DEP, BAR>(d_out, iters), rw = run<16, 16, 0, DEP, BAR>(d_out, iters);
    const float v1 = run<0, 0, 96, DEP, BAR>(d_out, iters), v2 = run<0, 0, 192, DEP, BAR>(d_out, iters), v3 = run<0, 0, 384, DEP, BAR>(d_out, iters);
    printf("  16 reads %7.1f   16 writes %7.1f   16 reads + 16 writes %7.1f   |   96 VALU %7.1f   192 VALU %7.1f   384 VALU %7.1f\n", r, w, rw, v1, v2, v3);
    const float b1 = run<16, 16, 96, DEP, BAR>(d_out, iters), b2 = run<16, 16, 192, DEP, BAR>(d_out, iters), b3 = run<16, 16, 384, DEP, BAR>(d_out, iters);
    printf("  16 reads + 16 writes + 96 VALU %7.1f (sum %7.1f, max %7.1f)   + 192 VALU %7.1f (sum %7.1f, max %7.1f)   + 384 VALU %7.1f (sum %7.1f, max %7.1f)\n",
           b1, rw + v1, rw > v1 ? rw : v1, b2, rw + v2, rw > v2 ? rw : v2, b3, rw + v3, rw > v3 ? rw : v3);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    float* d_out;
    CK(hipMalloc(&d_out, 256 * 1024 * sizeof(float)));
    table<0, 0>(d_out, iters);
    table<0, 1>(d_out, iters);
    table<1, 0>(d_out, iters);
    table<1, 1>(d_out, iters);
    return 0;
}
