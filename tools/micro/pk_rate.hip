// Issue rate of packed float32 VALU instructions on gfx950: the same complex butterfly chain written
// with scalar v_add/v_mul/v_fmac and with v_pk_add/v_pk_mul/v_pk_fma (op_sel swizzles in the
// instruction).  Build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o pk_rate pk_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2f cmul_pk(v2f a, v2f w) {
    v2f t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
}
__device__ __forceinline__ v2f sub_mi_pk(v2f a, v2f b) {   // -i (a - b) = (a.y - b.y, b.x - a.x)
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0] neg_lo:[0,1] neg_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
struct S2 { float x, y; };
__device__ __forceinline__ S2 cmul_s(S2 a, S2 w) { return {a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x}; }

template <bool PK>
__global__ __launch_bounds__(256) void k(float* out, int iters, float wx, float wy) {
    extern __shared__ float occupancy_limiter[];     // dynamic LDS sets the workgroups per CU
    if (iters < 0) occupancy_limiter[threadIdx.x] = 0.f;
    const int t = threadIdx.x + blockIdx.x * 256;
    if (PK) {
        v2f v[8], w = {wx, wy};
        for (int j = 0; j < 8; ++j) v[j] = v2f{(float)(t + j) * 1e-3f, (float)j};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v2f a = v[j], b = v[j + 4];
                v[j] = a + b;
                v[j + 4] = cmul_pk(sub_mi_pk(a, b), w);
            }
        }
        float s = 0;
        for (int j = 0; j < 8; ++j) s += v[j].x + v[j].y;
        out[t] = s;
    } else {
        S2 v[8], w = {wx, wy};
        for (int j = 0; j < 8; ++j) v[j] = S2{(float)(t + j) * 1e-3f, (float)j};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                S2 a = v[j], b = v[j + 4];
                v[j] = S2{a.x + b.x, a.y + b.y};
                S2 d = {a.y - b.y, b.x - a.x};
                v[j + 4] = cmul_s(d, w);
            }
        }
        float s = 0;
        for (int j = 0; j < 8; ++j) s += v[j].x + v[j].y;
        out[t] = s;
    }
}

__global__ void check(float* o, float ax, float ay, float bx, float by) {
    v2f a = {ax, ay}, b = {bx, by};
    v2f m = cmul_pk(a, b), d = sub_mi_pk(a, b);
    o[0] = m.x; o[1] = m.y; o[2] = d.x; o[3] = d.y;
}

int main() {
    {
        float* o; float h[4];
        hipMalloc(&o, 16);
        check<<<1, 1>>>(o, 1.5f, -2.f, 0.25f, 3.f);
        hipMemcpy(h, o, 16, hipMemcpyDeviceToHost);
        // (1.5-2i)(0.25+3i) = 6.375 + 4i ; -i((1.5-2i)-(0.25+3i)) = -i(1.25-5i) = -5 - 1.25i
        printf("cmul_pk = (%g, %g) want (6.375, 4);  sub_mi_pk = (%g, %g) want (-5, -1.25)\n", h[0], h[1], h[2], h[3]);
    }
    float* out;
    const int iters = 20000;
    hipMalloc(&out, 256 * 8 * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    // waves per SIMD = workgroups per CU (256 threads = one wave on each of the 4 SIMDs)
    for (int per_cu : {1, 2, 4, 8}) {
        const int wgs = 256 * per_cu;
        const size_t lds = (size_t)160 * 1024 / per_cu - 2048;
        hipFuncSetAttribute((const void*)k<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute((const void*)k<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        for (int pk = 0; pk < 2; ++pk)
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (pk) k<true><<<wgs, 256, lds>>>(out, iters, 0.6f, 0.8f);
                else k<false><<<wgs, 256, lds>>>(out, iters, 0.6f, 0.8f);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                const double flop = (double)wgs * 256 * iters * 40;
                if (rep) printf("%d waves/SIMD %s: %.3f ms  %.1f TFLOP/s\n", per_cu, pk ? "packed" : "scalar", ms, flop / ms * 1e-9);
            }
    }
    return 0;
}
