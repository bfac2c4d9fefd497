// Attainable HBM read rate for the access pattern of k_analyze: persistent 256-thread workgroups,
// each reads 32 KB segments as 16 strided dwordx2 per thread (frames t + j*256), against the same
// bytes read as linear dwordx4, with and without workgroup barriers between segments.
// Build: hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -o hbm_stream hbm_stream.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2 __attribute__((ext_vector_type(2)));
typedef float v4 __attribute__((ext_vector_type(4)));

template <int MODE, int BARRIERS>
__global__ __launch_bounds__(256) void k(const float* x, long long nseg, float* out) {
    float acc = 0.f;
    for (long long s = blockIdx.x; s < nseg; s += gridDim.x) {
        const float* seg = x + s * 8192;          // 4096 frames * 2 floats
        if (MODE == 0) {
            v2 v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = reinterpret_cast<const v2*>(seg)[threadIdx.x + j * 256];
#pragma unroll
            for (int j = 0; j < 16; ++j) acc += v[j].x * v[j].y;
        } else {
            v4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = reinterpret_cast<const v4*>(seg)[threadIdx.x + j * 256];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += v[j].x * v[j].y + v[j].z * v[j].w;
        }
#pragma unroll
        for (int b = 0; b < BARRIERS; ++b) __syncthreads();
    }
    if (acc == 1.2345f) out[0] = acc;
}

template <int MODE, int BARRIERS>
void run(const char* name, const float* x, long long nseg, float* out, int grid) {
    for (int rep = 0; rep < 3; ++rep) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        k<MODE, BARRIERS><<<grid, 256>>>(x, nseg, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("%-34s grid %5d: %.3f ms  %.2f TB/s\n", name, grid, ms, nseg * 32768.0 / ms * 1e-9);
    }
}

int main() {
    const long long nseg = 5168 * 8;            // eight 8-minute tracks: 1.35 GB, larger than the Infinity Cache
    float* x; float* out;
    hipMalloc(&x, nseg * 32768); hipMemset(x, 0, nseg * 32768); hipMalloc(&out, 4);
    for (int grid : {256 * 2, 256 * 4, 256 * 8}) {
        run<0, 0>("dwordx2 strided, no barrier", x, nseg, out, grid);
        run<0, 4>("dwordx2 strided, 4 barriers", x, nseg, out, grid);
        run<1, 0>("dwordx4 linear, no barrier", x, nseg, out, grid);
    }
    // one 8-minute track only (170 MB: fits the Infinity Cache once warm)
    run<0, 0>("170 MB, dwordx2 strided", x, 5168, out, 1024);
    run<1, 0>("170 MB, dwordx4 linear", x, 5168, out, 1024);
    return 0;
}
