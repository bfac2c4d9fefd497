// What does the raw-buffer range check of gfx950 look at?  Loads and stores through a descriptor of
// `n` bytes with the offset split between the VGPR (voffset) and the SGPR (soffset) operand.
// Build: hipcc --offload-arch=gfx950 -O3 -o buffer_range buffer_range.hip
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void k(float* buf, int nbytes, unsigned voff, unsigned soff, float* out, int do_store) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(buf, 0, nbytes, 0x00020000);
    if (do_store) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(-5.f), r, voff, soff, 0);
    else out[0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}

int main() {
    const int n = 1024;                       // floats in the view; the allocation is four times larger
    float* buf; float* out;
    hipMalloc(&buf, 4 * n * 4); hipMalloc(&out, 4);
    float h[4 * n];
    for (int i = 0; i < 4 * n; ++i) h[i] = (float)i;
    struct Case { const char* what; unsigned voff, soff; } cases[] = {
        {"in range, all in voffset", 400, 0},
        {"in range, split", 200, 200},
        {"past the end through soffset only (voffset in range)", (unsigned)(n * 4 - 8), 64},
        {"past the end through voffset only", (unsigned)(n * 4 + 56), 0},
        {"voffset wrapped negative (-800), soffset 1200: sum in range", (unsigned)-800, 1200},
        {"voffset wrapped negative (-800), soffset 0", (unsigned)-800, 0},
    };
    for (auto& c : cases) {
        hipMemcpy(buf, h, sizeof(h), hipMemcpyHostToDevice);
        k<<<1, 1>>>(buf, n * 4, c.voff, c.soff, out, 0);
        float got; hipMemcpy(&got, out, 4, hipMemcpyDeviceToHost);
        k<<<1, 1>>>(buf, n * 4, c.voff, c.soff, out, 1);
        float back[4 * n]; hipMemcpy(back, buf, sizeof(back), hipMemcpyDeviceToHost);
        int hit = -1;
        for (int i = 0; i < 4 * n; ++i) if (back[i] == -5.f) hit = i;
        const long long addr = (long long)(int)c.voff + c.soff;
        printf("%-62s load -> %g (memory there holds %g), store landed at float %d\n", c.what, got,
               addr >= 0 && addr < 16 * n ? h[addr / 4] : -1.f, hit);
    }
    return 0;
}
