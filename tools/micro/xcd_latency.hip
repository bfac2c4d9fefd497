// Round-trip time of an 8-byte word handed back and forth between two workgroups, as the limiter's
// look-back does it (agent-scope relaxed atomic store / load), for partners on the same XCD and on
// different XCDs.   hipcc --offload-arch=gfx950 -O3 -o xcd_latency tools/micro/xcd_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_where(int* xcc) {
    if (threadIdx.x == 0) xcc[blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | 20);   // HW_REG_XCC_ID[3:0]
}

template <int SCOPE_LOAD>
__global__ void k_pingpong(unsigned long long* words, int a, int b, int rounds, long long* ticks) {
    const int me = blockIdx.x;
    if (threadIdx.x != 0 || (me != a && me != b)) return;
    unsigned long long* mine = words + (me == a ? 0 : 32);       // 256 bytes apart
    unsigned long long* theirs = words + (me == a ? 32 : 0);
    const long long t0 = wall_clock64();
    for (int r = 1; r <= rounds; ++r) {
        if (me == a) {
            __hip_atomic_store(theirs, (unsigned long long)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int spin = 0; __hip_atomic_load(mine, __ATOMIC_RELAXED, SCOPE_LOAD) != (unsigned long long)r; ++spin) {
                if (spin > 200000) { if (me == a) *ticks = -1; return; }      // (a load that keeps hitting a stale line)
                __builtin_amdgcn_s_sleep(1);
            }
        } else {
            for (int spin = 0; __hip_atomic_load(mine, __ATOMIC_RELAXED, SCOPE_LOAD) != (unsigned long long)r; ++spin) {
                if (spin > 200000) { if (me == a) *ticks = -1; return; }      // (a load that keeps hitting a stale line)
                __builtin_amdgcn_s_sleep(1);
            }
            __hip_atomic_store(theirs, (unsigned long long)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (me == a) *ticks = wall_clock64() - t0;
}

int main() {
    const int G = 1024;
    int* xcc; hipMalloc(&xcc, G * 4);
    k_where<<<G, 64>>>(xcc);
    std::vector<int> hx(G); hipMemcpy(hx.data(), xcc, G * 4, hipMemcpyDeviceToHost);
    printf("xcc of workgroups 0..23:");
    for (int i = 0; i < 24; ++i) printf(" %d", hx[i]);
    int mism = 0; for (int i = 0; i < G; ++i) mism += hx[i] != hx[i % 8];
    printf("\nworkgroups whose xcc differs from that of workgroup (index %% 8): %d of %d\n", mism, G);
    unsigned long long* words; hipMalloc(&words, 4096);
    long long* ticks; hipMalloc(&ticks, 8);
    const int rounds = 2000;
    int wc_khz = 0; hipDeviceGetAttribute(&wc_khz, hipDeviceAttributeWallClockRate, 0);
    const int partners[] = {8, 16, 256, 1, 2, 4, 7};
    for (int p : partners) {
        for (int scope = 0; scope < 2; ++scope) {
            hipMemset(words, 0, 4096);
            if (scope == 0) k_pingpong<__HIP_MEMORY_SCOPE_AGENT><<<G, 64>>>(words, 0, p, rounds, ticks);
            else k_pingpong<__HIP_MEMORY_SCOPE_WORKGROUP><<<G, 64>>>(words, 0, p, rounds, ticks);
            if (hipDeviceSynchronize() != hipSuccess) { printf("failed\n"); return 1; }
            long long t; hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
            printf("workgroup 0 (xcc %d) <-> %4d (xcc %d), load scope %s: %.3f us per round trip\n", hx[0], p, hx[p],
                   scope == 0 ? "agent    " : "workgroup", (double)t / wc_khz * 1e3 / rounds);
        }
    }
    return 0;
}
