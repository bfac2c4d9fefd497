// __global__ wrappers around the phase functions (device only; included by mgx.hip).
//
// Wave-level reductions here are gfx950 wave64 shuffles; nothing in this file is
// exercised by the CPU emulation (which drives the phase functions directly).
#pragma once

#include <hip/hip_runtime.h>

#include "analysis2_kernel.h"
#include "conv2_kernel.h"
#include "conv_delay_kernel.h"
#include "conv_wide_kernel.h"
#include "fft2_lanes.h"
#include "fir_plan.h"
#include "limiter_general.h"

namespace mgx {

#define MGX_LDS extern __shared__ __attribute__((aligned(16))) char mgx_smem[]

// Workgroup barrier that orders LDS traffic only: global loads issued before it stay in flight
// (__syncthreads() would drain vmcnt as well and serialise a software prefetch).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Between two phases of a transform that lie between pass 0 and the inverse of pass 0.  In a plan with
// Fft2::WAVE_LOCAL every wave reads there only what it wrote itself, and the LDS executes one wave's instructions in
// program order: all that is needed is that the COMPILER keeps the order (a memory clobber), no s_barrier.  With
// sixteen waves per workgroup and one workgroup per CU (the 16384-point kernels) a barrier idles the whole CU
// until the last wave arrives; this takes them from ten per block to five.  Other plans get the barrier.
template <class F>
__device__ __forceinline__ void pass_sync() {
#ifdef MGX_PASS_BARRIERS                     // A/B build: a workgroup barrier between all passes, as until round 4
    lds_barrier();
#else
    if constexpr (F::WAVE_LOCAL) asm volatile("" ::: "memory");
    else lds_barrier();
#endif
}

// The middle passes of a transform, forward and inverse, as the kernels below call them (each followed / preceded by
// the pass_sync the separate passes had).  A four-pass plan (16384 points) runs its two radix-8 passes as ONE phase
// with the exchange between them in the wave's lanes (fft2_lanes.h: bit-identical results, one LDS round trip and one
// wait fewer per direction: k_conv_wide<14> 144.3 -> 139.4 us, k_conv_delay<14> 238.5 -> 232.7, profiles/r06_b_*);
// -DMGX_SEPARATE_MID_PASSES builds the two-phase form for an A/B.  LANES = false: kernels that hold an accumulator row
// or a second transform's registers across the middle passes (k_conv<14>, the two- and four-transform analysis) keep
// the two-phase form -- both butterflies of a thread alive at once is 16 registers they do not have.
template <class F, bool LANES = true>
__device__ __forceinline__ void fwd_middle_passes(int tid, float2* lds, const float2* mid_table) {
#ifndef MGX_SEPARATE_MID_PASSES
    if constexpr (F::P == 4 && LANES) {
        Fft2Lanes<ilog2(F::N)>::fwd_mid_fused(tid, lds, mid_table);
        pass_sync<F>();
        return;
    }
#endif
    if constexpr (F::P >= 3) {
        F::fwd_mid(tid, lds, mid_table);
        pass_sync<F>();
    }
    if constexpr (F::P == 4) {
        F::fwd_mid2(mgx_opaque(tid), lds, mid_table);
        pass_sync<F>();
    }
}
template <class F, bool LANES = true>
__device__ __forceinline__ void inv_middle_passes(int tid, float2* lds, const float2* mid_table) {
#ifndef MGX_SEPARATE_MID_PASSES
    if constexpr (F::P == 4 && LANES) {
        pass_sync<F>();
        Fft2Lanes<ilog2(F::N)>::inv_mid_fused(tid, lds, mid_table);
        return;
    }
#endif
    if constexpr (F::P == 4) {
        pass_sync<F>();
        F::inv_mid2(tid, lds, mid_table);
    }
    if constexpr (F::P >= 3) {
        pass_sync<F>();
        F::inv_mid(mgx_opaque(tid), lds, mid_table);
    }
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// inclusive prefix sum over the 64 lanes: four steps inside each row of 16 (row_shr 1, 2, 4, 8; lanes
// without a source add 0), then lane 15 of rows 0 and 2 into rows 1 and 3, then lane 31 into rows 2 and 3
__device__ __forceinline__ int wave_inclusive_sum(int x) {
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);    // row_bcast:15
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);    // row_bcast:31
    return x;
}
__device__ __forceinline__ double wave_max_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}
// scratch: at least (threads/64) floats / doubles of LDS; result valid on thread 0
template <int THREADS>
__device__ __forceinline__ float block_max(float v, float* scratch) {
    v = wave_max(v);
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = 0.f;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 0; w < THREADS / 64; ++w) r = fmaxf(r, scratch[w]);
    }
    return r;
}
// the same behind an LDS-only barrier: global loads asked for earlier (a software prefetch) stay in flight
template <int THREADS>
__device__ __forceinline__ float block_max_lds(float v, float* scratch) {
    v = wave_max(v);
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    lds_barrier();
    float r = 0.f;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 0; w < THREADS / 64; ++w) r = fmaxf(r, scratch[w]);
    }
    return r;
}
template <int THREADS>
__device__ __forceinline__ double block_sum(double v, double* scratch) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 0; w < THREADS / 64; ++w) r += scratch[w];
    }
    return r;
}

}  // namespace mgx
#include "small_fft_kernels.h"     // fft_size 8 .. 32 (needs the block reductions above)
namespace mgx {

// The handle's error words: ONE int per kind of failure (error[0] look-back, error[1] tail, error[2] input), each set by
// a plain store of 1 -- kernels of different kinds never write the same word, so a later tail expiry cannot erase an
// earlier lost look-back word; the host folds them into the DEVICE_ERROR_* bits below.
// A limiter look-back that expires (limiter_kernel.h) means a lost word: the
// audio is wrong and the call fails.  An expired wait of k_correction_tail means its workgroups were not resident
// together (another process's kernels held the compute units): the host then runs the rounds again as one launch
// each, which wait for nobody (mgx.hip, check_device_error).  DEVICE_ERROR_INPUT is not a wait at all: the level
// analysis met a NaN or an infinity (k_match_curve), where the reference raises.
constexpr int DEVICE_ERROR_LOOKBACK = 1, DEVICE_ERROR_TAIL = 2, DEVICE_ERROR_INPUT = 4;
constexpr int DEVICE_ERROR_SLOT_LOOKBACK = 0, DEVICE_ERROR_SLOT_TAIL = 1, DEVICE_ERROR_SLOT_INPUT = 2, DEVICE_ERROR_SLOTS = 3;
#ifdef MGX_TEST_TAIL_EXPIRE
__device__ int g_test_tail_launches;
#endif

// ---------------------------------------------------------------------------
// code warming
// ---------------------------------------------------------------------------
// The kernels of this library are long stretches of straight-line code (a fully unrolled 8192-point
// transform is 38 KB, the limiter 105 KB), and every launch finds them evicted from the L2s by the hundreds of
// megabytes the previous kernel streamed.  The instruction cache then pulls them in line by line behind the
// first wave: ~32 ns per 64-byte line on some boxes of the pool and ~170 ns on others (same clocks, same
// memory latencies; tools/probe, profiles/r03_*_box_class.json) -- the whole difference between a
// "fast" and a "slow" box.  So the first eight workgroups of a launch -- one per XCD: the L2s are per XCD --
// read their own kernel's code AS DATA, 4 KB per load instruction and all of it in flight at once, which puts it
// into the XCD's L2; instruction fetch then finds it there.  (Every workgroup of the first generation doing so was
// measured: the limiter lost 14 us to the wait; so was warming again every 16th or 64th workgroup of an XCD, in
// case the streamed audio pushes the code out of the L2 again: +6 / +13 us.)  Sizes come from the code object's
// symbol table (mgx.hip, code_sizes_from_library); zero = no warming.
enum { CODE_ANALYZE = 0, CODE_MATCH_CURVE, CODE_CONV_PREP, CODE_CONV, CODE_ROUND, CODE_TAIL, CODE_LIMIT, CODE_KERNELS };
constexpr int CODE_VARIANTS = 16;                                  // second index: log2 of the transform; 0 / 1 = 256 / 1024-block limiter
constexpr int CODE_VARIANT_CONV_DELAY = 15;                        // [CODE_CONV][15]: k_conv_delay<14>
constexpr int CODE_VARIANT_CONV_WIDE = 6;                          // [CODE_CONV][6]: k_conv_wide<14> (there is no k_conv<6>)
__device__ int g_code_bytes[CODE_KERNELS][CODE_VARIANTS];
__device__ __forceinline__ void warm_code(int which, int variant = 0) {
    if (blockIdx.x >= 8 || threadIdx.x >= 64) return;             // workgroup b runs on XCD b % 8: one wave per L2
    const int bytes = g_code_bytes[which][variant];
    // Where this kernel's code starts: dispatch packet -> kernel descriptor -> entry offset (the AMDHSA code object
    // ABI: hsa_kernel_dispatch_packet_t::kernel_object at byte 32 points at the 64-byte descriptor, whose
    // kernel_code_entry_byte_offset at byte 16 is relative to the descriptor).  The window read is then exactly
    // [entry, entry + symbol size): it cannot run past the kernel whatever the compiler did with the block that
    // holds these loads (the program counter of this block is not the entry point).
    const char* packet = (const char*)__builtin_amdgcn_dispatch_ptr();      // (constant address space -> generic)
    const char* descriptor = *reinterpret_cast<const char* const*>(packet + 32);
    const char* entry = descriptor + *reinterpret_cast<const long long*>(descriptor + 16);
    int acc = 0;
    for (int off = (int)threadIdx.x * 64; off < bytes; off += 4096) acc += *reinterpret_cast<const volatile int*>(entry + off);
    if (acc == 0x7ffffff1) asm volatile("s_nop 0");                // (the sum is needed: the loads are waited for here)
}

// ---------------------------------------------------------------------------
// convolution
// ---------------------------------------------------------------------------
template <int LOG2N>
constexpr size_t conv_lds_bytes() {
    // (+ 16 floats for block_max and one int, the next pair's number: see k_conv)
    return ((size_t)Fft2<LOG2N>::LDS_ELEMS + Fft2<LOG2N>::MID_TABLE) * sizeof(float2) + 80;
}

// The phases of a kernel share index arithmetic (LDS addresses derived from the thread id).  Left
// alone, the compiler computes it once and keeps dozens of addresses alive across the whole
// kernel -- in scratch memory once the registers run out.  An empty asm makes the id opaque so
// that each phase re-derives its few addresses instead (the same goes for loop-invariant LDS
// table reads: hoisted out of a persistent loop they would live in scratch).
__device__ __forceinline__ int opaque(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

// workgroups of k_conv a CU holds (LDS and thread limits), and the waves per SIMD that makes: the
// register budget the kernel is compiled for
template <int LOG2N>
constexpr int conv_workgroups_per_cu() {
    constexpr int by_lds = (int)((size_t)160 * 1024 / conv_lds_bytes<LOG2N>());
    constexpr int by_threads = 2048 / Fft2<LOG2N>::T;
    constexpr int w = by_lds < by_threads ? by_lds : by_threads;
    return w < 1 ? 1 : (w > 2 ? 2 : w);             // more than two co-resident transforms do not pay
}
template <int LOG2N>
constexpr int conv_waves_per_simd() {
    constexpr int w = conv_workgroups_per_cu<LOG2N>() * (Fft2<LOG2N>::T / 64) / 4;
    return w < 2 ? 2 : w;
}

// One channel of one pair, from pass 0 (`pass0`) up to the inverse middle pass (conv2_kernel.h).
template <int LOG2N, bool SIDE, class Pass0>
__device__ __forceinline__ void conv_channel(int tid, const Conv2Args& a, float2* lds, const float2* mid_table,
                                             Pass0 pass0) {
    using CB = Conv2Block<LOG2N>;
    using F = Fft2<LOG2N>;
    typename CB::RowFilter rf;
    pass0();
    // (the scheduler must not lift the filter loads above pass 0: there is no room for them yet)
    __builtin_amdgcn_sched_barrier(0);
    CB::fetch_filter(tid, SIDE ? a.h_side : a.h_mid, rf);      // a phase early: the middle pass hides its latency
    __syncthreads();
    fwd_middle_passes<F, false>(opaque(tid), lds, mid_table);
    CB::phase_filter(tid, rf, lds);
    inv_middle_passes<F, false>(opaque(tid), lds, mid_table);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);      // keeps the next phase's LDS reads from being lifted into this one
}
// uniformly partitioned overlap-save: one forward transform per filter partition, products
// accumulated on the thread's row, one inverse transform
template <int LOG2N, bool SIDE>
__device__ __forceinline__ void conv_channel_partitioned(int tid, long long pair, bool edge, const Conv2Args& a,
                                                         const typename Conv2Block<LOG2N>::Persist& ps,
                                                         float2* lds, const float2* mid_table) {
    using CB = Conv2Block<LOG2N>;
    using F = Fft2<LOG2N>;
    const float2* h = SIDE ? a.h_side : a.h_mid;
    typename CB::RowFilter rf;
    typename CB::RowAcc acc;
    CB::clear_acc(acc);
    // radix-32 middle passes (16384 points) need the registers the early filter fetch would hold
    constexpr bool LATE_FILTER = F::R(1) >= 32;
    for (int k = 0; k < a.parts; ++k) {
        CB::template phase_load<SIDE>(opaque(tid), pair, edge, a, ps, lds, k);
        __builtin_amdgcn_sched_barrier(0);
        if (!LATE_FILTER) CB::fetch_filter(tid, h + (size_t)k * F::N, rf);
        __syncthreads();
        fwd_middle_passes<F, false>(opaque(tid), lds, mid_table);
        __builtin_amdgcn_sched_barrier(0);
        if (LATE_FILTER) CB::fetch_filter(opaque(tid), h + (size_t)k * F::N, rf);
        CB::phase_accumulate(tid, rf, lds, acc);
        __syncthreads();                    // (the next partition's pass 0 writes everywhere)
        __builtin_amdgcn_sched_barrier(0);
    }
    CB::phase_finish_row(tid, acc, lds);
    __builtin_amdgcn_sched_barrier(0);
    inv_middle_passes<F, false>(opaque(tid), lds, mid_table);
    __syncthreads();
}

// One pair of output blocks: mid channel, then side channel + epilogue.
// (Issuing the NEXT pair's frame loads before the epilogue's stores -- software pipelining -- was
// built and measured: the 48-96 registers in flight make the 8192-point kernel spill at the 256 a
// wave has here, and a spilling kernel is far slower than an unpipelined one, 294 vs 153 us.)
template <int LOG2N, bool MULTI>
__device__ __forceinline__ float conv_pair(int tid, long long pair, const Conv2Args& a,
                                           const typename Conv2Block<LOG2N>::Persist& ps, float2* lds,
                                           const float2* mid_table) {
    using CB = Conv2Block<LOG2N>;
    const bool edge = !CB::interior(pair, a.n, a.parts);
    typename CB::Kept kept;
    if constexpr (MULTI) {
        conv_channel_partitioned<LOG2N, false>(tid, pair, edge, a, ps, lds, mid_table);
        CB::phase_keep_mid(tid, ps, lds, kept);
        __syncthreads();
        conv_channel_partitioned<LOG2N, true>(tid, pair, edge, a, ps, lds, mid_table);
    } else {
        typename CB::Raw raw;
        typename CB::Held held;
        CB::fetch_frames(tid, pair, a, 0, raw);
        conv_channel<LOG2N, false>(
            tid, a, lds, mid_table, [&]() { CB::phase_pass0_mid(tid, raw, ps, lds, held); });
        CB::phase_keep_mid(tid, ps, lds, kept);
        __syncthreads();
        conv_channel<LOG2N, true>(
            tid, a, lds, mid_table, [&]() { CB::phase_pass0_side(tid, held, ps, lds); });
    }
    const float pk = CB::phase_store(tid, pair, edge, a, ps, lds, kept);
    return pk;
}

// Persistent workgroups.  gridDim.x is a multiple of 8; workgroup w is (observed to be) placed on
// XCD w % 8, so XCD x walks its own contiguous eighth of the track and the workgroups resident on
// it work on neighbouring pairs: the overlap between neighbours is re-read from that XCD's L2,
// not from HBM.  Placement only affects speed, never results.
// After its first pair a workgroup takes the next pair of its XCD's range from a counter (asked for at
// the start of a pair, needed at its end) instead of a fixed stride: an 8-minute track is 5.05 pairs
// per workgroup, and with fixed strides the kernel lasted six pair-times for it.
// Two workgroups per CU (LDS): the second launch bound is waves per SIMD, i.e. the register budget.
// MULTI = more than one filter partition (its own instantiation: the accumulator row costs registers
// the plain kernel should not pay for)
template <int LOG2N, bool MULTI>
__global__ __launch_bounds__((Fft2<LOG2N>::T), (conv_waves_per_simd<LOG2N>())) void k_conv(Conv2Args a) {
    warm_code(CODE_CONV, LOG2N);
    using CB = Conv2Block<LOG2N>;
    using F = Fft2<LOG2N>;
    MGX_LDS;
    float2* lds = reinterpret_cast<float2*>(mgx_smem);
    float2* mid_table = lds + F::LDS_ELEMS;
    float* scratch = reinterpret_cast<float*>(mid_table + F::MID_TABLE);
    const int tid = threadIdx.x;
    typename CB::Persist ps;
    CB::load_persist(tid, a.tw, mid_table, ps);
    __syncthreads();
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    const long long per = (a.npairs + 7) >> 3;
    const long long end = min(a.npairs, (xcd + 1) * per);
    const long long first = xcd * per + slot;
    int* next_slot = reinterpret_cast<int*>(scratch + 16);
    long long pair = first;
    while (pair < end) {
        unsigned ticket = 0;
        if (tid == 0) ticket = atomicAdd(a.queue + xcd, 1u);
        // The pass-0 twiddles stay in registers across pairs, but nothing derived from them (or
        // from the thread id) should: hoisted out of this loop it would sit in VGPRs it does
        // not have.  An empty asm makes the values opaque per iteration.
#pragma unroll
        for (int q = 0; q < F::LB0; ++q) asm volatile("" : "+v"(ps.tw0.b[q].x), "+v"(ps.tw0.b[q].y));
        const float pk = conv_pair<LOG2N, MULTI>(tid, pair, a, ps, lds, mid_table);
        const float bp = block_max<F::T>(pk, scratch);
        if (tid == 0) {
            if (a.pair_peak) a.pair_peak[pair] = bp;
            *next_slot = (int)ticket;
        }
        __syncthreads();
        pair = xcd * per + slots + *next_slot;
    }
    // the last workgroup to run out of pairs leaves the counters at zero for the next launch
    if (tid == 0 && atomicAdd(a.queue + 8, 1u) == gridDim.x - 1) {
        for (int i = 0; i < 9; ++i) a.queue[i] = 0;
    }
}

#ifdef MGX_DEV_CONV_PHASES         // development builds only: where a block's time goes (tools/conv_delay_phases.py)
constexpr int DEV_CONV_BLOCKS = 16384;
__device__ unsigned mgx_dev_conv_ticks[DEV_CONV_BLOCKS][8];          // [block][mark]: ticks since the previous mark; [7] = start
#define DEV_CONV_MARK(k)                                                                             \
    do {                                                                                             \
        if (threadIdx.x == 0 && b >= 0 && b < DEV_CONV_BLOCKS) {                                     \
            const long long now = wall_clock64();                                                    \
            mgx_dev_conv_ticks[b][k] = (unsigned)(now - dev_last);                                   \
            dev_last = now;                                                                          \
        }                                                                                            \
    } while (0)
#else
#define DEV_CONV_MARK(k)
#endif
// A filter of two partitions as a frequency-domain delay line (conv_delay_kernel.h): workgroup w takes blocks
// [w * run, (w + 1) * run) one after the other, the partition-1 product of a block carried to the next in registers;
// it starts with the forward transform of block w * run - 1 (carry only).  a.npairs counts BLOCKS here and
// pair_peak has one entry per block.
template <int LOG2N>
__global__ __launch_bounds__((Fft2<LOG2N>::T), (conv_waves_per_simd<LOG2N>())) void k_conv_delay(Conv2Args a) {
    warm_code(CODE_CONV, CODE_VARIANT_CONV_DELAY);
    using CD = ConvDelay<LOG2N>;
    using CB = Conv2Block<LOG2N>;
    using F = Fft2<LOG2N>;
    MGX_LDS;
    float2* lds = reinterpret_cast<float2*>(mgx_smem);
    float2* mid_table = lds + F::LDS_ELEMS;
    float* scratch = reinterpret_cast<float*>(mid_table + F::MID_TABLE);
    const int tid = threadIdx.x;
    typename CB::Persist ps;
    CB::load_persist(tid, a.tw, mid_table, ps);
    __syncthreads();
    const long long first = (long long)blockIdx.x * a.run;
    const long long end = min(a.npairs, first + a.run);
    typename CD::Carry carry;
    CD::clear(carry);
    // the older half of a block's window is the newer half of the block before: kept in registers (as mid / side),
    // and the newer half is asked for a block ahead -- every frame is fetched once per run and its latency lies under
    // the end of the block before (275 us with the whole window loaded at the top of a block, 262 with the older half
    // kept, 243 with the newer half asked for ahead: profiles/r04_v_conv_delay_pairs.txt)
    typename CD::HeldFrames held;
    typename CD::HalfFrames newer;
    CD::prime(tid, first - 1, a, held);
    CD::template fetch_half<CD::R0 / 2>(tid, first - 1, a, newer);
    for (long long b = first - 1; b < end; ++b) {
        // (as in k_conv: nothing derived from the pass-0 twiddles or the thread id may be hoisted out of the loop)
#pragma unroll
        for (int q = 0; q < F::LB0; ++q) asm volatile("" : "+v"(ps.tw0.b[q].x), "+v"(ps.tw0.b[q].y));
#ifdef MGX_DEV_CONV_PHASES
        long long dev_last = wall_clock64();
        if (threadIdx.x == 0 && b >= 0 && b < DEV_CONV_BLOCKS) mgx_dev_conv_ticks[b][7] = (unsigned)dev_last;
#endif
        CD::phase_pass0_held(opaque(tid), ps, held, newer, lds);       // (the newer half was asked for a block ago)
        lds_barrier();
        DEV_CONV_MARK(0);                   // frames, pass 0, barrier
        fwd_middle_passes<F>(opaque(tid), lds, mid_table);
        __builtin_amdgcn_sched_barrier(0);
        DEV_CONV_MARK(1);                   // middle passes
        CD::phase_row(opaque(tid), lds);
        lds_barrier();
        DEV_CONV_MARK(2);                   // row forward, barrier
        CD::phase_multiply(opaque(tid), a, carry, lds);
        lds_barrier();                    // every mirror row has been read: the rows may be written again
        __builtin_amdgcn_sched_barrier(0);
        DEV_CONV_MARK(3);                   // multiply, barrier
        if (b < first) {                    // (uniform) the block in front of the run: its carry only
            CD::template fetch_half<CD::R0 / 2>(opaque(tid), b + 1, a, newer);
            continue;
        }
        CD::phase_row_back(opaque(tid), lds);
        inv_middle_passes<F>(opaque(tid), lds, mid_table);
        // the newer half of the next block's window: asked for here, its latency under the last inverse pass and the
        // stores (asked for a phase earlier, behind the multiply, it costs 24 B more scratch and 11 us)
        CD::template fetch_half<CD::R0 / 2>(opaque(tid), b + 1, a, newer);
        lds_barrier();                      // (every barrier of the loop orders LDS traffic only: a __syncthreads() would
                                            // wait for these loads, and for the stores below, which nobody here reads)
        DEV_CONV_MARK(4);                   // row back, inverse middle passes, barrier
        const float pk = CD::phase_store(opaque(tid), b, a, ps, lds);
        const float bp = block_max_lds<F::T>(pk, scratch);      // (a barrier inside: the LDS is free for the next block)
        if (tid == 0 && a.pair_peak) a.pair_peak[b] = bp;
        DEV_CONV_MARK(5);                   // inverse pass 0, stores, peak
    }
}

// F taps on N = 4F blocks (conv_wide_kernel.h): every workgroup takes one block per round of the grid and asks for the
// frames of its next block while the current one is in its inverse transform.  a.npairs counts BLOCKS of 3N/4 frames
// here and pair_peak has one entry per block.
// Which block: workgroup w is (observed to be) placed on XCD w % 8.  Round i covers blocks [i G, (i + 1) G) in eight runs
// of G / 8 consecutive blocks, one run per XCD, so that a block and its neighbour -- whose windows share N / 4 frames --
// are fetched through the same L2 (until round 6: blocks w, w + G, ... -- neighbours on different XCDs, the shared quarter
// fetched from HBM twice: FETCH_SIZE 112 MiB x 2 for 169 MB of frames).  Placement affects speed only.
__device__ __forceinline__ long long conv_wide_block(unsigned w, unsigned round, unsigned grid) {
#ifndef MGX_CONV_WIDE_STRIDED
    if ((grid & 7u) == 0) return ((long long)round * 8 + (w & 7u)) * (grid >> 3) + (w >> 3);
#endif
    return (long long)round * grid + w;
}
template <int LOG2N>
__global__ __launch_bounds__((Fft2<LOG2N>::T), (conv_waves_per_simd<LOG2N>())) void k_conv_wide(Conv2Args a) {
    warm_code(CODE_CONV, CODE_VARIANT_CONV_WIDE);
    using CW = ConvWide<LOG2N>;
    using CB = Conv2Block<LOG2N>;
    using F = Fft2<LOG2N>;
    MGX_LDS;
    float2* lds = reinterpret_cast<float2*>(mgx_smem);
    float2* mid_table = lds + F::LDS_ELEMS;
    float* scratch = reinterpret_cast<float*>(mid_table + F::MID_TABLE);
    const int tid = threadIdx.x;
    typename CB::Persist ps;
    CB::load_persist(tid, a.tw, mid_table, ps);
    typename CW::Frames frames;
    CW::fetch(tid, conv_wide_block(blockIdx.x, 0, gridDim.x), a, frames);
    __syncthreads();
    for (unsigned round = 0;; ++round) {
        const long long b = conv_wide_block(blockIdx.x, round, gridDim.x);
        if (b >= a.npairs) break;
        // (as in k_conv: nothing derived from the pass-0 twiddles or the thread id may be hoisted out of the loop)
#pragma unroll
        for (int q = 0; q < F::LB0; ++q) asm volatile("" : "+v"(ps.tw0.b[q].x), "+v"(ps.tw0.b[q].y));
#ifdef MGX_DEV_CONV_PHASES
        long long dev_last = wall_clock64();
        if (threadIdx.x == 0 && b < DEV_CONV_BLOCKS) mgx_dev_conv_ticks[b][7] = (unsigned)dev_last;
#endif
        CW::phase_pass0(opaque(tid), ps, frames, lds);
        lds_barrier();
        DEV_CONV_MARK(0);                   // pass 0, barrier
        fwd_middle_passes<F>(opaque(tid), lds, mid_table);
        __builtin_amdgcn_sched_barrier(0);
        DEV_CONV_MARK(1);                   // middle passes
        typename CW::Filters filt;
        CW::fetch_filters(opaque(tid), a, filt);          // needed behind the next barrier
#ifdef MGX_CONV_WIDE_ROW_IN_LDS                           // A/B build: the whole row through the LDS, as until round 6
        CW::phase_row(opaque(tid), lds);
        lds_barrier();
        DEV_CONV_MARK(2);                   // row forward, barrier
        CW::phase_multiply(opaque(tid), filt, lds);
        lds_barrier();                    // the partner has written this row's upper half
        __builtin_amdgcn_sched_barrier(0);
        DEV_CONV_MARK(3);                   // multiply, barrier
        CW::phase_row_back(opaque(tid), lds);
#else
        typename CW::Kept kept;                           // the own half of the row stays in registers (conv_wide_kernel.h)
        CW::phase_row_keep(opaque(tid), kept, lds);
        lds_barrier();
        DEV_CONV_MARK(2);                   // row forward, barrier
        CW::phase_multiply_keep(opaque(tid), filt, kept, lds);
        lds_barrier();                    // the partner has written this row's upper half
        __builtin_amdgcn_sched_barrier(0);
        DEV_CONV_MARK(3);                   // multiply, barrier
        CW::phase_row_back_keep(opaque(tid), kept, lds);
#endif
        inv_middle_passes<F>(opaque(tid), lds, mid_table);
        // the next block's window: its latency under the last inverse pass and the stores -- the barriers from here
        // to the top of the loop order LDS traffic only (a __syncthreads() would wait for these loads)
        CW::fetch(opaque(tid), conv_wide_block(blockIdx.x, round + 1, gridDim.x), a, frames);
        lds_barrier();
        DEV_CONV_MARK(4);                   // row back, inverse middle passes, barrier
        const float pk = CW::phase_store(opaque(tid), b, a, ps, lds);
        const float bp = block_max_lds<F::T>(pk, scratch);      // (a barrier inside: the LDS is free for the next block)
        if (tid == 0 && a.pair_peak) a.pair_peak[b] = bp;
        DEV_CONV_MARK(5);                   // inverse pass 0, stores, peak
    }
}
// its filter spectra: grid = 2 (mid, side); taps = [2][N/4] float
template <int LOG2N>
__global__ __launch_bounds__((Fft2<LOG2N>::T)) void k_conv_wide_prep(const float* taps, const float2* tw, float2* tables,
                                                                   const double* gain_ptr, double gain) {
    warm_code(CODE_CONV_PREP, CODE_VARIANT_CONV_WIDE);
    using CW = ConvWide<LOG2N>;
    using CB = Conv2Block<LOG2N>;
    using F = Fft2<LOG2N>;
    MGX_LDS;
    float2* lds = reinterpret_cast<float2*>(mgx_smem);
    float2* mid_table = lds + F::LDS_ELEMS;
    const int tid = threadIdx.x, ch = blockIdx.x;
    const double g = gain_ptr ? *gain_ptr * gain : gain;
    typename CB::Persist ps;
    CB::load_persist(tid, tw, mid_table, ps);
    CW::phase_load_taps(tid, taps + (size_t)ch * CW::TAPS, ps, lds);
    __syncthreads();
    fwd_middle_passes<F>(tid, lds, mid_table);
    CB::phase_write_filter(tid, lds, (float)(g / (double)F::N), tables + (size_t)ch * F::N);
}

// filter spectra: grid = 2 * parts; taps = [2][parts * N/2] float (mid then side), tables =
// [2][parts][N] float2.  Workgroup (ch, k) transforms partition k of channel ch.
template <int LOG2N>
__global__ __launch_bounds__((Fft2<LOG2N>::T)) void k_conv_prep(const float* taps, const float2* tw, float2* tables,
                                                              int parts, const double* gain_ptr, double gain) {
    warm_code(CODE_CONV_PREP, LOG2N);
    using CB = Conv2Block<LOG2N>;
    using F = Fft2<LOG2N>;
    MGX_LDS;
    float2* lds = reinterpret_cast<float2*>(mgx_smem);
    float2* mid_table = lds + F::LDS_ELEMS;
    const int tid = threadIdx.x, ch = blockIdx.x / parts, k = blockIdx.x % parts;
    const double g = gain_ptr ? *gain_ptr * gain : gain;      // asked for first: nothing below should wait for it
    typename CB::Persist ps;
    CB::load_persist(tid, tw, mid_table, ps);
    CB::phase_load_taps(tid, taps + ((size_t)ch * parts + k) * CB::TAPS, ps, lds);
    __syncthreads();
    fwd_middle_passes<F>(tid, lds, mid_table);
    CB::phase_write_filter(tid, lds, (float)(g / (double)F::N), tables + ((size_t)ch * parts + k) * F::N);
}

// ---------------------------------------------------------------------------
// analysis
// ---------------------------------------------------------------------------
template <int LOG2N>
constexpr size_t analysis_lds_bytes() {
    // (+ a double and a float of reduction scratch per wave of the workgroup)
    return ((size_t)Fft2<LOG2N>::LDS_ELEMS + Fft2<LOG2N>::MID_TABLE) * sizeof(float2) + (Fft2<LOG2N>::T / 64) * 12;
}

// waves per SIMD the LDS footprint admits (4 SIMDs per CU): the register budget follows from it
#ifndef MGX_ANALYZE_MAX_WGS
#define MGX_ANALYZE_MAX_WGS 8          // experiments: cap the workgroups per CU (more registers each)
#endif
template <int LOG2N>
constexpr int analysis_waves_per_simd() {
    constexpr int by_lds = (int)((size_t)160 * 1024 / analysis_lds_bytes<LOG2N>());
    constexpr int wgs = by_lds < MGX_ANALYZE_MAX_WGS ? by_lds : MGX_ANALYZE_MAX_WGS;
    constexpr int w = (wgs > 8 ? 8 : wgs) * (Fft2<LOG2N>::T / 64) / 4;
    return w < 1 ? 1 : (w > 8 ? 8 : w);
}

// One grid for one or two tracks: workgroups [0, nwg0) belong to `a0`, the rest to `a1` (target and
// reference of a pair in ONE launch: no boundary and no half-empty chip between the two passes).
template <int LOG2N>
__global__ __launch_bounds__(Fft2<LOG2N>::T, analysis_waves_per_simd<LOG2N>()) void k_analyze(AnalysisArgs a0, AnalysisArgs a1,
                                                                                             int nwg0) {
    warm_code(CODE_ANALYZE, LOG2N);
    using AB = Analysis2Block<LOG2N>;
    using F = Fft2<LOG2N>;
    MGX_LDS;
    float2* lds = reinterpret_cast<float2*>(mgx_smem);
    float2* mid_table = lds + F::LDS_ELEMS;
    double* dscratch = reinterpret_cast<double*>(mid_table + F::MID_TABLE);
    float* fscratch = reinterpret_cast<float*>(dscratch + F::T / 64);   // (one slot per wave each)
    const bool second = (int)blockIdx.x >= nwg0;                 // uniform
    const AnalysisArgs& a = second ? a1 : a0;
    const int tid = threadIdx.x, wg = second ? blockIdx.x - nwg0 : blockIdx.x;
    const int d = wg / a.chunks_per_piece, ch = wg % a.chunks_per_piece;
    typename AB::Thread th;
    AB::init(th);
    typename AB::Persist ps;
    AB::load_persist(tid, a.tw, mid_table, ps);
    __syncthreads();
    int s0, s1;
    AB::chunk_segments(a, ch, s0, s1);
#ifdef MGX_ANALYZE_NOT_AHEAD                 // A/B build: every segment's frames asked for at its top, as until round 6
    constexpr bool AHEAD = false;
#elif defined(MGX_ANALYZE_AHEAD_ALL)        // experiment: the smaller transforms too
    constexpr bool AHEAD = LOG2N >= 11;
#else
    constexpr bool AHEAD = LOG2N == 14;
#endif
    typename AB::Raw raw;
    if (AHEAD) AB::fetch(tid, (long long)d * a.piece + (long long)s0 * F::N, a, raw);
    for (int s = s0; s < s1; ++s) {
        // (A software prefetch of the next segment, round 4: asked for after pass 0, after the middle pass or after
        // the row pass, with the thread id opaque per iteration so that nothing is hoisted -- at 128 VGPRs every
        // variant spills (116 - 172 B: the kernel sits at 124 registers without it) and the spilling kernel ran 152 us;
        // at three workgroups per CU and 168 VGPRs it fits without scratch and changes nothing: 100 us with and
        // without it, against 92 us for four workgroups without.  Hiding the loads is worth exactly the fourth
        // workgroup it costs.  profiles/r04_b_ab_variants.txt.  Round 5: one word of each 128-byte line of the next
        // segment asked for behind this segment's loads (one register, no scratch at 16384 points) so that the L2
        // holds it -- 149 -> 173 us at 16384 points, 88 -> 126 us at 4096: profiles/r05_p_analysis_touch_next_segment.txt)
        // (Round 6, 16384 points only -- one workgroup per CU, nobody else to cover a segment's load latency: the next
        // segment's frames are asked for behind the row pass, so that they travel under the magnitudes phase; the registers
        // they wait in are live across that phase only: k_analyze<14> 149.6 -> 130.2 us, profiles/r06_f_*.  One `raw`, no branches around it -- the form k_conv_wide
        // prefetches in: behind the last segment of the chunk the loads run into the next chunk's frames, or past the
        // track where the buffer's range check answers zeros, and are dropped.)
        if (!AHEAD) AB::fetch(tid, (long long)d * a.piece + (long long)s * F::N, a, raw);
        AB::phase_load(tid, raw, ps, th, lds);
        lds_barrier();
        fwd_middle_passes<F>(tid, lds, mid_table);
        typename AB::Row own;
        AB::phase_row(tid, own, lds);
        if (AHEAD) {
            __builtin_amdgcn_sched_barrier(0);          // (not above the row pass: its butterflies need the registers)
            AB::fetch(mgx_opaque(tid), (long long)d * a.piece + (long long)(s + 1) * F::N, a, raw);
            __builtin_amdgcn_sched_barrier(0);
        }
        lds_barrier();
        AB::phase_magnitudes(tid, own, th, lds);
        lds_barrier();
    }
    if (ch == a.chunks_per_piece - 1) {
        AB::phase_loose_frames(tid, (long long)d * a.piece + (long long)a.segs_per_piece * F::N,
                               (long long)(d + 1) * a.piece, true, a, th);
        if (d == a.divisions - 1)
            AB::phase_loose_frames(tid, (long long)a.divisions * a.piece, a.n, false, a, th);
    }
    AB::phase_write_spectrum(tid, wg, a, th);
    const double ss = block_sum<F::T>(th.sumsq, dscratch);
    const float pk = block_max<F::T>(th.peak, fscratch);
    if (tid == 0) {
        a.wg_sumsq[wg] = ss;
        a.wg_peak[wg] = pk;
    }
}

// fft_size = 2 * Fft2<LOG2H>::N (analysis2_kernel.h, AnalysisDouble): the same grid layout and outputs, two
// transforms per segment
template <int LOG2H>
__global__ __launch_bounds__(Fft2<LOG2H>::T, analysis_waves_per_simd<LOG2H>()) void k_analyze_double(AnalysisArgs a0,
                                                                                                    AnalysisArgs a1,
                                                                                                    int nwg0) {
    using AD = AnalysisDouble<LOG2H>;
    using F = Fft2<LOG2H>;
    MGX_LDS;
    float2* lds = reinterpret_cast<float2*>(mgx_smem);
    float2* mid_table = lds + F::LDS_ELEMS;
    double* dscratch = reinterpret_cast<double*>(mid_table + F::MID_TABLE);
    float* fscratch = reinterpret_cast<float*>(dscratch + F::T / 64);   // (one slot per wave each)
    const bool second = (int)blockIdx.x >= nwg0;                 // uniform
    const AnalysisArgs& a = second ? a1 : a0;
    const int tid = threadIdx.x, wg = second ? blockIdx.x - nwg0 : blockIdx.x;
    const int d = wg / a.chunks_per_piece, ch = wg % a.chunks_per_piece;
    typename AD::Thread th;
    AD::init(th);
    typename AD::Persist ps;
    AD::AB::load_persist(tid, a.tw, mid_table, ps);
    __syncthreads();
    int s0, s1;
    AD::AB::chunk_segments(a, ch, s0, s1);
    for (int s = s0; s < s1; ++s) {
        const long long start = (long long)d * a.piece + (long long)s * 2 * F::N;
        AD::template phase_load<false>(tid, start, a, ps, th, lds);
        lds_barrier();
        fwd_middle_passes<F, false>(tid, lds, mid_table);
        AD::phase_row(tid, lds);
        lds_barrier();
        AD::template phase_magnitudes<false>(tid, th, lds);
        lds_barrier();
        AD::template phase_load<true>(tid, start, a, ps, th, lds);
        lds_barrier();
        fwd_middle_passes<F, false>(tid, lds, mid_table);
        AD::phase_row(tid, lds);
        lds_barrier();
        AD::template phase_magnitudes<true>(tid, th, lds);
        lds_barrier();
    }
    if (ch == a.chunks_per_piece - 1) {
        AD::phase_loose_frames(tid, (long long)d * a.piece + (long long)a.segs_per_piece * 2 * F::N,
                               (long long)(d + 1) * a.piece, true, a, th);
        if (d == a.divisions - 1)
            AD::phase_loose_frames(tid, (long long)a.divisions * a.piece, a.n, false, a, th);
    }
    AD::phase_write_spectrum(tid, wg, a, th);
    const double ss = block_sum<F::T>(th.sumsq, dscratch);
    const float pk = block_max<F::T>(th.peak, fscratch);
    if (tid == 0) {
        a.wg_sumsq[wg] = ss;
        a.wg_peak[wg] = pk;
    }
}

// fft_size = 4 * Fft2<LOG2H>::N (analysis2_kernel.h, AnalysisQuad): the same grid layout and outputs, four transforms
// per segment; the spectrum sums live in the workgroup's slice of a.wg_pack until the end
template <int LOG2H>
__global__ __launch_bounds__(Fft2<LOG2H>::T, analysis_waves_per_simd<LOG2H>()) void k_analyze_quad(AnalysisArgs a0,
                                                                                                  AnalysisArgs a1,
                                                                                                  int nwg0) {
    using AQ = AnalysisQuad<LOG2H>;
    using F = Fft2<LOG2H>;
    MGX_LDS;
    float2* lds = reinterpret_cast<float2*>(mgx_smem);
    float2* mid_table = lds + F::LDS_ELEMS;
    double* dscratch = reinterpret_cast<double*>(mid_table + F::MID_TABLE);
    float* fscratch = reinterpret_cast<float*>(dscratch + F::T / 64);   // (one slot per wave each)
    const bool second = (int)blockIdx.x >= nwg0;                 // uniform
    const AnalysisArgs& a = second ? a1 : a0;
    const int tid = threadIdx.x, wg = second ? blockIdx.x - nwg0 : blockIdx.x;
    const int d = wg / a.chunks_per_piece, ch = wg % a.chunks_per_piece;
    typename AQ::Thread th;
    AQ::init(th);
    typename AQ::Persist ps;
    AQ::AB::load_persist(tid, a.tw, mid_table, ps);
    AQ::phase_clear(tid, wg, a);
    __syncthreads();
    int s0, s1;
    AQ::AB::chunk_segments(a, ch, s0, s1);
    auto transform = [&]() {
        lds_barrier();
        fwd_middle_passes<F, false>(opaque(tid), lds, mid_table);
        AQ::phase_row(opaque(tid), lds);
        lds_barrier();
    };
    for (int s = s0; s < s1; ++s) {
        const long long start = (long long)d * a.piece + (long long)s * 4 * F::N;
        typename AQ::Pairs u;
        AQ::template phase_load<false, 0>(opaque(tid), start, a, ps, th, lds);
        transform();
        AQ::phase_unmix(opaque(tid), u, lds);
        AQ::phase_keep(opaque(tid), wg, a, u);
        lds_barrier();
        AQ::template phase_load<false, 1>(opaque(tid), start, a, ps, th, lds);
        transform();
        AQ::phase_unmix(opaque(tid), u, lds);
        AQ::template phase_magnitudes<false>(opaque(tid), wg, a, u);
        lds_barrier();
        AQ::template phase_load<true, 0>(opaque(tid), start, a, ps, th, lds);
        transform();
        AQ::phase_unmix(opaque(tid), u, lds);
        AQ::phase_keep(opaque(tid), wg, a, u);
        lds_barrier();
        AQ::template phase_load<true, 1>(opaque(tid), start, a, ps, th, lds);
        transform();
        AQ::phase_unmix(opaque(tid), u, lds);
        AQ::template phase_magnitudes<true>(opaque(tid), wg, a, u);
        lds_barrier();
    }
    if (ch == a.chunks_per_piece - 1) {
        AQ::phase_loose_frames(tid, (long long)d * a.piece + (long long)a.segs_per_piece * 4 * F::N,
                               (long long)(d + 1) * a.piece, true, a, th);
        if (d == a.divisions - 1)
            AQ::phase_loose_frames(tid, (long long)a.divisions * a.piece, a.n, false, a, th);
    }
    AQ::phase_write_spectrum(tid, wg, a);
    const double ss = block_sum<F::T>(th.sumsq, dscratch);
    const float pk = block_max<F::T>(th.peak, fscratch);
    if (tid == 0) {
        a.wg_sumsq[wg] = ss;
        a.wg_peak[wg] = pk;
    }
}

// ---- piece statistics -> decisions (match_levels.py:62-71,93-103), one 1024-thread workgroup ----
// Step 1: wave w sums the chunk partials of pieces w, w+16, ... (lanes = chunks) into LDS.
// Step 2: thread d owns piece d: rms, mean of squares, rms >= average, RMS of the loud ones.
// All reductions are fixed trees, so results are run-to-run identical.
__device__ __forceinline__ void piece_sums_to_lds(const double* partial, int chunks, int divisions, double* sums) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwaves = blockDim.x >> 6;
    for (int d = wave; d < divisions; d += nwaves) {
        double s = 0.0;
        for (int ch = lane; ch < chunks; ch += 64) s += partial[(size_t)d * chunks + ch];
        s = wave_sum(s);
        if (lane == 0) sums[d] = s;
    }
    __syncthreads();
}
// The same sums with L lanes side by side on a piece (L a power of two, as many as the workgroup has for
// `divisions` pieces, at most 64) and a butterfly over them: a fixed order too, and no wave walks alone
// through its pieces.
__device__ __forceinline__ void piece_sums_by_groups(const double* partial, int chunks, int divisions, double* sums) {
    int l = 64;
    while (l > 1 && l * divisions > (int)blockDim.x) l >>= 1;
    const int part = threadIdx.x & (l - 1), per_pass = blockDim.x / l;
    for (int d0 = 0; d0 < divisions; d0 += per_pass) {
        const int d = d0 + threadIdx.x / l;
        double s = 0.0;
        if (d < divisions)
            for (int ch = part; ch < chunks; ch += l) s += partial[(size_t)d * chunks + ch];
        for (int o = l >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (d < divisions && part == 0) sums[d] = s;
    }
    __syncthreads();
}
// returns (on every thread) average rms, match rms and the loud count; optionally stores rms/loud
template <int THREADS>
__device__ __forceinline__ void decide_loud(const double* sums, int divisions, long long piece, double inv_c,
                                            double* red, double* rms_out, int* loud_out, double& avg,
                                            double& match, int& count) {
    double acc = 0.0;
    for (int d = threadIdx.x; d < divisions; d += blockDim.x) {
        const double r = sqrt(sums[d] / (double)piece) * inv_c;
        acc += r * r;
    }
    double tot = block_sum<THREADS>(acc, red);
    if (threadIdx.x == 0) red[16] = sqrt(tot / divisions);
    __syncthreads();
    avg = red[16];
    double lacc = 0.0, lcnt = 0.0;
    for (int d = threadIdx.x; d < divisions; d += blockDim.x) {
        const double r = sqrt(sums[d] / (double)piece) * inv_c;
        const bool l = r >= avg;
        if (l) { lacc += r * r; lcnt += 1.0; }
        if (rms_out) rms_out[d] = r;
        if (loud_out) loud_out[d] = l ? 1 : 0;
    }
    __syncthreads();
    tot = block_sum<THREADS>(lacc, red);
    if (threadIdx.x == 0) red[17] = tot;
    __syncthreads();
    const double cnt = block_sum<THREADS>(lcnt, red + 18);
    if (threadIdx.x == 0) red[40] = cnt;
    __syncthreads();
    count = (int)red[40];
    match = sqrt(red[17] / red[40]);
}

struct LevelsArgs {
    const double* wg_sumsq;
    const float* wg_peak;
    int chunks_per_piece, divisions;
    long long piece;
    int is_reference;
    TrackStats* st;
    double* rms;
    int* loud;
};
__device__ __forceinline__ void levels_body(const LevelsArgs& t, double threshold, double eps) {
    MGX_LDS;
    double* red = reinterpret_cast<double*>(mgx_smem);          // 64 doubles of reduction scratch
    double* sums = red + 64;                                     // [divisions]
    float* fred = reinterpret_cast<float*>(red + 52);
    float m = 0.f;
    for (int w = threadIdx.x; w < t.divisions * t.chunks_per_piece; w += blockDim.x) m = fmaxf(m, t.wg_peak[w]);
    const float pk = block_max<1024>(m, fred);
    if (threadIdx.x == 0) red[41] = (double)pk;
    __syncthreads();
    const double peak = red[41];
    double c = 1.0;
    if (t.is_reference && peak < threshold) c = fmax(eps, peak / threshold);     // dsp.py:98-99
    piece_sums_to_lds(t.wg_sumsq, t.chunks_per_piece, t.divisions, sums);
    double avg, match;
    int count;
    decide_loud<1024>(sums, t.divisions, t.piece, 1.0 / c, red, t.rms, t.loud, avg, match, count);
    if (threadIdx.x == 0) {
        TrackStats s;
        s.peak = peak;
        s.amplitude_c = c;
        s.average_rms = avg;
        s.match_rms = match;
        s.divisions = t.divisions;
        s.loud_count = count;
        s.piece = t.piece;
        *t.st = s;
    }
}
// one workgroup per track: grid = 1 (a single track) or 2 (target, reference)
__global__ __launch_bounds__(1024) void k_levels(LevelsArgs t0, LevelsArgs t1, double threshold, double eps) {
    levels_body(blockIdx.x == 0 ? t0 : t1, threshold, eps);
}

// mean over loud pieces and segments of |rfft|/F (match_frequencies.py:42), float64
// Stage 1 of a fixed-order two-stage sum: grid (bin tiles of 64, 2 planes, SPEC_SLICES); a
// workgroup = 64 bins x 16 lanes over its slice of the analysis workgroups.  Output
// part[z][plane][bins] (unscaled sums over the LOUD pieces' workgroups); the consumer adds
// the SPEC_SLICES slices and applies spectrum_scale().
constexpr int SPEC_SLICES = 8;
struct SpectraArgs {
    const float* wg_spec;
    const int* loud;
    int chunks_per_piece, nwg;
    double* part;
};
// grid (bin tiles of 64, 2 planes, SPEC_SLICES * tracks)
__global__ __launch_bounds__(1024) void k_average_spectra(SpectraArgs t0, SpectraArgs t1, int bins) {
    __shared__ double red[1024];
    const SpectraArgs& t = blockIdx.z < SPEC_SLICES ? t0 : t1;
    const int bin = blockIdx.x * 64 + (threadIdx.x & 63), lane = threadIdx.x >> 6;
    const int plane = blockIdx.y, z = blockIdx.z % SPEC_SLICES;
    const int per = (t.nwg + SPEC_SLICES - 1) / SPEC_SLICES;
    const int w0 = z * per, w1 = min(t.nwg, w0 + per);
    double s = 0.0;
    if (bin < bins) {
        for (int w = w0 + lane; w < w1; w += 16)
            if (t.loud[w / t.chunks_per_piece]) s += (double)t.wg_spec[((size_t)w * 2 + plane) * bins + bin];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (lane == 0 && bin < bins) {
        double acc = 0.0;
#pragma unroll
        for (int l = 0; l < 16; ++l) acc += red[l * 64 + (threadIdx.x & 63)];
        t.part[((size_t)z * 2 + plane) * bins + bin] = acc;
    }
}
__device__ __forceinline__ double spectrum_scale(const TrackStats* st, int segs_per_piece, int fft) {
    return 1.0 / ((double)st->loud_count * (double)segs_per_piece * (double)fft * st->amplitude_c);
}
__device__ __forceinline__ double spectrum_at(const double* part, int plane, int bins, int k) {
    double t = 0.0;
#pragma unroll
    for (int z = 0; z < SPEC_SLICES; ++z) t += part[((size_t)z * 2 + plane) * bins + k];
    return t;
}
// mean |rfft|/F over the loud pieces (match_frequencies.py:42) for the stage-level API
__global__ void k_finish_spectra(const double* part, const TrackStats* st, int segs_per_piece, int fft,
                                 double* avg /* [2][bins] */) {
    const int bins = fft / 2 + 1;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * bins) return;
    avg[i] = spectrum_at(part, i / bins, bins, i % bins) * spectrum_scale(st, segs_per_piece, fft);
}

// ---------------------------------------------------------------------------
// FIR design on the device (fir_plan.h): four small kernels, no host round trip
// ---------------------------------------------------------------------------
struct FirInputs {
    const double* part_t;        // target spectra partial sums   [SPEC_SLICES][2][bins]
    const double* part_r;        // reference
    const TrackStats* st_t;
    const TrackStats* st_r;
    int segs_t, segs_r;
    double eps;
};
__device__ __forceinline__ FirScratch fir_scratch(double* base, const FirPlanView& pl, int plane) {
    const size_t per = (size_t)3 * pl.bins + (size_t)3 * pl.nlog + pl.lw.anchors;
    double* p = base + (size_t)plane * per;
    FirScratch s;
    s.raw = p;
    s.m1 = s.raw + pl.bins;
    s.smooth = s.m1 + pl.bins;
    s.on_log = s.smooth + pl.bins;
    s.log_s = s.on_log + pl.nlog;
    s.m2 = s.log_s + pl.nlog;
    s.fit = s.m2 + pl.nlog;
    return s;
}
__device__ __forceinline__ void fir_solve(const SplineTables& sp, const double* y, double* m, Affine* sc) {
    using FD = FirDesign;
    const int tid = threadIdx.x;
    FD::phase_fwd_local(tid, sp, y, sc);
    __syncthreads();
    FD::Scan::scan_groups(sc, tid);
    __syncthreads();
    FD::Scan::scan_top(sc, tid);
    __syncthreads();
    FD::phase_fwd_apply(tid, sp, y, sc, m);
    __syncthreads();
    FD::phase_bwd_local(tid, sp, m, sc);
    __syncthreads();
    FD::Scan::scan_groups(sc, tid);
    __syncthreads();
    FD::Scan::scan_top(sc, tid);
    __syncthreads();
    FD::phase_bwd_apply(tid, sp, sc, m);
    __syncthreads();
    FD::phase_closure(tid, sp, m);
    __syncthreads();
}
// ---- the chain raw -> smooth as ONE dense operator ---------------------------------------------
// For a given Config, smooth = M * raw with a fixed (bins x bins) float64 matrix: splines and
// LOWESS (it = 0) are linear in their input and the pinned bins are rows of M.  M is built once
// per plan ON THE DEVICE by pushing unit vectors through the very kernels above/below
// (k_fir_unit_a, k_fir_lowess, k_fir_b, k_fir_gather); per pair the design is then k_fir_raw +
// one 34 MB matrix-vector product for both channels (k_fir_matvec) instead of ~90 us of serial
// single-workgroup scans.
__global__ __launch_bounds__(1024) void k_fir_unit_a(FirPlanView pl, double* scratch, int col0) {
    MGX_LDS;
    Affine* sc = reinterpret_cast<Affine*>(mgx_smem);
    const int tid = threadIdx.x, plane = blockIdx.x;
    FirScratch s = fir_scratch(scratch, pl, plane);
    for (int k = tid; k < pl.bins; k += 1024) s.raw[k] = k == col0 + plane ? 1.0 : 0.0;
    __syncthreads();
    fir_solve(pl.s1, s.raw, s.m1, sc);
    FirDesign::phase_eval(tid, pl.s1, s.raw, s.m1, s.on_log);
}
// ---- lowess_it > 0: LOWESS is no longer linear in the data, so the chain runs on the curve itself ----
// raw curve -> spline onto the log grid.  grid = 2 (mid, side)
__global__ __launch_bounds__(1024) void k_fir_direct_a(FirPlanView pl, double* scratch, const double* raw /* [2][bins] */) {
    MGX_LDS;
    Affine* sc = reinterpret_cast<Affine*>(mgx_smem);
    const int tid = threadIdx.x, plane = blockIdx.x;
    FirScratch s = fir_scratch(scratch, pl, plane);
    for (int k = tid; k < pl.bins; k += 1024) s.raw[k] = raw[(size_t)plane * pl.bins + k];
    __syncthreads();
    fir_solve(pl.s1, s.raw, s.m1, sc);
    FirDesign::phase_eval(tid, pl.s1, s.raw, s.m1, s.on_log);
}
// the k-th smallest (k from 0) of n non-negative doubles, by bisection on the bit pattern (which orders
// them): 63 counting passes, every thread of the 1024 gets the result.  red: 17 ints of LDS
__device__ __forceinline__ double block_select(const double* v, int n, int k, int* red) {
    const int tid = threadIdx.x;
    unsigned long long lo = 0ull, hi = 0x7ff0000000000000ull;
    while (lo < hi) {
        const unsigned long long mid = lo + ((hi - lo) >> 1);
        int c = 0;
        for (int q = tid; q < n; q += 1024) c += double_bits(v[q]) <= mid ? 1 : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
        if ((tid & 63) == 0) red[tid >> 6] = c;
        __syncthreads();
        if (tid == 0) {
            int t = 0;
            for (int w = 0; w < 16; ++w) t += red[w];
            red[16] = t;
        }
        __syncthreads();
        if (red[16] >= k + 1) hi = mid;
        else lo = mid + 1;
        __syncthreads();                      // red[16] is rewritten in the next pass
    }
    return bits_double(lo);
}
// LOWESS with robustness iterations (fir_plan.h: phase_lowess_fit_robust ...), one workgroup per channel;
// work: [2][2][nlog] doubles (robustness weights, residuals).  Leaves the anchors' fits of the last pass
// in s.fit, where k_fir_b picks them up.
__global__ __launch_bounds__(1024) void k_fir_lowess_robust(FirPlanView pl, double* scratch, double* work, int it) {
    __shared__ int red[17];
    const int tid = threadIdx.x, plane = blockIdx.x, n = pl.nlog;
    FirScratch s = fir_scratch(scratch, pl, plane);
    double* robust = work + (size_t)plane * 2 * n;
    double* resid = robust + n;
    FirDesign::phase_robust_init(tid, n, robust);
    __syncthreads();
    for (int pass = 0; pass <= it; ++pass) {
        FirDesign::phase_lowess_fit_robust(tid, pl.lw, s.on_log, robust, s.fit);
        __syncthreads();
        if (pass == it) break;
        FirDesign::phase_lowess_fill(tid, pl.lw, s.fit, s.log_s);
        __syncthreads();
        FirDesign::phase_residuals(tid, n, s.on_log, s.log_s, resid);
        __syncthreads();
        double median = block_select(resid, n, n / 2, red);                       // numpy.median
        if (!(n & 1)) median = 0.5 * (block_select(resid, n, n / 2 - 1, red) + median);
        FirDesign::phase_robust_weights(tid, n, resid, median, robust);
        __syncthreads();
    }
}
// M[i][col0 + c] = smooth of unit vector col0 + c, bin i
__global__ __launch_bounds__(256) void k_fir_gather(FirPlanView pl, double* scratch, int col0, int ncols, double* M) {
    const int c = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
    if (c >= ncols) return;
    const FirScratch s = fir_scratch(scratch, pl, c);
    M[(size_t)i * pl.bins + col0 + c] = s.smooth[i];
}
// raw matching curves of both channels (match_frequencies.py:93-94) + the level gain c0
struct CorrectionState;
__device__ void correction_reset(CorrectionState* cs, double gain);
__global__ __launch_bounds__(256) void k_fir_raw(FirPlanView pl, FirInputs in, double* raw /* [2][bins] */,
                                                 double* c0_out, CorrectionState* cs_init) {
    const int k = blockIdx.x * 256 + threadIdx.x, plane = blockIdx.y;
    const double c0 = in.st_r->match_rms / fmax(in.eps, in.st_t->match_rms);      // match_levels.py:106-111
    if (plane == 0 && k == 0) {
        *c0_out = c0;
        if (cs_init) correction_reset(cs_init, 1.0);       // stages.py:138-170 starts from gain 1
    }
    if (k >= pl.bins) return;
    const double sc_t = spectrum_scale(in.st_t, in.segs_t, pl.fft) * c0;          // stages.py:90-91
    const double sc_r = spectrum_scale(in.st_r, in.segs_r, pl.fft);
    const double at = spectrum_at(in.part_t, plane, pl.bins, k) * sc_t;
    const double ar = spectrum_at(in.part_r, plane, pl.bins, k) * sc_r;
    raw[(size_t)plane * pl.bins + k] = ar / fmax(pl.min_value, at);
}
// The piece decisions of match_levels.py:62-71,93-103 by ONE wave, without a barrier: sums[d] = sum of
// mid^2 of piece d (LDS); every lane returns the same average rms, match rms and loud count, and the loud
// flags go to `loud_out` (LDS).  Lane-strided loops and butterfly sums: a fixed order.
__device__ __forceinline__ void wave_decide(const double* sums, int divisions, long long piece, double inv_c,
                                            double* rms_out, int* loud_out, double& avg, double& match, int& count) {
    const int lane = threadIdx.x & 63;
    double acc = 0.0;
    for (int d = lane; d < divisions; d += 64) {
        const double r = sqrt(sums[d] / (double)piece) * inv_c;
        acc += r * r;
    }
    avg = sqrt(wave_sum(acc) / divisions);
    double lacc = 0.0, lcnt = 0.0;
    for (int d = lane; d < divisions; d += 64) {
        const double r = sqrt(sums[d] / (double)piece) * inv_c;
        const bool l = r >= avg;
        if (l) { lacc += r * r; lcnt += 1.0; }
        if (rms_out) rms_out[d] = r;
        if (loud_out) loud_out[d] = l ? 1 : 0;
    }
    const double cnt = wave_sum(lcnt);
    count = (int)cnt;
    match = sqrt(wave_sum(lacc) / cnt);
}

// ---- levels + loud-piece spectra + raw matching curve in ONE launch --------------------------------
// k_levels -> k_average_spectra -> k_fir_raw are three dependent launches of a few microseconds of
// work each; here every workgroup re-derives the (tiny) piece decisions of both tracks in its own LDS
// (one batch of loads, the two tracks decided side by side by one wave each), sums its tile of bins
// over the loud workgroup rows of both tracks and writes the raw curve (match_frequencies.py:93-94)
// directly.  Grid (bin tiles of 32, 2 planes) x 1024 threads = 32 bins x 32 row lanes; every sum runs in
// a fixed order.  Workgroup (0, 0) also leaves the TrackStats, the piece tables, the level gain c0
// (stages.py:80-88) and the reset correction state.
struct CurveTrack {
    LevelsArgs lv;
    const float* wg_spec;        // [nwg][2][bins]
    int nwg, segs_per_piece;
};
// LDS carve, in doubles: acc[1024] | scal[16] | sums[2][max_div] | ss[nwg_t + nwg_r] ; then ints loud[2][max_div]
// and floats pk[nwg_t + nwg_r]
__host__ __device__ inline size_t match_curve_lds_bytes(int max_div, int rows) {
    return ((size_t)1024 + 16 + 2 * (size_t)max_div + rows) * 8 + (2 * (size_t)max_div + rows + 4) * 4;
}
// TILE bins x ROWL = 1024 / TILE row lanes per workgroup.  32 x 32 unless 33 x 31 (one idle thread) saves a round of
// workgroups: the spectrum has 2^k + 1 bins, so tiles of 32 leave ONE bin for a last tile per channel, and at
// fft_size 16384 those two workgroups are numbers 513 and 514 on a chip that holds 256 at a time (29 -> 20 us).
template <int TILE>
__global__ __launch_bounds__(1024) void k_match_curve(CurveTrack tt, CurveTrack tr, int bins, int fft, int max_div,
                                                      double threshold, double eps, double curve_floor,
                                                      double* raw /* [2][bins] */, double* c0_out,
                                                      CorrectionState* cs_init, int* error) {
    warm_code(CODE_MATCH_CURVE);
    MGX_LDS;
    const int rows = tt.nwg + tr.nwg;
    double* acc = reinterpret_cast<double*>(mgx_smem);
    double* scal = acc + 1024;                                  // [k*4 + {amplitude_c, match, count, -}]
    double* sums = scal + 16;                                   // [2][max_div]
    double* ss = sums + 2 * max_div;                            // [rows] piece-chunk sums of mid^2, target rows first
    int* loud = reinterpret_cast<int*>(ss + rows);              // [2][max_div]
    float* pk = reinterpret_cast<float*>(loud + 2 * max_div);   // [rows]
    const bool writer = blockIdx.x == 0 && blockIdx.y == 0;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // The spectra of the first 16 ROWL workgroup rows of EACH track are asked for before anything else: what is summed
    // depends on the level decisions below, what is loaded does not, so the decisions (two LDS round trips and a
    // wave's worth of arithmetic) run while the loads are in flight.  Buffer views: ONE lane offset per track, the row
    // step is a scalar displacement, and rows past the end of a track read as zero through the range check.
    constexpr int ROWL = 1024 / TILE;
    const int b = threadIdx.x % TILE, row_lane = threadIdx.x / TILE, plane = blockIdx.y;      // (row_lane == ROWL: the idle thread)
    const int bin = row_lane < ROWL ? blockIdx.x * TILE + b : bins;
    const MemView vt = mem_view(tt.wg_spec, (long long)tt.nwg * 2 * bins * 4);
    const MemView vr = mem_view(tr.wg_spec, (long long)tr.nwg * 2 * bins * 4);
    // (a bin past the end reads from past the end of the view: zeros)
    const unsigned lane_off = bin < bins ? (unsigned)((((size_t)row_lane * 2 + plane) * bins + bin) * 4) : 0xfffffff0u;
    const unsigned row_step = (unsigned)((size_t)ROWL * 2 * bins * 4);
    double ssv[2] = {0.0, 0.0};
    float pkv[2] = {0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 2; ++u) {                                // (rows <= 2048: host_params / run_fir_design)
        const int w = threadIdx.x + 1024 * u;
        if (w < rows) {
            const bool second = w >= tt.nwg;
            const LevelsArgs& t = second ? tr.lv : tt.lv;
            const int i = second ? w - tt.nwg : w;
            ssv[u] = t.wg_sumsq[i];
            pkv[u] = t.wg_peak[i];
        }
    }
    float v0[2][16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const unsigned disp = (unsigned)u * row_step;
        v0[0][u] = ld_f1(vt, lane_off, disp);
        v0[1][u] = ld_f1(vr, lane_off, disp);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int w = threadIdx.x + 1024 * u;
        if (w < rows) {
            ss[w] = ssv[u];
            pk[w] = pkv[u];
        }
    }
    for (int w = threadIdx.x + 2048; w < rows; w += 1024) {       // (more rows than that: the plain way)
        const bool second = w >= tt.nwg;
        const LevelsArgs& t = second ? tr.lv : tt.lv;
        const int i = second ? w - tt.nwg : w;
        ss[w] = t.wg_sumsq[i];
        pk[w] = t.wg_peak[i];
    }
    lds_barrier();
    for (int p = threadIdx.x; p < tt.lv.divisions + tr.lv.divisions; p += 1024) {
        const bool second = p >= tt.lv.divisions;
        const LevelsArgs& t = second ? tr.lv : tt.lv;
        const int d = second ? p - tt.lv.divisions : p;
        const double* src = ss + (second ? tt.nwg : 0) + (size_t)d * t.chunks_per_piece;
        double sum = 0.0;
        for (int ch = 0; ch < t.chunks_per_piece; ++ch) sum += src[ch];
        sums[(second ? max_div : 0) + d] = sum;
    }
    lds_barrier();
    if (wave < 2) {                                             // wave 0: target, wave 1: reference
        const int k = wave;
        const LevelsArgs& t = k == 0 ? tt.lv : tr.lv;
        const float* p = pk + (k == 0 ? 0 : tt.nwg);
        float m = 0.f;
        for (int w = lane; w < (k == 0 ? tt.nwg : tr.nwg); w += 64) m = fmaxf(m, p[w]);
        const double peak = (double)wave_max(m);
        double c = 1.0;
        if (t.is_reference && peak < threshold) c = fmax(eps, peak / threshold);     // dsp.py:98-99
        double avg, match;
        int count;
        wave_decide(sums + k * max_div, t.divisions, t.piece, 1.0 / c, writer ? t.rms : nullptr, loud + k * max_div, avg,
                    match, count);
        if (lane == 0) {
            scal[k * 4 + 0] = c;
            scal[k * 4 + 1] = match;
            scal[k * 4 + 2] = (double)count;
        }
        if (writer) {
            for (int d = lane; d < t.divisions; d += 64) t.loud[d] = loud[k * max_div + d];
            if (lane == 0) {
                TrackStats st;
                st.peak = peak;
                st.amplitude_c = c;
                st.average_rms = avg;
                st.match_rms = match;
                st.divisions = t.divisions;
                st.loud_count = count;
                st.piece = t.piece;
                *t.st = st;
                // A NaN or an infinity among the samples: no piece is "loud" (every comparison with NaN fails) or
                // the loud pieces' RMS is not a number.  The reference stops there (match_frequencies.py:42 is
                // handed an empty selection); here the handle's error word makes the next blocking call fail.
                if (error && (count == 0 || !(fabs(match) < 1.0e300))) error[DEVICE_ERROR_SLOT_INPUT] = 1;
            }
        }
    }
    lds_barrier();
    const double c0 = scal[4 + 1] / fmax(eps, scal[1]);         // match_levels.py:106-111
    if (writer && threadIdx.x == 0) {
        *c0_out = c0;
        if (cs_init) correction_reset(cs_init, 1.0);            // stages.py:138-170 starts from gain 1
    }
    // sixteen rows of EACH track per thread and batch: one round trip for a pair of 8-minute tracks
    double sacc[2] = {0.0, 0.0};
    {
        const int longest = max(tt.nwg, tr.nwg);
        // workgroup row -> piece without a division per row: floor(w * ceil(2^32 / d) / 2^32) = w / d for w, d < 2^16
        const unsigned magic[2] = {(unsigned)((0x100000000ull + tt.lv.chunks_per_piece - 1) / tt.lv.chunks_per_piece),
                                   (unsigned)((0x100000000ull + tr.lv.chunks_per_piece - 1) / tr.lv.chunks_per_piece)};
#pragma unroll 1
        for (int w0 = 0; w0 < longest; w0 += ROWL * 16) {
            if (opaque(w0) > 0) {                               // (the first batch is in flight since the top; opaque: one copy of the sums)
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const unsigned disp = (unsigned)(w0 / ROWL + u) * row_step;
                    v0[0][u] = ld_f1(vt, lane_off, disp);
                    v0[1][u] = ld_f1(vr, lane_off, disp);
                }
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const CurveTrack& t = k == 0 ? tt : tr;
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int w = w0 + row_lane + ROWL * u;
                    const int piece = t.lv.chunks_per_piece == 1 ? w : (int)__umulhi((unsigned)w, magic[k]);
                    const bool on = w < t.nwg && loud[k * max_div + (w < t.nwg ? piece : 0)] != 0;
                    sacc[k] += on ? (double)v0[k][u] : 0.0;
                }
            }
        }
    }
    double level[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const CurveTrack& t = k == 0 ? tt : tr;
        acc[threadIdx.x] = sacc[k];
        __syncthreads();
        double total = 0.0;
        if (row_lane == 0) {
#pragma unroll 8
            for (int l = 0; l < ROWL; ++l) total += acc[l * TILE + b];
        }
        // mean over loud pieces and segments of |rfft|/F of the normalised track (match_frequencies.py:42)
        level[k] = total / (scal[k * 4 + 2] * (double)t.segs_per_piece * (double)fft * scal[k * 4 + 0]);
        __syncthreads();
    }
    if (row_lane == 0 && bin < bins)
        raw[(size_t)plane * bins + bin] = level[1] / fmax(curve_floor, level[0] * c0);   // stages.py:90-91 on the target
}

// The operator is numerically banded: LOWESS looks at 3.75 % of the log-frequency grid and the
// splines' influence decays geometrically, so outside a window around the diagonal (about a third of
// the bin index wide) every entry is below 1e-18 of the row's largest.  k_fir_band records that
// window per row once per plan; the product then reads only it (a sixth of the matrix).
constexpr double BAND_EPS = 1e-18;
__global__ __launch_bounds__(256) void k_fir_band(const double* M, int bins, int2* band) {
    __shared__ double dred[4];
    __shared__ int ired[2][4];
    const int row = blockIdx.x, tid = threadIdx.x;
    const double* m = M + (size_t)row * bins;
    double mx = 0.0;
    for (int j = tid; j < bins; j += 256) mx = fmax(mx, fabs(m[j]));
    mx = wave_max_f64(mx);
    if ((tid & 63) == 0) dred[tid >> 6] = mx;
    __syncthreads();
    const double cut = fmax(fmax(dred[0], dred[1]), fmax(dred[2], dred[3])) * BAND_EPS;
    int lo = bins, hi = 0;
    for (int j = tid; j < bins; j += 256)
        if (fabs(m[j]) > cut) { lo = min(lo, j); hi = max(hi, j + 1); }
    for (int o = 32; o > 0; o >>= 1) { lo = min(lo, __shfl_xor(lo, o)); hi = max(hi, __shfl_xor(hi, o)); }
    if ((tid & 63) == 0) { ired[0][tid >> 6] = lo; ired[1][tid >> 6] = hi; }
    __syncthreads();
    if (tid == 0) {
        lo = min(min(ired[0][0], ired[0][1]), min(ired[0][2], ired[0][3]));
        hi = max(max(ired[1][0], ired[1][1]), max(ired[1][2], ired[1][3]));
        band[row] = lo < hi ? make_int2(lo, hi) : make_int2(0, 0);
    }
}
// smooth[plane][i] = sum_j M[i][j] raw[plane][j] over the row's window: one 256-thread workgroup per
// row, both channels per pass, ten loads per thread in flight
__global__ __launch_bounds__(256) void k_fir_matvec(FirPlanView pl, const double* M, const int2* band,
                                                    const double* raw, double* scratch) {
    __shared__ double red[2][4];
    const int row = blockIdx.x, tid = threadIdx.x;
    const double* m = M + (size_t)row * pl.bins;
    const double* r0 = raw;
    const double* r1 = raw + pl.bins;
    const int first = band[row].x, last = band[row].y;
    double a0 = 0.0, a1 = 0.0;
    for (int j0 = first + tid; j0 < last; j0 += 256 * 10) {
        double v[10];
#pragma unroll
        for (int u = 0; u < 10; ++u) {
            const int j = j0 + 256 * u;
            v[u] = j < last ? m[j] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 10; ++u) {
            const int j = j0 + 256 * u;
            if (j < last) {
                a0 = fma(v[u], r0[j], a0);
                a1 = fma(v[u], r1[j], a1);
            }
        }
    }
    a0 = wave_sum(a0);
    a1 = wave_sum(a1);
    if ((tid & 63) == 0) { red[0][tid >> 6] = a0; red[1][tid >> 6] = a1; }
    __syncthreads();
    if (tid == 0) {
        fir_scratch(scratch, pl, 0).smooth[row] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        fir_scratch(scratch, pl, 1).smooth[row] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

// ---- the same chain in TWO factors, for fft_size >= 16384 ----------------------------------------------
// raw -> smooth passes through the LOWESS fits at the anchors (match_frequencies.py:62-64 with lowess_delta: only
// ~1/delta = 1000 points of the log grid get a regression, the rest is interpolated between them), so
//     smooth = B * (A * raw),   A = [anchors x bins]: spline onto the log grid + the anchors' regressions,
//                               B = [bins x anchors]: fill between anchors + spline back + pinned bins.
// Both are banded like M (A's row of anchor a covers the linear bins under its 3.75 % neighbourhood, B's row of bin
// i the handful of anchors around it) and stored PACKED, row after row, only the window k_fir_band found:
// 3.4 MB at fft_size 16384 where M is 537 MB (of which the product read 83 MB per pair, 28 us), a few MB more at
// 65536 where M would be 8.6 GB.  Built like M, by pushing unit vectors through the chain's own kernels.
// One 16-byte descriptor per row: its window [first, last) and where it starts in the packed array (one load, not a
// chain of two, in front of a row's data).
struct FactorRow {
    int first, last;
    long long off;
};
// unit anchor `col0 + plane` as the fits, nothing on the raw side (phase_pin reads raw[1])
__global__ __launch_bounds__(256) void k_fir_unit_fit(FirPlanView pl, double* scratch, int col0) {
    const int plane = blockIdx.x;
    FirScratch s = fir_scratch(scratch, pl, plane);
    for (int a = threadIdx.x; a < pl.lw.anchors; a += 256) s.fit[a] = a == col0 + plane ? 1.0 : 0.0;
    if (threadIdx.x == 0) s.raw[1] = 0.0;
}
// dense[i][col0 + c] = (fits | smooth curve) of plane c at i; grid = (ceil(ncols / 256), rows)
__global__ __launch_bounds__(256) void k_fir_gather_plane(FirPlanView pl, double* scratch, int col0, int ncols, int stride,
                                                          int from_fit, double* dense) {
    const int c = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
    if (c >= ncols) return;
    const FirScratch s = fir_scratch(scratch, pl, c);
    dense[(size_t)i * stride + col0 + c] = from_fit ? s.fit[i] : s.smooth[i];
}
// packed[off[row] + j - band[row].x] = dense[row][j] over the row's window; grid = rows
__global__ __launch_bounds__(256) void k_fir_pack(const double* dense, int stride, const FactorRow* rows, double* packed) {
    const int row = blockIdx.x;
    const int first = rows[row].first, last = rows[row].last;
    const double* src = dense + (size_t)row * stride;
    double* dst = packed + rows[row].off - first;
    for (int j = first + threadIdx.x; j < last; j += 256) dst[j] = src[j];
}
// fit[plane][a] = sum_j A[a][j] raw[plane][j]: one 256-thread workgroup per anchor, both channels per pass, the whole
// window in flight at once (the widest is 0.31 * bins + 62 columns: eleven loads per thread at fft_size 16384);
// grid = anchors.  (One WAVE per anchor, the first version, walked the wide windows in five dependent batches:
// 20 us for the two factors against the dense product's 28, profiles/r05_p_fir_factored_first_version.txt.)
__global__ __launch_bounds__(256) void k_fir_apply_a(FirPlanView pl, const double* A, const FactorRow* rows, const double* raw,
                                                     double* scratch) {
    __shared__ double red[2][4];
    const int a = blockIdx.x, tid = threadIdx.x;
    const FactorRow row = rows[a];
    const int first = row.first, last = row.last;
    const double* m = A + row.off - first;
    const double* r0 = raw;
    const double* r1 = raw + pl.bins;
    double a0 = 0.0, a1 = 0.0;
    for (int j0 = first + tid; j0 < last; j0 += 256 * 12) {
        double v[12], x0[12], x1[12];
#pragma unroll
        for (int u = 0; u < 12; ++u) {
            const int j = j0 + 256 * u;
            const bool in = j < last;
            v[u] = in ? m[j] : 0.0;
            x0[u] = in ? r0[j] : 0.0;
            x1[u] = in ? r1[j] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 12; ++u) {
            a0 = fma(v[u], x0[u], a0);
            a1 = fma(v[u], x1[u], a1);
        }
    }
    a0 = wave_sum(a0);
    a1 = wave_sum(a1);
    if ((tid & 63) == 0) { red[0][tid >> 6] = a0; red[1][tid >> 6] = a1; }
    __syncthreads();
    if (tid == 0) {
        fir_scratch(scratch, pl, 0).fit[a] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        fir_scratch(scratch, pl, 1).fit[a] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}
// smooth[plane][i] = sum_a B[i][a] fit[plane][a] (a handful of terms: one thread per bin), bins 0 and 1 pinned
// (match_frequencies.py:72-73); grid = ceil(bins / 256)
__global__ __launch_bounds__(256) void k_fir_apply_b(FirPlanView pl, const double* B, const FactorRow* rows, const double* raw,
                                                     double* scratch) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= pl.bins) return;
    const FirScratch s0 = fir_scratch(scratch, pl, 0), s1 = fir_scratch(scratch, pl, 1);
    const FactorRow row = rows[i];
    const int first = row.first, last = row.last;
    const double* m = B + row.off - first;
    double a0 = 0.0, a1 = 0.0;
    for (int a = first; a < last; a += 8) {                      // (windows are 4 - 10 anchors wide: one batch, two at most)
        double v[8], f0[8], f1[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool in = a + u < last;
            v[u] = in ? m[a + u] : 0.0;
            f0[u] = in ? s0.fit[a + u] : 0.0;
            f1[u] = in ? s1.fit[a + u] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            a0 = fma(v[u], f0[u], a0);
            a1 = fma(v[u], f1[u], a1);
        }
    }
    if (i == 0) a0 = a1 = 0.0;
    if (i == 1) { a0 = raw[1]; a1 = raw[pl.bins + 1]; }
    s0.smooth[i] = a0;
    s1.smooth[i] = a1;
}

// fft_size 8 .. 32 (small_fft_kernels.h): taps[i] = hann[i] * irfft(smooth)[(i + F/2) mod F] by the plain cosine sum (fir_plan.h, phase_taps); grid = 2
__global__ __launch_bounds__(1024) void k_fir_taps_direct(FirPlanView pl, const double* scratch, float* taps /* [2][F] */) {
    const int plane = blockIdx.x;
    const FirScratch s = fir_scratch(const_cast<double*>(scratch), pl, plane);
    FirDesign::phase_taps(threadIdx.x, pl, s.smooth, taps + (size_t)plane * pl.fft, nullptr);
}

// LOWESS regressions: one wave per anchor.  grid = (ceil(anchors/16), 2)
__global__ __launch_bounds__(1024) void k_fir_lowess(FirPlanView pl, double* scratch) {
    const int plane = blockIdx.y;
    const FirScratch s = fir_scratch(scratch, pl, plane);
    const int a = blockIdx.x * 16 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (a >= pl.lw.anchors) return;
    const double* p = pl.lw.p + (size_t)a * pl.lw.k;
    const double* y = s.on_log + pl.lw.lo[a];
    double acc = 0.0;
    for (int j = lane; j < pl.lw.k; j += 64) acc = fma(p[j], y[j], acc);
    acc = wave_sum(acc);
    if (lane == 0) s.fit[a] = acc;
}
// skipped points -> spline back onto the linear grid -> pinned bins.  grid = 2
__global__ __launch_bounds__(1024) void k_fir_b(FirPlanView pl, double* scratch) {
    MGX_LDS;
    Affine* sc = reinterpret_cast<Affine*>(mgx_smem);
    const int tid = threadIdx.x, plane = blockIdx.x;
    FirScratch s = fir_scratch(scratch, pl, plane);
    FirDesign::phase_lowess_fill(tid, pl.lw, s.fit, s.log_s);
    __syncthreads();
    fir_solve(pl.s2, s.log_s, s.m2, sc);
    FirDesign::phase_eval(tid, pl.s2, s.log_s, s.m2, s.smooth);
    __syncthreads();
    FirDesign::phase_pin(tid, s);
}
// irfft + ifftshift + Hann (match_frequencies.py:98-99) as a cosine sum in float64.  The spectrum is real
// and even, so h0[m] = (H[0] + (-1)^m H[F/2] + 2 S(m)) / F with S(m) = sum_{k=1}^{F/2-1} H[k] cos(2 pi k m / F),
// and S has two symmetries: S(F - m) = S(m), and with the even-k and odd-k halves E(m), O(m) of the sum,
// S(m) = E + O while S(F/2 - m) = E - O.  One pair (E, O) for m in [0, F/4] therefore gives four taps:
// a quarter of the F^2 / 2 terms of the plain sum (at 16 k taps 100 us of this kernel).
// grid = (ceil((F/4 + 1) / TAP_ROWS), 2); a workgroup takes TAP_ROWS values of m, its TAP_SLICES = 1024 /
// TAP_ROWS groups of lanes each sum a slice of the bins.  cos(2 pi k m / F) for the consecutive k of a
// slice comes from a rotation: start and step are exact table values, the steps in between cost four
// float64 operations each and add ~1e-14 of error over a slice -- no cosine table in LDS (filling 32 KB of
// it per workgroup was most of this kernel's time) and no gather through the L2.
constexpr int TAP_ROWS = 8, TAP_SLICES = 1024 / TAP_ROWS;
__global__ __launch_bounds__(1024) void k_fir_taps(FirPlanView pl, const double* scratch, float* taps /* [2][F] */) {
    MGX_LDS;
    double* sm = reinterpret_cast<double*>(mgx_smem);       // [bins]
    double* red = sm + pl.bins;                             // [2][1024]: even-k and odd-k partial sums
    const int plane = blockIdx.y, f = pl.fft, half = f / 2, quarter = f / 4;
    const FirScratch s = fir_scratch(const_cast<double*>(scratch), pl, plane);
    const int row = threadIdx.x % TAP_ROWS, slice = threadIdx.x / TAP_ROWS;
    const int mm = min(blockIdx.x * TAP_ROWS + row, quarter);       // (rows past F/4 repeat it: same stores)
    // bins 1 .. half-1 in slices of an even number of bins, so that every slice starts on an odd bin
    int per = (half - 1 + TAP_SLICES - 1) / TAP_SLICES;
    per += per & 1;
    const int k0 = 1 + slice * per, k1 = min(half, k0 + per);
    // asked for before the barrier: four table look-ups per thread
    const int i0 = (int)(((long long)k0 * mm) & (f - 1));
    double c = pl.cos_table[i0], sn = pl.cos_table[(i0 - quarter) & (f - 1)];        // sin x = cos(x - pi/2)
    const double dc = pl.cos_table[mm], ds = pl.cos_table[(mm - quarter) & (f - 1)];
    for (int k = threadIdx.x; k < pl.bins; k += 1024) sm[k] = s.smooth[k];
    __syncthreads();
    double odd = 0.0, even = 0.0;
    for (int k = k0; k < k1; k += 2) {                      // k odd, k + 1 even
        odd = fma(sm[k], c, odd);
        double cn = fma(c, dc, -sn * ds);
        sn = fma(sn, dc, c * ds);
        c = cn;
        if (k + 1 < k1) even = fma(sm[k + 1], c, even);
        cn = fma(c, dc, -sn * ds);
        sn = fma(sn, dc, c * ds);
        c = cn;
    }
    red[threadIdx.x] = even;
    red[1024 + threadIdx.x] = odd;
    __syncthreads();
    // TAP_ROWS x 2 sums of TAP_SLICES partials each: one wave per (row, parity)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave < 2 * TAP_ROWS) {
        const int r = wave >> 1, parity = wave & 1;
        double t = 0.0;
        for (int l = lane; l < TAP_SLICES; l += 64) t += red[parity * 1024 + l * TAP_ROWS + r];
        t = wave_sum(t);
        if (lane == 0) red[2048 + wave] = t;
    }
    __syncthreads();
    if (threadIdx.x < 4 * TAP_ROWS) {
        const int r = threadIdx.x >> 2, which = threadIdx.x & 3;
        const int m0 = min(blockIdx.x * TAP_ROWS + r, quarter);
        const double e = red[2048 + 2 * r], o = red[2048 + 2 * r + 1];
        // which: 0 -> m0, 1 -> F - m0 (E + O);  2 -> F/2 - m0, 3 -> F/2 + m0 (E - O)
        const int m = (which == 0 ? m0 : which == 1 ? f - m0 : which == 2 ? half - m0 : half + m0) & (f - 1);
        const double sum = which < 2 ? e + o : e - o;
        const int i = (m + half) & (f - 1);                 // ifftshift: tap i holds h0[(i + F/2) mod F]
        const double v = (sm[0] + ((m & 1) ? -sm[half] : sm[half]) + 2.0 * sum) / f * pl.hann[i];
        taps[(size_t)plane * f + i] = (float)v;
    }
}

// The same taps by a transform, for fft_size >= 8192 (below that the sum above is one 8 us launch and wins).
// numpy.fft.irfft of the real, even spectrum H[0..F/2] as ONE complex inverse DFT of N = F/2 points:
//     Z[k] = (H[k] + H[N-k]) + i w^k (H[k] - H[N-k]),  w = exp(2 pi i / F);   z = IDFT_N(Z) / F;
//     h0[2n] = Re z[n],  h0[2n+1] = Im z[n]
// (the even and odd output samples are the real and imaginary parts of one half-length transform), then
// ifftshift and the Hann window as above.  A workgroup is one CU, and one CU needs 32 us for 8192 points in
// float64 however the passes are arranged (measured: instruction issue, not LDS or its banks), so the transform
// is split by decimation in time over R = 8 workgroups per channel:
//   k_fir_taps_sub<LOG2M>   grid (R, 2): workgroup rho transforms Z[R m + rho], m < M = N/R, in LDS -- Z goes in
//       bit-reversed, then passes that do TWO radix-2 stages in registers (four points per group, a barrier per
//       pass; one plain radix-2 stage first when log2 M is odd); twiddles from a quarter-wave table in LDS
//       (M/4 + 1 exact values of cos_table, the rest by symmetry), so no pass waits on global memory;
//   k_fir_taps_combine      grid (N/256, 2): z[n] = sum_rho w_N^(rho n) A_rho[n mod M] in a fixed order, scale,
//       window, store the two taps of z[n].
// O(F log F) instead of the cosine sum's O(F^2 / 8): 13 us against 47 at 16384 taps; float64 throughout,
// twiddles are table values (cos_table holds cos(2 pi j / F)).  M <= 2048 (32768 taps) is 38 KB of LDS.
// (The bit-reversed store puts a wavefront's 64 consecutive m at a stride of M/64 points: one LDS bank.  A spare
// point after every M/64 turns that stride odd -- TapFft::at -- and costs the later passes nothing.)
constexpr int TAP_SPLIT = 8, TAP_TRANSFORM_FROM = 8192;
__host__ __device__ inline size_t fir_taps_sub_lds_bytes(int m) { return ((size_t)m + 64) * 16 + ((size_t)m / 4 + 1) * 8; }
template <int LOG2M>
struct TapFft {
    static constexpr int M = 1 << LOG2M, PAD_SHIFT = LOG2M >= 8 ? LOG2M - 6 : 30;
    static constexpr int T = M / 4 < 64 ? 64 : (M / 4 > 1024 ? 1024 : M / 4);         // threads: a group of four points each
    static __device__ __forceinline__ int at(int i) { return i + (i >> PAD_SHIFT); }
};
template <int M>
__device__ __forceinline__ double2 quarter_wave_twiddle(const double* q, int t) {      // exp(+2 pi i t / M), 0 <= t < M/2
    const int d = t - M / 4;
    return make_double2(d <= 0 ? q[t] : -q[M / 2 - t], q[d < 0 ? -d : d]);
}
__device__ __forceinline__ void dit_butterfly(double2& x, double2& y, double2 w) {
    const double yr = fma(y.x, w.x, -y.y * w.y), yi = fma(y.x, w.y, y.y * w.x);
    y = make_double2(x.x - yr, x.y - yi);
    x = make_double2(x.x + yr, x.y + yi);
}
template <int LOG2M>
__global__ __launch_bounds__(TapFft<LOG2M>::T) void k_fir_taps_sub(FirPlanView pl, const double* scratch,
                                                                   double2* sub /* [2][R][M] */) {
    MGX_LDS;
    using P = TapFft<LOG2M>;
    constexpr int M = P::M, T = P::T;
    double2* z = reinterpret_cast<double2*>(mgx_smem);       // [M + 64], indexed through P::at
    double* q = reinterpret_cast<double*>(z + M + 64);       // [M/4 + 1]: cos(2 pi j / M)
    const int plane = blockIdx.y, rho = blockIdx.x, r = gridDim.x, n = M * r, f = 2 * n;
    const FirScratch s = fir_scratch(const_cast<double*>(scratch), pl, plane);
    for (int j = threadIdx.x; j <= M / 4; j += T) q[j] = pl.cos_table[2 * r * j];
#pragma unroll
    for (int m0 = 0; m0 < M; m0 += T) {
        const int m = m0 + threadIdx.x;
        if (M >= T || m < M) {
            const int k = r * m + rho;
            const double a = s.smooth[k], b = s.smooth[n - k];
            const double c = pl.cos_table[k], sn = pl.cos_table[(k - f / 4) & (f - 1)];     // w^k = c + i sn
            const double sum = a + b, d = a - b;
            z[P::at((int)(__brev((unsigned)m) >> (32 - LOG2M)))] = make_double2(sum - sn * d, c * d);
        }
    }
    __syncthreads();
    int h = 1;
    if (LOG2M & 1) {
#pragma unroll
        for (int b0 = 0; b0 < M / 2; b0 += T) {
            const int b = b0 + threadIdx.x;
            if (M / 2 >= T || b < M / 2) {
                const int i0 = P::at(2 * b), i1 = P::at(2 * b + 1);
                double2 x = z[i0], y = z[i1];
                dit_butterfly(x, y, make_double2(1.0, 0.0));
                z[i0] = x, z[i1] = y;
            }
        }
        __syncthreads();
        h = 2;
    }
#pragma unroll 1
    for (; h < M; h <<= 2) {
#pragma unroll
        for (int b0 = 0; b0 < M / 4; b0 += T) {
            const int b = b0 + threadIdx.x;
            if (M / 4 >= T || b < M / 4) {
                const int rr = b & (h - 1), j = ((b - rr) << 2) + rr;
                const int ta = rr * (M / 2 / h), tb = ta >> 1;
                const int i0 = P::at(j), i1 = P::at(j + h), i2 = P::at(j + 2 * h), i3 = P::at(j + 3 * h);
                double2 a0 = z[i0], a1 = z[i1], a2 = z[i2], a3 = z[i3];
                const double2 wa = quarter_wave_twiddle<M>(q, ta);
                dit_butterfly(a0, a1, wa);
                dit_butterfly(a2, a3, wa);
                dit_butterfly(a0, a2, quarter_wave_twiddle<M>(q, tb));
                dit_butterfly(a1, a3, quarter_wave_twiddle<M>(q, tb + M / 4));
                z[i0] = a0, z[i1] = a1, z[i2] = a2, z[i3] = a3;
            }
        }
        __syncthreads();
    }
    double2* out = sub + ((size_t)plane * r + rho) * M;
#pragma unroll
    for (int i0 = 0; i0 < M; i0 += T) {
        const int i = i0 + threadIdx.x;
        if (M >= T || i < M) out[i] = z[P::at(i)];
    }
}
__global__ __launch_bounds__(256) void k_fir_taps_combine(FirPlanView pl, const double2* sub /* [2][R][M] */, int r,
                                                          float* taps /* [2][F] */) {
    const int plane = blockIdx.y, f = pl.fft, n_all = f / 2, m_len = n_all / r;
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= n_all) return;
    const double2* a = sub + (size_t)plane * n_all + (n & (m_len - 1));
    double zr = 0.0, zi = 0.0;
    for (int rho = 0; rho < r; ++rho) {
        const int e = (2 * rho * n) & (f - 1);                // w_N^(rho n) = exp(2 pi i (2 rho n) / F)
        const double c = pl.cos_table[e], sn = pl.cos_table[(e - f / 4) & (f - 1)];
        const double2 v = a[(size_t)rho * m_len];
        zr += fma(v.x, c, -v.y * sn);
        zi += fma(v.x, sn, v.y * c);
    }
    const double inv = 1.0 / (double)f;
    const int i = (2 * n + n_all) & (f - 1);                  // ifftshift: tap i holds h0[(i + F/2) mod F]; i is even
    float2 t;
    t.x = (float)(zr * inv * pl.hann[i]);
    t.y = (float)(zi * inv * pl.hann[i + 1]);
    *reinterpret_cast<float2*>(taps + (size_t)plane * f + i) = t;
}

#ifdef MGX_TAIL_TRACE      // experiments: 100 MHz timestamps of the phases (tools/tail_trace.py)
__device__ unsigned long long g_tail_trace[160 * 32];
__device__ unsigned long long g_round_trace[8];
#define TAIL_STAMP(slot) do { if (threadIdx.x == 0 && blockIdx.x < 160) g_tail_trace[blockIdx.x * 32 + (slot)] = wall_clock64(); } while (0)
#define ROUND_STAMP(slot) do { if (threadIdx.x == 0) g_round_trace[slot] = wall_clock64(); } while (0)
#else
#define TAIL_STAMP(slot) do {} while (0)
#define ROUND_STAMP(slot) do {} while (0)
#endif
// ---------------------------------------------------------------------------
// level correction (stages.py:138-170)
// ---------------------------------------------------------------------------
struct CorrectionState {
    double gain;              // product of the coefficients so far
    double coeffs[16];
    double result_peak;       // max |gain * y|
    double normalize_c;       // stages.py:186-191
    int limiter_active;
    int steps_done;
};

// partial[d*chunks + ch] = sum over the chunk of clip(gain*mid, -1, 1)^2
__global__ __launch_bounds__(256) void k_clipped_sumsq(const float* mid, long long piece, int chunks,
                                                       const double* gain_ptr, double gain_mul,
                                                       double* partial) {
    __shared__ double scratch[4];
    const int d = blockIdx.x / chunks, ch = blockIdx.x % chunks;
    const long long len = (piece + chunks - 1) / chunks;
    const long long b = (long long)d * piece + ch * len;
    const long long e = min((long long)(d + 1) * piece, b + len);
    const double g = (gain_ptr ? *gain_ptr : 1.0) * gain_mul;
    double acc = 0.0;
    // float64 product then clip: the reference clips the float64 mid (dsp.py:109-110)
    long long i = b + threadIdx.x;
    for (; i + 3 * 256 < e; i += 4 * 256) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = mid[i + u * 256];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const double c = fmin(fmax((double)v[u] * g, -1.0), 1.0);
            acc = fma(c, c, acc);
        }
    }
    for (; i < e; i += 256) {
        const double c = fmin(fmax((double)mid[i] * g, -1.0), 1.0);
        acc = fma(c, c, acc);
    }
    const double s = block_sum<256>(acc, scratch);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// one round of stages.py:149-168 on the partial sums
__global__ __launch_bounds__(1024) void k_correction_step(const double* partial, int chunks, int divisions,
                                                          long long piece, const double* reference_match_rms,
                                                          double eps, CorrectionState* cs) {
    MGX_LDS;
    double* red = reinterpret_cast<double*>(mgx_smem);
    double* sums = red + 64;
    piece_sums_to_lds(partial, chunks, divisions, sums);
    double avg, match;
    int count;
    decide_loud<1024>(sums, divisions, piece, 1.0, red, nullptr, nullptr, avg, match, count);
    if (threadIdx.x == 0) {
        const double c = *reference_match_rms / fmax(eps, match);      // match_levels.py:106-111
        if (cs->steps_done < 16) cs->coeffs[cs->steps_done] = c;
        cs->steps_done += 1;
        cs->gain *= c;
    }
}

// One round of stages.py:149-168 in ONE launch: every workgroup sums its chunk of
// clip(gain*mid)^2, and the workgroup that arrives last at the ticket counter takes the decision
// (loud pieces, coefficient, accumulated gain) -- split-K style "last arriver combines"
// (MI355X_MICROARCH.md, fanin / splitk-seam): partials are published write-through (sc1) and drained
// before the ticket, the last arriver acquires before reading them.  With `final_peaks` the same
// workgroup also derives the peak / limiter early-out / normalisation scalars (k_finalize_scalars).
// Band bookkeeping of one workgroup's chunk (k_correction_round).  Every correction coefficient is
// close to 1 (it is the ratio of two loudness estimates of nearly the same signal), so the
// accumulated gain g stays inside [BAND_G_LO, BAND_G_HI].  For such g a sample with
// |m| <= 1/BAND_G_HI is never clipped (contributes g^2 m^2), one with |m| > 1/BAND_G_LO always is
// (contributes 1), and only the few samples in between -- the band -- need to be looked at again.
// Round 0 streams the whole mid plane once, evaluates its own sum directly AND leaves
// {sum of m^2 of the never-clipped, count of the always-clipped, the band's values compacted per
// wave}; later rounds read just that (a few MB instead of 85) unless the gain has left the range,
// in which case they stream the plane again.  The split is exact: sum min(g^2 m^2, 1) is the same
// number either way, up to float64 summation order.
constexpr double BAND_G_LO = 0.7, BAND_G_HI = 1.5;
constexpr int BAND_SLACK = 2048;             // floats of padding per workgroup in the band buffer
// float32 thresholds on |m|, each rounded towards the inside of the band: |m| <= never implies
// |m| <= 1/BAND_G_HI exactly, |m| >= always implies |m| > 1/BAND_G_LO
__device__ __forceinline__ float band_threshold_never() {
    const float t = (float)(1.0 / BAND_G_HI);
    return (double)t <= 1.0 / BAND_G_HI ? t : __uint_as_float(__float_as_uint(t) - 1u);
}
__device__ __forceinline__ float band_threshold_always() {
    const float t = (float)(1.0 / BAND_G_LO);
    return (double)t > 1.0 / BAND_G_LO ? t : __uint_as_float(__float_as_uint(t) + 1u);
}
struct BandInfo {
    double unclipped_sumsq;                  // A: sum of m^2 over |m| <= 1/BAND_G_HI
    double clipped_count;                    // C: samples with |m| > 1/BAND_G_LO
    int count[4];                            // band samples compacted by each of the four waves
    int pad[2];
};
// frames [b, e) of chunk `ch` of piece `d`, and where the four per-wave band lists of that chunk start:
// a region of (e - b) + BAND_SLACK floats per chunk, a quarter of it (each wave sees a quarter of the
// chunk's samples, give or take the scalar head and tail) per wave
struct BandChunk {
    long long b, e, wave_cap;
    float* lists;
};
__device__ __forceinline__ BandChunk band_chunk(float* band, long long piece, int chunks, int d, int ch) {
    BandChunk c;
    const long long len = (piece + chunks - 1) / chunks;
    c.b = (long long)d * piece + ch * len;
    c.e = min((long long)(d + 1) * piece, c.b + len);
    c.wave_cap = (c.e - c.b + 3) / 4 + BAND_SLACK / 4 - 4;
    c.lists = band + c.b + ((long long)d * chunks + ch) * BAND_SLACK;
    return c;
}
struct RoundArgs {
    const float* mid;
    long long piece;
    int chunks, divisions;
    double* partial;            // [divisions][chunks]
    unsigned* arrivals;         // [1 + divisions] counters, zero between launches: [0] pieces done, [1+d] chunks of piece d
    const double* reference_match_rms;
    double eps, threshold;
    CorrectionState* cs;
    const float* final_peaks;   // per-pair peaks of the convolution, or null
    long long npeaks;
    float* band;                // [n + workgroups * BAND_SLACK] compacted band samples
    BandInfo* info;             // [workgroups]
    int build_band;             // 1: round 0 (stream + build), 0: later rounds (use the band if g allows)
    int step;                   // index of the (first) round this launch runs
    // the limiter's look-back words, preset to "unpublished" here when a limiter launch follows (saves
    // two fill launches on the stream); null otherwise
    unsigned long long* lim_published;
    long long lim_words;
    int* lim_ticket;
    unsigned long long* tail_gains;   // [tail_rounds + 1] gains published between the rounds of k_correction_tail (slot r:
                                      // the gain after its round r; slot tail_rounds: the gain after round 0), or null;
                                      // behind them [tail_rounds][tail_total] words for its workgroups' partial sums
                                      // (the value is the flag)
    int tail_total;                   // summing workgroups of the k_correction_tail launch that follows (0: none);
                                      // with one, this launch leaves its partials and the decision to that kernel
    int tail_rounds;                  // rounds that kernel runs (rms_correction_steps - 1; any number: defaults.py:118-120)
    int* error;                       // set when a bounded wait expired
};
// The decision of one round (stages.py:149-168), taken by ONE 256-thread workgroup after every partial
// sum has been published: piece sums -> loud pieces -> coefficient -> accumulated gain; with
// `final_peaks` also the peak / limiter early-out / normalisation scalars.  `total` partials, `per`
// of them per piece.  Every load is a cold miss: issued in batches of eight per thread, staged in LDS.
// `step` = index of this round, `gain_in` = the gain it ran with: nothing is read back from the
// CorrectionState, whose last writer may sit behind another XCD's L2 when rounds share a launch.
__device__ __forceinline__ double correction_decide(const RoundArgs& a, int total, int per, double* red, double* sums,
                                                    bool reset_arrivals, int step, double gain_in) {
    double* stage = sums + a.divisions;                          // [total]
    for (int k0 = threadIdx.x; k0 < total; k0 += 8 * 256) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)       // (write-through stores on the other side, L2-bypassing loads here)
            v[u] = k0 + 256 * u < total ? __hip_atomic_load(a.partial + k0 + 256 * u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (k0 + 256 * u < total) stage[k0 + 256 * u] = v[u];
    }
    __shared__ float fscratch[4];
    __shared__ double new_gain;
    float m = 0.f;
    if (a.final_peaks) {
        for (long long k0 = threadIdx.x; k0 < a.npeaks; k0 += 8 * 256) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = k0 + 256 * u < a.npeaks ? a.final_peaks[k0 + 256 * u] : 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) m = fmaxf(m, v[u]);
        }
    }
    __syncthreads();
    // (the last arriver works alone while the chip waits: lanes side by side on a piece and ONE wave's decision,
    // 2.5 us where wave-per-piece sums and three block-wide reductions took 5.3 -- profiles/r03_z_correction_phases.txt)
    piece_sums_by_groups(stage, per, a.divisions, sums);
    float pk = 0.f;
    if (a.final_peaks) pk = block_max<256>(m, fscratch);                   // (uniform)
    double avg = 0.0, match = 1.0;
    int count = 0;
    if (threadIdx.x < 64) wave_decide(sums, a.divisions, a.piece, 1.0, nullptr, nullptr, avg, match, count);
    if (threadIdx.x == 0) {
        const double c = *a.reference_match_rms / fmax(a.eps, match);      // match_levels.py:106-111
        CorrectionState* cs = a.cs;
        new_gain = gain_in * c;
        if (step < 16) cs->coeffs[step] = c;
        cs->steps_done = step + 1;
        cs->gain = new_gain;
        if (a.final_peaks) {
            const double peak = (double)(float)((double)pk * new_gain);      // max |float32(y*gain)|
            cs->result_peak = peak;
            const double rect = fmax(peak, a.threshold) / a.threshold;
            cs->limiter_active = fabs(rect - 1.0) > (1e-8 + 1e-5) ? 1 : 0;   // numpy.isclose defaults, hyrax.py:83
            cs->normalize_c = fmax(a.eps, peak / a.threshold);               // dsp.py:93-100
        }
        if (reset_arrivals)                                                  // ready for the next round or launch
            __hip_atomic_store(a.arrivals, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    return new_gain;
}

__global__ __launch_bounds__(256) void k_correction_round(RoundArgs a) {
    if (blockIdx.x == 0) ROUND_STAMP(0);
    warm_code(CODE_ROUND);
    MGX_LDS;
    double* red = reinterpret_cast<double*>(mgx_smem);          // 64 doubles of scratch
    double* sums = red + 64;                                     // [divisions]
    __shared__ int is_last;
    const int d = blockIdx.x / a.chunks, ch = blockIdx.x % a.chunks;
    const BandChunk bc = band_chunk(a.band, a.piece, a.chunks, d, ch);
    const long long b = bc.b, e = bc.e;
    const double g = a.cs->gain;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (a.tail_gains && blockIdx.x == 0)                                                         // "not yet": k_correction_tail
        for (int i = threadIdx.x; i < a.tail_rounds + 1 + a.tail_rounds * a.tail_total; i += 256) a.tail_gains[i] = ~0ull;
    if (a.lim_published) {
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < a.lim_words; i += (long long)gridDim.x * 256)
            a.lim_published[i] = ~0ull;
        if (blockIdx.x == 0 && threadIdx.x == 0) a.lim_ticket[0] = a.lim_ticket[2] = 0;      // ticket and "gave up" (mgx.hip run_limiter); a raised error sticks
    }
    // this workgroup's slice of the band buffer, one compacted list per wave
    float* wave_band = bc.lists + wave * bc.wave_cap;
    BandInfo* info = a.info + blockIdx.x;
    double acc = 0.0;
    // float64 product then clip: the reference clips the float64 mid (dsp.py:109-110)
    auto add = [&](float v) {
        const double c = fmin(fmax((double)v * g, -1.0), 1.0);
        acc = fma(c, c, acc);
    };
    const bool use_band = !a.build_band && g >= BAND_G_LO && g <= BAND_G_HI;
    if (use_band) {
        const int count = info->count[wave];
        for (int k = lane; k < count; k += 64) add(wave_band[k]);
        if (threadIdx.x == 0) acc += g * g * info->unclipped_sumsq + info->clipped_count;
    } else {
        // Straight-line per-sample code (no divergent branches: every lane walks the same iterations and
        // masks with `ok`; a branchy version of this loop made round 0 instruction-bound).  The band
        // test runs in float32 against thresholds rounded INTO the band, which can only move a sample
        // from the closed-form parts into the list -- the sum is the same either way.
        double low = 0.0;
        int filled = 0, clipped = 0;          // wave-uniform: band samples stored, always-clipped samples seen
        const unsigned long long below = (1ull << lane) - 1ull;
        const float t_never = band_threshold_never(), t_always = band_threshold_always();
        const bool build = a.build_band != 0;
        auto visit = [&](float v, bool ok) {
            const double d = (double)(ok ? v : 0.f);
            const double c = fmin(fmax(d * g, -1.0), 1.0);
            acc = fma(c, c, acc);
            if (build) {                       // uniform
                const float m = fabsf(v);
                const bool never = ok && m <= t_never, always = ok && m >= t_always;
                low = fma(never ? d : 0.0, d, low);
                clipped += __popcll(__ballot(always));
                const bool in_band = ok && !never && !always;
                const unsigned long long mask = __ballot(in_band);
                const long long slot = filled + __popcll(mask & below);
                if (in_band && slot < bc.wave_cap) wave_band[slot] = v;      // (a wave's quarter + slack never overflows)
                filled += __popcll(mask);
            }
        };
        // scalar head up to a 16-byte boundary, float4 body (64 B per thread in flight), scalar tail
        const long long head = min(e, (b + 3) & ~3ll);
        {
            const bool ok = b + threadIdx.x < head;
            visit(ok ? a.mid[b + threadIdx.x] : 0.f, ok);
        }
        const long long body_end = head + ((e - head) & ~3ll);
        int clipped_mine = 0;                  // per thread (the fast path below)
        if (build && g == 1.0) {
            // Round 0 of mgx_master (the level gain of stages.py:80-88 is in the filter, so g is exactly 1):
            // float32 arithmetic -- clip(v) is exact, the squares are summed 16 at a time before they join
            // the float64 sums -- and the band samples are compacted per THREAD: each thread counts its
            // own, one prefix sum over the wave places them, no ballot and no scalar chain per sample.
            // (A frame past the end loads as 0: clips to 0, counts as never clipped, adds nothing.)
            // MGX_ROUND0_GROUPS x four 16-byte loads per thread in flight, each group of sixteen samples then summed and
            // compacted in order (bit-identical whatever the number).  One group is the product: two and three were
            // measured in round 6 -- the whole stage 52.1 -> 54.2 / 54.3 us -- as were more, shorter workgroups (51.5 ->
            // 56.1 at twice as many): this kernel is not short of loads in flight (profiles/r06_b_*, r06_g_*).
#ifndef MGX_ROUND0_GROUPS
#define MGX_ROUND0_GROUPS 1
#endif
            constexpr int GROUPS = MGX_ROUND0_GROUPS;
            for (long long s0 = head; s0 < body_end; s0 += GROUPS * 4 * 1024) {
                float xs[GROUPS][16];
#pragma unroll
                for (int gq = 0; gq < GROUPS; ++gq) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const long long i = s0 + (gq * 4 + u) * 1024 + 4ll * threadIdx.x;
                        const float4 q = i < body_end ? *reinterpret_cast<const float4*>(a.mid + i) : make_float4(0.f, 0.f, 0.f, 0.f);
                        xs[gq][4 * u] = q.x; xs[gq][4 * u + 1] = q.y; xs[gq][4 * u + 2] = q.z; xs[gq][4 * u + 3] = q.w;
                    }
                }
#pragma unroll
                for (int gq = 0; gq < GROUPS; ++gq) {
                    const float (&x)[16] = xs[gq];
                    float sq = 0.f, lo = 0.f;
                    int mine = 0;
                    unsigned bits = 0;
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const float c = __builtin_amdgcn_fmed3f(x[i], -1.0f, 1.0f);
                        sq = fmaf(c, c, sq);
                        const float m = fabsf(x[i]);
                        const bool never = m <= t_never, always = m >= t_always;
                        lo = fmaf(never ? x[i] : 0.f, x[i], lo);
                        clipped_mine += always ? 1 : 0;
                        const bool in_band = !never && !always;
                        mine += in_band ? 1 : 0;
                        bits |= (in_band ? 1u : 0u) << i;
                    }
                    acc += (double)sq;
                    low += (double)lo;
#ifdef MGX_DEV_ROUND0_SUMS_ONLY         // development builds only: what round 0 costs without its band lists (results are wrong)
                    const int through = mine;
                    int slot = filled;
                    if (false) {
#else
                    const int through = wave_inclusive_sum(mine);
                    int slot = filled + through - mine;
                    if (bits) {
#endif
#pragma unroll
                        for (int i = 0; i < 16; ++i)
                            if (bits & (1u << i)) {
                                if (slot < bc.wave_cap) wave_band[slot] = x[i];
                                ++slot;
                            }
                    }
                    filled += __builtin_amdgcn_readlane(through, 63);
                }
            }
        } else
        for (long long s0 = head; s0 < body_end; s0 += 4 * 1024) {
            float4 v[4];
            bool ok[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long long i = s0 + u * 1024 + 4ll * threadIdx.x;
                ok[u] = i < body_end;
                v[u] = ok[u] ? *reinterpret_cast<const float4*>(a.mid + i) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                visit(v[u].x, ok[u]);
                visit(v[u].y, ok[u]);
                visit(v[u].z, ok[u]);
                visit(v[u].w, ok[u]);
            }
        }
        {
            const bool ok = body_end + threadIdx.x < e;
            visit(ok ? a.mid[body_end + threadIdx.x] : 0.f, ok);
        }
        if (build) {
            if (lane == 0) info->count[wave] = filled;
            const double lo = block_sum<256>(low, red);
            __syncthreads();
            const double hi = block_sum<256>((lane == 0 ? (double)clipped : 0.0) + (double)clipped_mine, red + 8);
            if (threadIdx.x == 0) {
                info->unclipped_sumsq = lo;
                info->clipped_count = hi;
            }
            __syncthreads();
        }
    }
    const double s = block_sum<256>(acc, red);
    if (a.build_band && a.tail_total > 0) {                              // uniform: k_correction_tail decides round 0
        if (threadIdx.x == 0) a.partial[blockIdx.x] = s;
        return;
    }
    if (threadIdx.x == 0) {
        // write-through 8-byte store + drained vmcnt instead of a release fence (a fence per workgroup
        // would write back the XCD's whole L2 two thousand times)
        __hip_atomic_store(a.partial + blockIdx.x, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // two-level arrival count (one word takes ~88 atomics per microsecond; 2048 workgroups on a
        // single word would cost more than the sums themselves)
        is_last = 0;
        if (atomicAdd(a.arrivals + 1 + d, 1u) == (unsigned)a.chunks - 1) {
            a.arrivals[1 + d] = 0;                                         // ready for the next launch
            is_last = atomicAdd(a.arrivals, 1u) == (unsigned)a.divisions - 1;
        }
        if (is_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (!is_last) return;
    ROUND_STAMP(1);
    correction_decide(a, a.divisions * a.chunks, a.chunks, red, sums, true, a.step, g);
    ROUND_STAMP(2);
}

// Rounds 1 .. K-1 of stages.py:149-168 in ONE launch.  After round 0 a round only touches the band
// lists (a few MB) and a handful of scalars, so a launch per round was mostly launch, ramp and a chain
// of cold round trips: 16 us each for ~1 us of work.  Here a small grid (divisions x groups workgroups,
// at most ~128: every one of them must be resident at once, also next to other handles' kernels) keeps
// running.  Before the first round a workgroup adds up the closed-form parts of its chunks, copies
// their band lists into LDS (when they fit) and reduces its share of the convolution's pair peaks; a
// round is then: sum from LDS -> publish the partial as an 8-byte word whose value is the flag (preset
// to all-ones by round 0) -> the deciding workgroup (one past the summing ones: tail_decider, which also
// takes round 0's decision while the others copy their lists) polls the words, one lane per word, decides
// with one wave and publishes the new gain the same way -> everybody polls it (one lane, bounded) and goes on.  Every wait
// is bounded and raises the handle's error word.  A gain outside [BAND_G_LO, BAND_G_HI] makes a
// workgroup stream its part of the mid plane instead (slow with so few workgroups, and never seen:
// coefficients are ratios of two loudness estimates of nearly the same signal).
// Phase stamps of this kernel and of round 0's last workgroup: profiles/r03_z_correction_phases.txt
// (-DMGX_TAIL_TRACE, tools/tail_trace.py).
#ifdef MGX_TEST_TAIL_MAX_SPINS                             // tests/test_device_errors.py: a tail that gives up quickly
constexpr int TAIL_MAX_SPINS = MGX_TEST_TAIL_MAX_SPINS;
#else
constexpr int TAIL_MAX_SPINS = 0;                          // product: bounded by time (wait_on, limiter_kernel.h)
#endif
// one lane's bounded wait for an 8-byte flag word to leave the all-ones pattern (`on_expiry` and the error word on expiry)
__device__ __forceinline__ unsigned long long poll_word(const unsigned long long* w, int* error, unsigned long long on_expiry) {
    unsigned long long v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int spins = 0;
    long long t0 = 0;
    while (v == ~0ull && wait_on(spins, t0, nullptr, TAIL_MAX_SPINS)) {
        if (spins < 64) __builtin_amdgcn_s_sleep(1);
        else __builtin_amdgcn_s_sleep(16);
        v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ++spins;
    }
    if (v == ~0ull) {
        error[DEVICE_ERROR_SLOT_TAIL] = 1;
        v = on_expiry;
    }
    return v;
}
// CorrectionState::coeffs mirrors mgx_report: the first 16 coefficients are kept for the log, the product of all is the gain
__device__ __forceinline__ void keep_coefficient(CorrectionState* cs, int step, double c) {
    if (step < 16) __hip_atomic_store(&cs->coeffs[step], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// The deciding workgroup of k_correction_tail (the one past the summing ones).  First round 0's decision from the
// partials k_correction_round left (the launch boundary made them visible; that kernel skips its own arrival
// count and decision when a tail follows -- they were 4 us with the whole chip waiting, here they run beside the
// other workgroups' list copies), then every round: poll the summing workgroups' words, decide, publish.
__device__ __forceinline__ void tail_decider(const RoundArgs& a, int groups, int rounds, int total, double* red, double* sums,
                                             double* stage, double* stage0, float* fscratch) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __shared__ double decided;
    double* peak_words = a.partial + (size_t)a.divisions * a.chunks;
    const float* final_peaks = a.final_peaks;
    auto put = [](double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto puti = [](int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    {
        const int n0 = a.divisions * a.chunks;
        for (int k0 = threadIdx.x; k0 < n0; k0 += 8 * 256) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = k0 + 256 * u < n0 ? a.partial[k0 + 256 * u] : 0.0;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (k0 + 256 * u < n0) stage0[k0 + 256 * u] = v[u];
        }
        const double gain_in = a.cs->gain;
        __syncthreads();
        piece_sums_by_groups(stage0, a.chunks, a.divisions, sums);
        if (wave == 0) {
            double avg, match;
            int count;
            wave_decide(sums, a.divisions, a.piece, 1.0, nullptr, nullptr, avg, match, count);
            if (lane == 0) {
                const double c = *a.reference_match_rms / fmax(a.eps, match);          // match_levels.py:106-111
                const double next = gain_in * c;
#ifdef MGX_TEST_TAIL_EXPIRE               // tests/test_device_errors.py: the first tail of the process never hears of round 0's gain
                if (atomicAdd(&g_test_tail_launches, 1) > 0)
#endif
                __hip_atomic_store(a.tail_gains + a.tail_rounds, double_bits(next), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                keep_coefficient(a.cs, a.step - 1, c);
                decided = next;
            }
        }
        __syncthreads();
    }
    double g = decided;
    for (int r = 0; r < rounds; ++r) {
        const unsigned long long* words = a.tail_gains + a.tail_rounds + 1 + (size_t)r * total;
        const bool last_round = r == rounds - 1;
        for (int k = threadIdx.x; k < total; k += 256) stage[k] = bits_double(poll_word(words + k, a.error, 0ull));
        asm volatile("" ::: "memory");        // the peak words are read AFTER their flag words were seen (compiler order;
                                              // the publisher waited for its peak store before it stored the flag)
        float m = 0.f;
        if (last_round && final_peaks)
            for (int k = threadIdx.x; k < total; k += 256)
                m = fmaxf(m, (float)__hip_atomic_load(peak_words + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        __syncthreads();
        for (int p = threadIdx.x; p < a.divisions; p += 256) {
            double t = 0.0;
            for (int q = 0; q < groups; ++q) t += stage[p * groups + q];
            sums[p] = t;
        }
        const float pk = block_max<256>(m, fscratch);                     // (barrier inside: sums[] is complete after it)
        if (wave == 0) {
            double avg, match;
            int count;
            wave_decide(sums, a.divisions, a.piece, 1.0, nullptr, nullptr, avg, match, count);
            if (lane == 0) {
                const double c = *a.reference_match_rms / fmax(a.eps, match);          // match_levels.py:106-111
                const double next = g * c;
                // the gain word first: a hundred workgroups are polling it
                if (!last_round)
                    __hip_atomic_store(a.tail_gains + r, double_bits(next), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                CorrectionState* cs = a.cs;
                keep_coefficient(cs, a.step + r, c);
                if (last_round) {
                    puti(&cs->steps_done, a.step + r + 1);
                    put(&cs->gain, next);
                    if (final_peaks) {
                        const double peak = (double)(float)((double)pk * next);      // max |float32(y*gain)|
                        const double rect = fmax(peak, a.threshold) / a.threshold;
                        put(&cs->result_peak, peak);
                        puti(&cs->limiter_active, fabs(rect - 1.0) > (1e-8 + 1e-5) ? 1 : 0);   // numpy.isclose defaults, hyrax.py:83
                        put(&cs->normalize_c, fmax(a.eps, peak / a.threshold));               // dsp.py:93-100
                    }
                }
                decided = next;
            }
        }
        __syncthreads();
        g = decided;
    }
}
constexpr int TAIL_CACHE_PER_WAVE = 3072;        // floats of band list a wave keeps in LDS
__host__ __device__ inline size_t correction_tail_lds_bytes(int divisions, int groups, int chunks) {
    const size_t cache = (size_t)4 * TAIL_CACHE_PER_WAVE * 4, stage0 = (size_t)divisions * chunks * 8;   // (the decider's)
    return ((size_t)64 + divisions + (size_t)divisions * groups) * 8 + (cache > stage0 ? cache : stage0) + 16;
}
__global__ __launch_bounds__(256) void k_correction_tail(RoundArgs a, int groups, int rounds) {
    warm_code(CODE_TAIL);
    TAIL_STAMP(0);
    MGX_LDS;
    double* red = reinterpret_cast<double*>(mgx_smem);          // 64 doubles of scratch
    double* sums = red + 64;                                     // [divisions]
    double* stage = sums + a.divisions;                          // [divisions * groups]
    float* cache = reinterpret_cast<float*>(stage + a.divisions * groups);
    __shared__ double gain_now;
    __shared__ float fscratch[4];
    const int d = blockIdx.x / groups, grp = blockIdx.x % groups;
    const int ch0 = (int)((long long)grp * a.chunks / groups), ch1 = (int)((long long)(grp + 1) * a.chunks / groups);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, total = a.divisions * groups;
    if (a.lim_published) {
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < a.lim_words; i += (long long)gridDim.x * 256)
            a.lim_published[i] = ~0ull;
        if (blockIdx.x == 0 && threadIdx.x == 0) a.lim_ticket[0] = a.lim_ticket[2] = 0;      // ticket and "gave up" (mgx.hip run_limiter); a raised error sticks
    }
    TAIL_STAMP(1);
    if ((int)blockIdx.x == total) {                                      // uniform: the extra workgroup decides
        tail_decider(a, groups, rounds, total, red, sums, stage, reinterpret_cast<double*>(cache), fscratch);
        return;
    }
    // ---- once: closed-form parts and band lists of this workgroup's chunks (lane c <-> chunk ch0 + c) ----
    const int nch = ch1 - ch0;                                           // <= 64 (host)
    int my_count = 0;
    double part_a = 0.0, part_c = 0.0;
    if (lane < nch) {
        const BandInfo* info = a.info + d * a.chunks + ch0 + lane;
        my_count = info->count[wave];
        if (wave == 0) { part_a = info->unclipped_sumsq; part_c = info->clipped_count; }
    }
    const double closed_a = wave_sum(part_a), closed_c = wave_sum(part_c);   // meaningful on wave 0
    int before = my_count;                                               // exclusive prefix of the counts over the lanes
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(before, o, 64);
        if (lane >= o) before += v;
    }
    const int wave_total = __shfl(before, 63, 64);
    before -= my_count;
    const bool cached = wave_total <= TAIL_CACHE_PER_WAVE;               // uniform per wave
    float* mine = cache + wave * TAIL_CACHE_PER_WAVE;
    if (cached) {
        // two chunks' lists at a time, six loads per lane and list in flight before the first is stored: a loop
        // of load -> wait -> store per 64 samples was 48 round trips in a row (profiles/r03_z_correction_phases.txt).
        // (Four chunks at a time, the first four asked for before the counts are known, eight chunks at a time
        // with 16-byte loads, and round 0's loads software-pipelined were all measured slower: more loads in
        // flight on these cold, scattered lists cost more than they hide.)
        constexpr int PER = 6;
        for (int c0 = 0; c0 < nch; c0 += 2) {
            const float* list[2];
            int n[2], off[2];
            float v[2][PER];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int c = c0 + u < nch ? c0 + u : c0;
                const BandChunk bc = band_chunk(a.band, a.piece, a.chunks, d, ch0 + c);
                list[u] = bc.lists + wave * bc.wave_cap;
                n[u] = c0 + u < nch ? __shfl(my_count, c, 64) : 0;
                off[u] = __shfl(before, c, 64);
#pragma unroll
                for (int j = 0; j < PER; ++j) v[u][j] = lane + 64 * j < n[u] ? list[u][lane + 64 * j] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
#pragma unroll
                for (int j = 0; j < PER; ++j)
                    if (lane + 64 * j < n[u]) mine[off[u] + lane + 64 * j] = v[u][j];
                for (int k = lane + 64 * PER; k < n[u]; k += 64) mine[off[u] + k] = list[u][k];
            }
        }
    }
    // this workgroup's share of the convolution's pair peaks, for the decider of the last round
    float my_peak = 0.f;
    if (a.final_peaks) {
        float m = 0.f;
        for (long long k = (long long)blockIdx.x * 256 + threadIdx.x; k < a.npeaks; k += (long long)total * 256)
            m = fmaxf(m, a.final_peaks[k]);
        my_peak = block_max<256>(m, fscratch);
    }
    double* peak_words = a.partial + (size_t)a.divisions * a.chunks;     // [total], behind round 0's partials
    const float* final_peaks = a.final_peaks;
    TAIL_STAMP(2);
    if (threadIdx.x == 0) gain_now = bits_double(poll_word(a.tail_gains + a.tail_rounds, a.error, double_bits(1.0)));
    __syncthreads();
    double g = gain_now;                                                 // round 0's, from the deciding workgroup
    for (int r = 0; r < rounds; ++r) {
        double acc = 0.0;
        auto add = [&](float v) {
            const double c = fmin(fmax((double)v * g, -1.0), 1.0);       // float64 product, then clip (dsp.py:109-110)
            acc = fma(c, c, acc);
        };
        if (g >= BAND_G_LO && g <= BAND_G_HI) {                          // uniform over the grid
            if (cached) {
                int k = lane;
                for (; k + 192 < wave_total; k += 256) {                 // four LDS loads in flight, summed in order
                    const float v0 = mine[k], v1 = mine[k + 64], v2 = mine[k + 128], v3 = mine[k + 192];
                    add(v0), add(v1), add(v2), add(v3);
                }
                for (; k < wave_total; k += 64) add(mine[k]);
            } else {
                for (int c = 0; c < nch; ++c) {
                    const BandChunk bc = band_chunk(a.band, a.piece, a.chunks, d, ch0 + c);
                    const float* list = bc.lists + wave * bc.wave_cap;
                    const int n = __shfl(my_count, c, 64);
                    for (int k = lane; k < n; k += 64) add(list[k]);
                }
            }
            if (threadIdx.x == 0) acc += g * g * closed_a + closed_c;
        } else {
            for (int c = 0; c < nch; ++c) {
                const BandChunk bc = band_chunk(a.band, a.piece, a.chunks, d, ch0 + c);
                for (long long i = bc.b + threadIdx.x; i < bc.e; i += 256) add(a.mid[i]);
            }
        }
        const double s = block_sum<256>(acc, red);
        TAIL_STAMP(3 + 6 * r);
        // The partial sum is published as an 8-byte word whose value is the flag (a sum of squares is never the
        // all-ones pattern round 0 left there); the deciding workgroup polls the words, one lane per word.  An
        // arrival counter cost each round the publisher's wait for its store, the atomic's round trip (a hundred
        // of them on one word take a microsecond) and the last arriver's read of the partials.
        unsigned long long* words = a.tail_gains + a.tail_rounds + 1 + (size_t)r * total;
        const bool last_round = r == rounds - 1;
        if (threadIdx.x == 0) {
            if (last_round && final_peaks) {                             // the peak word first, and landed
                __hip_atomic_store(peak_words + blockIdx.x, (double)my_peak, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __hip_atomic_store(words + blockIdx.x, double_bits(s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        TAIL_STAMP(5 + 6 * r);
        if (last_round) break;
        if (threadIdx.x == 0) gain_now = bits_double(poll_word(a.tail_gains + r, a.error, double_bits(1.0)));
        __syncthreads();
        g = gain_now;
        __syncthreads();
        TAIL_STAMP(8 + 6 * r);
    }
}

__device__ void correction_reset(CorrectionState* cs, double gain) {
    cs->gain = gain;
    cs->steps_done = 0;
    cs->result_peak = 0.0;
    cs->normalize_c = 1.0;
    cs->limiter_active = 1;
    for (int i = 0; i < 16; ++i) cs->coeffs[i] = 0.0;
}
__global__ void k_correction_init(CorrectionState* cs, double gain) {
    if (blockIdx.x == 0 && threadIdx.x == 0) correction_reset(cs, gain);
}

// peak of the corrected result, limiter early-out decision (hyrax.py:83-85 with numpy.isclose
// defaults) and the normalisation coefficient of stages.py:186-191 / dsp.py:93-100
__global__ __launch_bounds__(256) void k_finalize_scalars(const float* block_peak, long long nblocks,
                                                          double threshold, double eps, CorrectionState* cs) {
    __shared__ float scratch[4];
    float m = 0.f;
    for (long long i = threadIdx.x; i < nblocks; i += 256) m = fmaxf(m, block_peak[i]);
    const float pk = block_max<256>(m, scratch);
    if (threadIdx.x == 0) {
        const double peak = (double)(float)((double)pk * cs->gain);      // max |float32(y*gain)|
        cs->result_peak = peak;
        const double rect = fmax(peak, threshold) / threshold;
        cs->limiter_active = fabs(rect - 1.0) > (1e-8 + 1e-5) ? 1 : 0;
        cs->normalize_c = fmax(eps, peak / threshold);
    }
}

// result_no_limiter = y*gain (dsp.py:89-90) and/or the normalised variant
__global__ __launch_bounds__(256) void k_scale_outputs(const float2* y, long long n, const double* gain_ptr,
                                                       double gain_mul, const double* normalize_ptr,
                                                       float2* out_plain, float2* out_normalized) {
    const double g = (gain_ptr ? *gain_ptr : 1.0) * gain_mul;
    const double inv = normalize_ptr ? *normalize_ptr : 1.0;
    // two frames (16 bytes) per access where every buffer allows it; the odd last frame, if any, goes alone
    const bool wide = (((size_t)y | (size_t)out_plain | (size_t)out_normalized) & 15) == 0;
    if (!wide) {
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
            const float2 v = y[i];
            const double l = (double)v.x * g, r = (double)v.y * g;
            if (out_plain) out_plain[i] = make_float2((float)l, (float)r);
            if (out_normalized) out_normalized[i] = make_float2((float)(l / inv), (float)(r / inv));
        }
        return;
    }
    const long long pairs = n >> 1;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < pairs; i += (long long)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(y)[i];
        const double a = (double)v.x * g, b = (double)v.y * g, c = (double)v.z * g, d = (double)v.w * g;
        // (plain stores: non-temporal ones measured 58 vs 54 us here)
        if (out_plain) reinterpret_cast<float4*>(out_plain)[i] = make_float4((float)a, (float)b, (float)c, (float)d);
        if (out_normalized)
            reinterpret_cast<float4*>(out_normalized)[i] =
                make_float4((float)(a / inv), (float)(b / inv), (float)(c / inv), (float)(d / inv));
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const float2 v = y[n - 1];
        const double l = (double)v.x * g, r = (double)v.y * g;
        if (out_plain) out_plain[n - 1] = make_float2((float)l, (float)r);
        if (out_normalized) out_normalized[n - 1] = make_float2((float)(l / inv), (float)(r / inv));
    }
}

// ---- A/B previews (preview_creator.py:30-94) --------------------------------------------------------
// dsp.py:128-143 strided_app_2d + batch_rms_2d: windows of `size` frames every `step` frames; the loudest one
// is argmax of sqrt(mean(x^2)) over both channels = argmax of the plain sum of squares.  grid = (chunks,
// windows): workgroup (c, w) sums chunk c of window w in float64 (float32 products are exact in float64);
// the host adds a window's chunks in order and takes the argmax of a few hundred numbers.
__global__ __launch_bounds__(256) void k_window_energy(const float2* x, long long size, long long step, int chunks,
                                                       double* partial /* [windows][chunks] */, long long first_window) {
    __shared__ double scratch[4];
    partial += (size_t)first_window * chunks;                   // (grids of at most 65535 windows each)
    const long long begin = (first_window + (long long)blockIdx.y) * step;
    const long long len = (size + chunks - 1) / chunks;
    const long long b = begin + (long long)blockIdx.x * len, e = min(begin + size, b + len);
    double acc = 0.0;
    for (long long i = b + threadIdx.x; i < e; i += 256) {
        const float2 v = x[i];
        acc = fma((double)v.x, (double)v.x, acc);
        acc = fma((double)v.y, (double)v.y, acc);
    }
    const double s = block_sum<256>(acc, scratch);
    if (threadIdx.x == 0) partial[(size_t)blockIdx.y * chunks + blockIdx.x] = s;
}
// the cut: out[i] = fade(i) * clip(x[begin + i], -limit, limit) for i < size (dsp.py:109-110 clip -- limit <= 0:
// none --, dsp.py:146-152 fade: numpy.linspace(0, 1, fade) over the first `fade` frames, its mirror over the
// last `fade`; both factors where the two ramps overlap, as the reference's two in-place products give)
__global__ __launch_bounds__(256) void k_preview_cut(const float2* x, long long begin, long long size, long long fade,
                                                     double limit, float2* out) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < size; i += (long long)gridDim.x * 256) {
        const float2 v = x[begin + i];
        double l = v.x, r = v.y;
        if (limit > 0.0) {
            l = fmin(fmax(l, -limit), limit);
            r = fmin(fmax(r, -limit), limit);
        }
        double g = 1.0;
        if (fade > 0) {
            const double denom = fade > 1 ? (double)(fade - 1) : 1.0;         // linspace(0, 1, 1) = [0]
            if (i < fade) g *= (double)i / denom;
            if (i >= size - fade) g *= (double)(size - 1 - i) / denom;
        }
        out[i] = make_float2((float)(l * g), (float)(r * g));
    }
}

// per-block max(|L|,|R|) of interleaved frames (4096 frames per block)
__global__ __launch_bounds__(256) void k_frame_peaks(const float2* x, long long n, float* block_peak) {
    __shared__ float scratch[4];
    const long long b = (long long)blockIdx.x * 4096;
    float m = 0.f;
    for (int i = threadIdx.x; i < 4096; i += 256) {
        const long long f = b + i;
        if (f < n) {
            const float2 v = x[f];
            m = fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y)));
        }
    }
    const float r = block_max<256>(m, scratch);
    if (threadIdx.x == 0) block_peak[blockIdx.x] = r;
}

// ---------------------------------------------------------------------------
// limiter (limiter_kernel.h): one launch, grid = chunks
// ---------------------------------------------------------------------------
// Ordered composition of affine maps across a workgroup: inclusive scan over the 64 lanes of each
// wave by shuffles (scan order = lane order, or reversed), wave totals through LDS, then every
// thread composes the totals of the waves before it.  Returns the composition of all maps BEFORE
// this thread in scan order; `*whole` (if wanted) the composition of everything.
template <bool REVERSE>
__device__ __forceinline__ Affine wave_inclusive(Affine m) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        Affine o;
        o.a = REVERSE ? __shfl_down(m.a, d, 64) : __shfl_up(m.a, d, 64);
        o.b = REVERSE ? __shfl_down(m.b, d, 64) : __shfl_up(m.b, d, 64);
        const bool has = REVERSE ? (lane + d < 64) : (lane >= d);
        if (has) m = affine_then(o, m);
    }
    return m;
}
template <bool REVERSE>
__device__ __forceinline__ Affine wave_exclusive(Affine inclusive) {
    const int lane = threadIdx.x & 63;
    Affine o;
    o.a = REVERSE ? __shfl_down(inclusive.a, 1, 64) : __shfl_up(inclusive.a, 1, 64);
    o.b = REVERSE ? __shfl_down(inclusive.b, 1, 64) : __shfl_up(inclusive.b, 1, 64);
    const bool first = REVERSE ? lane == 63 : lane == 0;
    return first ? affine_identity() : o;
}
// totals[w] = inclusive total of wave w (written by the caller before the barrier)
template <bool REVERSE, int WAVES>
__device__ __forceinline__ Affine compose_waves(const Affine* totals, Affine exclusive_in_wave, Affine* whole) {
    const int w = threadIdx.x >> 6;
    Affine before = affine_identity(), all = affine_identity();
#pragma unroll
    for (int i = 0; i < WAVES; ++i) {
        const int k = REVERSE ? WAVES - 1 - i : i;           // waves in scan order
        const Affine t = totals[k];
        const bool earlier = REVERSE ? k > w : k < w;
        if (earlier) before = affine_then(before, t);
        all = affine_then(all, t);
    }
    if (whole) *whole = all;
    return affine_then(before, exclusive_in_wave);
}

// maximum over the eight lanes that share lane >> 3 (non-negative values): three DPP steps
__device__ __forceinline__ float dpp_max8(float v) {
    // (integer maxima of the bit patterns: the values are non-negative, pmax in mgx_hd.h)
    int x = __float_as_int(v);
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, true));   // row_half_mirror
    return __int_as_float(x);
}

#ifdef MGX_DEV_LIMITER_PHASES      // development builds only: where a chunk's time goes (tools/limiter_phases.py)
constexpr int DEV_PHASE_CHUNKS = 16384;
__device__ unsigned mgx_dev_phase_ticks[DEV_PHASE_CHUNKS][16];       // [chunk][mark]: ticks since the previous mark; [15] = start time
// every chunk's life, quiet ones included: {kernel entry, frames loaded, end, kind (0 edge, 1 busy, 2 quiet) | cu << 8 | xcc << 20}
__device__ long long mgx_dev_chunk_life[DEV_PHASE_CHUNKS][8];    // [4] hold word out, [5] hold carry in, [6] release word out, [7] release carry in
#define DEV_LIFE(slot, value)                                                                        \
    do {                                                                                             \
        if (threadIdx.x == 0 && chunk < DEV_PHASE_CHUNKS) mgx_dev_chunk_life[chunk][slot] = (value); \
    } while (0)
#define DEV_MARK(k)                                                                                  \
    do {                                                                                             \
        if (threadIdx.x == 0 && chunk < DEV_PHASE_CHUNKS) {                                          \
            const long long now = wall_clock64();                                                    \
            mgx_dev_phase_ticks[chunk][k] = (unsigned)(now - dev_last);                              \
            dev_last = now;                                                                          \
        }                                                                                            \
    } while (0)
#else
#define DEV_MARK(k)
#define DEV_LIFE(slot, value)
#endif
// one chunk, from the load phase to the store; FULL = the chunk lies strictly inside the track
// one chunk, from the load phase to the store; FULL = the chunk lies strictly inside the track
template <int T, bool FULL>
__device__ __forceinline__ void limit_chunk(const LimiterArgs& a, long long chunk, float* lds) {
    using LB = LimiterBlock<T>;
    // (opaque: nothing derived from the thread id may be hoisted out of a persistent caller's loop)
    const int tid = opaque((int)threadIdx.x), lane = tid & 63, wave = tid >> 6;
#ifdef MGX_DEV_LIMITER_PHASES
    long long dev_last = wall_clock64();
    if (threadIdx.x == 0 && chunk < DEV_PHASE_CHUNKS) mgx_dev_phase_ticks[chunk][15] = (unsigned)dev_last;
#endif
    if (!FULL) {                                 // (a FULL chunk was loaded by the kernel: limit_chunk_quiet's frames)
        float pm[LB::E / 2];
        LB::template phase_load<FULL>(opaque(tid), chunk, a, lds, pm);
#pragma unroll
        for (int j = 0; j < LB::E / 2; ++j) {
            const float m = dpp_max8(pm[j]);
            if ((tid & 7) == 0) LB::block_max(lds)[LB::block_of(tid, j)] = m;
        }
        __syncthreads();
    }
    DEV_MARK(0);      // load

    // hold filter first (scan 1): its aggregate is published as early as possible
    typename LB::Thread th;
    Affine whole;
    const bool busy = __any(LB::neighbourhood_max(opaque(tid), a, lds) > 0.f) != 0;      // wave-uniform
    {
        const Affine m1 = LB::template phase_hold_window<FULL>(opaque(tid), chunk, a, th, lds, busy);
        const Affine i1 = wave_inclusive<false>(m1);
        if (lane == 63) LB::wave_totals(lds, 1)[wave] = i1;
        const Affine e1 = wave_exclusive<false>(i1);
        __syncthreads();
        th.hold_pre = compose_waves<false, LB::WAVES>(LB::wave_totals(lds, 1), e1, &whole);
    }
    DEV_MARK(1);      // hold window + scan
    if (tid == 0) LB::lookback_publish(chunk, 0, a, whole.b);
    DEV_LIFE(4, wall_clock64());
    // ask for the predecessors' words now, take them after the attack path (wave 0: hold, wave 1: attack)
    typename LB::Polls polls;
    if (wave == 0) LB::lookback_ask(lane, chunk, 0, a, polls);
    // forward attack smoother (scan 0)
    Affine p0;
    {
        const Affine m0 = LB::template phase_attack_window<FULL>(opaque(tid), a, th, lds, busy);
        const Affine i0 = wave_inclusive<false>(m0);
        if (lane == 63) LB::wave_totals(lds, 0)[wave] = i0;
        const Affine e0 = wave_exclusive<false>(i0);
        __syncthreads();
        p0 = compose_waves<false, LB::WAVES>(LB::wave_totals(lds, 0), e0, nullptr);
    }
    DEV_MARK(2);      // attack window + scan
    if (tid == LB::T - a.gr) LB::lookback_publish(chunk, 2, a, p0.b);          // attack state at the end of the core
    if (wave == 1) LB::lookback_ask(lane, chunk, 2, a, polls);
    const bool tail = !FULL && LB::tail_chunk(chunk, a);                        // uniform
    double att_now = 0.0;
    if (tail) {
        if (wave == 1) {
            const double s = wave_sum(LB::lookback_take(lane, chunk, 2, a, polls));
            if (lane == 0) LB::scalars(lds)[2] = s;
        }
        __syncthreads();
        att_now = LB::scalars(lds)[2];
    }

    // backward attack smoother, right to left (scan 2)
    const Affine mb = LB::template phase_attack_forward<FULL>(opaque(tid), a, th, p0, att_now, lds);
    const Affine ib = wave_inclusive<true>(mb);
    if (lane == 0) LB::wave_totals(lds, 2)[wave] = ib;
    const Affine eb = wave_exclusive<true>(ib);
    __syncthreads();
    const Affine pb = compose_waves<true, LB::WAVES>(LB::wave_totals(lds, 2), eb, nullptr);
    LB::template phase_attack_backward<FULL>(opaque(tid), a, th, pb);
    DEV_MARK(3);      // attack forward, scan, backward
    if (wave == 0) {
        const double s = wave_sum(LB::lookback_take(lane, chunk, 0, a, polls));
        if (lane == 0) LB::scalars(lds)[0] = s;
    }
    if (wave == 1 && !tail) {
        const double s = wave_sum(LB::lookback_take(lane, chunk, 2, a, polls));
        if (lane == 0) LB::scalars(lds)[2] = s;
    }
    DEV_MARK(4);      // take hold (wave 0)
    DEV_LIFE(5, wall_clock64());
    __syncthreads();
    DEV_MARK(5);      // barrier after the takes (waits for wave 1's attack take)

    // hold output, release filter (scan 3)
    const Affine mr = LB::template phase_hold<FULL>(opaque(tid), a, th, LB::scalars(lds)[0], tail ? 0.0 : LB::scalars(lds)[2]);
    const Affine ir = wave_inclusive<false>(mr);
    if (lane == 63) LB::wave_totals(lds, 3)[wave] = ir;
    const Affine er = wave_exclusive<false>(ir);
    __syncthreads();
    const Affine pr = compose_waves<false, LB::WAVES>(LB::wave_totals(lds, 3), er, &whole);
    DEV_MARK(6);      // hold output + release scan
    if (tid == 0) LB::lookback_publish(chunk, 1, a, whole.b);
    DEV_LIFE(6, wall_clock64());
    if (wave == 0) LB::lookback_ask(lane, chunk, 1, a, polls);
    typename LB::Reload again;
    if (FULL) LB::phase_reload(opaque(tid), chunk, a, again);
    DEV_MARK(7);      // publish, ask, reload issue
    if (wave == 0) {
        const double s = wave_sum(LB::lookback_take(lane, chunk, 1, a, polls));
        if (lane == 0) LB::scalars(lds)[1] = s;
    }
    DEV_MARK(8);      // take release
    DEV_LIFE(7, wall_clock64());
    __syncthreads();
    LB::template phase_gain<FULL>(opaque(tid), a, th, pr, LB::scalars(lds)[1], lds);
    __syncthreads();
    DEV_MARK(9);      // gain
    if (FULL) LB::phase_store_reloaded(opaque(tid), chunk, a, again, lds);
    else LB::template phase_store<FULL>(opaque(tid), chunk, a, true, lds);
    DEV_MARK(10);     // store
}

// A chunk without a single frame above the threshold (limiter_kernel.h, "quiet chunks"): two look-backs, no
// windows, no scans, no reload.  45 % of the chunks of the benchmark's 8-minute pair; none of a track that is
// limited everywhere.
template <int T>
__device__ __forceinline__ void limit_chunk_quiet(const LimiterArgs& a, long long chunk, float* lds,
                                                  const typename LimiterBlock<T>::Reload& kept) {
    using LB = LimiterBlock<T>;
    const int tid = opaque((int)threadIdx.x), lane = tid & 63, wave = tid >> 6;
    if (tid == 0) {
        LB::lookback_publish(chunk, 0, a, 0.0);
        LB::lookback_publish(chunk, 2, a, 0.0);
    }
    DEV_LIFE(4, wall_clock64());
    typename LB::Polls polls;
    if (wave == 0) {
        LB::lookback_ask(lane, chunk, 0, a, polls);
        const double s = wave_sum(LB::lookback_take(lane, chunk, 0, a, polls));
        if (lane == 0) LB::scalars(lds)[0] = s;
    }
    if (wave == 1) {
        LB::lookback_ask(lane, chunk, 2, a, polls);
        const double s = wave_sum(LB::lookback_take(lane, chunk, 2, a, polls));
        if (lane == 0) LB::scalars(lds)[2] = s;
    }
    __syncthreads();
    const double hc = LB::scalars(lds)[0], ac = LB::scalars(lds)[2];
    DEV_LIFE(5, wall_clock64());
    if (tid == 0) LB::lookback_publish(chunk, 1, a, hc * a.quiet_rel_gain);
    DEV_LIFE(6, wall_clock64());
    if (wave == 0) {
        LB::lookback_ask(lane, chunk, 1, a, polls);
        const double s = wave_sum(LB::lookback_take(lane, chunk, 1, a, polls));
        if (lane == 0) LB::scalars(lds)[1] = s;
    }
    __syncthreads();
    DEV_LIFE(7, wall_clock64());
    LB::phase_quiet_store(tid, chunk, a, kept, hc, ac, LB::scalars(lds)[1]);
}

// ---- PCM at the boundary (loader.py:35 / saver.py:27-33: what soundfile does on the host) ------------
// Files hold integer samples; moving those over PCIe instead of float32 halves (16 bit) the bytes either
// way.  Scaling follows libsndfile: read x = v / 2^(bits-1) (exact in float32 up to 24 bits), write
// v = rint(x * (2^(bits-1) - 1)) clipped to the integer range, computed in float64 so that the result is
// the one the host codec (audio_io.write_wav) produces from the same float32 sample.  24-bit samples are
// packed little-endian, three bytes each: a thread moves four of them as three 32-bit words.
__global__ __launch_bounds__(256) void k_pcm_decode(const void* pcm, long long samples, int bits, float* out) {
    const long long stride = (long long)gridDim.x * 256;
    if (bits == 16) {
        const short* in = static_cast<const short*>(pcm);
        for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i < samples; i += stride * 4) {
            if (i + 4 <= samples) {
                const short4 v = *reinterpret_cast<const short4*>(in + i);
                *reinterpret_cast<float4*>(out + i) = make_float4(v.x * (1.f / 32768.f), v.y * (1.f / 32768.f),
                                                                  v.z * (1.f / 32768.f), v.w * (1.f / 32768.f));
            } else {
                for (long long k = i; k < samples; ++k) out[k] = in[k] * (1.f / 32768.f);
            }
        }
    } else if (bits == 32) {
        const int* in = static_cast<const int*>(pcm);
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < samples; i += stride)
            out[i] = (float)((double)in[i] * (1.0 / 2147483648.0));
    } else {                                     // 24 bits packed: samples 4q .. 4q+3 = bytes 12q .. 12q+11
        const unsigned* in = static_cast<const unsigned*>(pcm);
        const unsigned char* bytes = static_cast<const unsigned char*>(pcm);
        const long long quads = samples / 4;
        for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < quads; q += stride) {
            const unsigned w0 = in[3 * q], w1 = in[3 * q + 1], w2 = in[3 * q + 2];
            const int v0 = (int)(w0 << 8) >> 8;
            const int v1 = (int)(((w0 >> 24) | (w1 << 8)) << 8) >> 8;
            const int v2 = (int)(((w1 >> 16) | (w2 << 16)) << 8) >> 8;
            const int v3 = (int)w2 >> 8;
            *reinterpret_cast<float4*>(out + 4 * q) = make_float4(v0 * (1.f / 8388608.f), v1 * (1.f / 8388608.f),
                                                                  v2 * (1.f / 8388608.f), v3 * (1.f / 8388608.f));
        }
        if (blockIdx.x == 0 && threadIdx.x < (int)(samples - 4 * quads)) {
            const long long k = 4 * quads + threadIdx.x;
            const int v = (int)(((unsigned)bytes[3 * k] | ((unsigned)bytes[3 * k + 1] << 8) | ((unsigned)bytes[3 * k + 2] << 16)) << 8) >> 8;
            out[k] = v * (1.f / 8388608.f);
        }
    }
}
__device__ __forceinline__ int pcm_quantise(float x, double top) {
    const double q = rint((double)x * top);
    return (int)fmin(fmax(q, -top - 1.0), top);
}
__global__ __launch_bounds__(256) void k_pcm_encode(const float* x, long long samples, int bits, void* pcm) {
    const long long stride = (long long)gridDim.x * 256;
    const double top = (double)((1ll << (bits - 1)) - 1);
    if (bits == 16) {
        short* out = static_cast<short*>(pcm);
        for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i < samples; i += stride * 4) {
            if (i + 4 <= samples) {
                const float4 v = *reinterpret_cast<const float4*>(x + i);
                short4 o;
                o.x = (short)pcm_quantise(v.x, top); o.y = (short)pcm_quantise(v.y, top);
                o.z = (short)pcm_quantise(v.z, top); o.w = (short)pcm_quantise(v.w, top);
                *reinterpret_cast<short4*>(out + i) = o;
            } else {
                for (long long k = i; k < samples; ++k) out[k] = (short)pcm_quantise(x[k], top);
            }
        }
    } else if (bits == 32) {
        int* out = static_cast<int*>(pcm);
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < samples; i += stride)
            out[i] = pcm_quantise(x[i], top);
    } else {
        unsigned* out = static_cast<unsigned*>(pcm);
        unsigned char* bytes = static_cast<unsigned char*>(pcm);
        const long long quads = samples / 4;
        for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < quads; q += stride) {
            const float4 v = *reinterpret_cast<const float4*>(x + 4 * q);
            const unsigned a = (unsigned)pcm_quantise(v.x, top) & 0xFFFFFFu, b = (unsigned)pcm_quantise(v.y, top) & 0xFFFFFFu;
            const unsigned c = (unsigned)pcm_quantise(v.z, top) & 0xFFFFFFu, d = (unsigned)pcm_quantise(v.w, top) & 0xFFFFFFu;
            out[3 * q] = a | (b << 24);
            out[3 * q + 1] = (b >> 8) | (c << 16);
            out[3 * q + 2] = (c >> 16) | (d << 8);
        }
        if (blockIdx.x == 0 && threadIdx.x < (int)(samples - 4 * quads)) {
            const long long k = 4 * quads + threadIdx.x;
            const unsigned v = (unsigned)pcm_quantise(x[k], top);
            bytes[3 * k] = (unsigned char)v; bytes[3 * k + 1] = (unsigned char)(v >> 8); bytes[3 * k + 2] = (unsigned char)(v >> 16);
        }
    }
}

// dsp.py:49-54 count_max_peaks on frames in HBM: the largest magnitude, then how many samples numpy.isclose
// would put on it (|x - m| <= 1e-8 + 1e-5 m, either sign), evaluated in float64 like numpy does on the
// float32 values.  out[0] = bits of the maximum (a non-negative float orders like its bit pattern),
// out[1] = the count; both zeroed by the caller.
__global__ __launch_bounds__(256) void k_peak_max(const float* x, long long samples, unsigned long long* out) {
    __shared__ float red[4];
    float m = 0.f;
    for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i < samples; i += (long long)gridDim.x * 1024) {
        if (i + 4 <= samples) {
            const float4 v = *reinterpret_cast<const float4*>(x + i);
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        } else {
            for (long long k = i; k < samples; ++k) m = fmaxf(m, fabsf(x[k]));
        }
    }
    const float b = block_max<256>(m, red);
    if (threadIdx.x == 0) atomicMax(out, (unsigned long long)__float_as_uint(b));
}
__global__ __launch_bounds__(256) void k_peak_count(const float* x, long long samples, unsigned long long* out) {
    __shared__ double red[4];
    const double peak = (double)__uint_as_float((unsigned)out[0]);
    const double tol = 1e-8 + 1e-5 * peak;
    int c = 0;
    for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i < samples; i += (long long)gridDim.x * 1024) {
        if (i + 4 <= samples) {
            const float4 v = *reinterpret_cast<const float4*>(x + i);
            c += (fabs(fabs((double)v.x) - peak) <= tol) + (fabs(fabs((double)v.y) - peak) <= tol) +
                 (fabs(fabs((double)v.z) - peak) <= tol) + (fabs(fabs((double)v.w) - peak) <= tol);
        } else {
            for (long long k = i; k < samples; ++k) c += fabs(fabs((double)x[k]) - peak) <= tol;
        }
    }
    const double total = block_sum<256>((double)c, red);
    if (threadIdx.x == 0 && total > 0.0) atomicAdd(out + 1, (unsigned long long)total);
}

// One chunk with hold / release filters of order up to K (limiter_general.h): the load, window, attack and
// store phases of the first-order kernel; the two low-passes as K-state maps scanned through LDS.
template <int K>
__device__ __forceinline__ void limit_chunk_general(const LimiterArgs& a, const GeneralArgs<K>& g, long long chunk, float* lds) {
    using LB = LimiterBlock<256>;
    using LG = LimiterGeneral<K>;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    {
        float pm[LB::E / 2];
        LB::template phase_load<false>(tid, chunk, a, lds, pm);
#pragma unroll
        for (int j = 0; j < LB::E / 2; ++j) {
            const float m = dpp_max8(pm[j]);
            if ((tid & 7) == 0) LB::block_max(lds)[LB::block_of(tid, j)] = m;
        }
    }
    __syncthreads();
    typename LB::Thread th;
    LB::template phase_hold_window<false>(tid, chunk, a, th, lds);          // (its first-order map is not used)
    LG::scan_put(lds, tid, th.core && th.valid > 0 ? LG::block_map(g.hold, th.sh, th.valid, g.pow_hold) : LG::identity());
    __syncthreads();
    LG::scan_groups(lds, tid);
    __syncthreads();
    LG::scan_top(lds, tid);
    __syncthreads();
    const StateMap<K> hold_pre = LG::scan_prefix(lds, tid);
    if (tid == 0) LG::publish(g, a.nchunks, 0, chunk, LG::scan_whole(lds).v);

    // the attack path, as in limit_chunk
    typename LB::Polls polls;
    Affine p0;
    {
        const Affine m0 = LB::template phase_attack_window<false>(tid, a, th, lds);
        const Affine i0 = wave_inclusive<false>(m0);
        if (lane == 63) LB::wave_totals(lds, 0)[wave] = i0;
        const Affine e0 = wave_exclusive<false>(i0);
        __syncthreads();
        p0 = compose_waves<false, LB::WAVES>(LB::wave_totals(lds, 0), e0, nullptr);
    }
    if (tid == LB::T - a.gr) LB::lookback_publish(chunk, 2, a, p0.b);
    if (wave == 1) LB::lookback_ask(lane, chunk, 2, a, polls);
    const bool tail = LB::tail_chunk(chunk, a);
    double att_now = 0.0;
    if (tail) {
        if (wave == 1) {
            const double s = wave_sum(LB::lookback_take(lane, chunk, 2, a, polls));
            if (lane == 0) LB::scalars(lds)[2] = s;
        }
        __syncthreads();
        att_now = LB::scalars(lds)[2];
    }
    const Affine mb = LB::template phase_attack_forward<false>(tid, a, th, p0, att_now, lds);
    const Affine ib = wave_inclusive<true>(mb);
    if (lane == 0) LB::wave_totals(lds, 2)[wave] = ib;
    const Affine eb = wave_exclusive<true>(ib);
    __syncthreads();
    const Affine pb = compose_waves<true, LB::WAVES>(LB::wave_totals(lds, 2), eb, nullptr);
    LB::template phase_attack_backward<false>(tid, a, th, pb);

    // carries of the hold filter (wave 0) and of the attack smoother (wave 1)
    double* carries = LG::carries(lds);
    if (wave == 0) {
        double acc[K];
        LG::take(lane, chunk, 0, g, a, acc);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const double s = wave_sum(acc[k]);
            if (lane == 0) carries[k] = s;
        }
    }
    if (wave == 1 && !tail) {
        const double s = wave_sum(LB::lookback_take(lane, chunk, 2, a, polls));
        if (lane == 0) LB::scalars(lds)[2] = s;
    }
    __syncthreads();
    double hold_carry[K];
#pragma unroll
    for (int k = 0; k < K; ++k) hold_carry[k] = carries[k];
    const StateMap<K> mr = LG::phase_hold(tid, a, g, th, hold_pre, hold_carry, tail ? 0.0 : LB::scalars(lds)[2]);
    __syncthreads();                                                          // every prefix of the hold scan has been read
    LG::scan_put(lds, tid, mr);
    __syncthreads();
    LG::scan_groups(lds, tid);
    __syncthreads();
    LG::scan_top(lds, tid);
    __syncthreads();
    const StateMap<K> rel_pre = LG::scan_prefix(lds, tid);
    if (tid == 0) LG::publish(g, a.nchunks, 1, chunk, LG::scan_whole(lds).v);
    if (wave == 0) {
        double acc[K];
        LG::take(lane, chunk, 1, g, a, acc);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const double s = wave_sum(acc[k]);
            if (lane == 0) carries[K + k] = s;
        }
    }
    __syncthreads();
    double rel_carry[K];
#pragma unroll
    for (int k = 0; k < K; ++k) rel_carry[k] = carries[K + k];
    LG::phase_gain(tid, g, th, rel_pre, rel_carry, lds);
    __syncthreads();
    LB::template phase_store<false>(tid, chunk, a, true, lds);
}

template <int K>
__global__ __launch_bounds__(256, 2) void k_limit_general(LimiterArgs a, GeneralArgs<K> g) {
    using LB = LimiterBlock<256>;
    MGX_LDS;
    float* lds = reinterpret_cast<float*>(mgx_smem);
    int& ticket = *reinterpret_cast<int*>(LB::scalars(lds) + 4);
    const bool active = a.active ? (*a.active != 0) : true;
    if (!active) {                       // hyrax.py:83-85
        LB::phase_store(threadIdx.x, blockIdx.x, a, false, lds);
        return;
    }
    long long chunk = blockIdx.x;            // (the workgroup's number, or a ticket: see k_limit)
    if (a.ticket) {
        if (threadIdx.x == 0) ticket = atomicAdd(a.ticket, 1);
        __syncthreads();
        chunk = ticket;
    }
    limit_chunk_general<K>(a, g, chunk, lds);
}

// T = threads = 16-frame blocks per chunk (256, or 1024 for long attack / hold times); WGS = workgroups
// per CU the kernel is compiled for (register budget 512 / (WGS * T / 256) per lane)
// HW / HB / GR >= 0: an instantiation for ONE window geometry (attack half window, hold look-back, right halo blocks;
// gl and gw follow from them): the bounds of every window loop are literals, the masked ragged-edge reads of the
// general form fold away and the code is a third shorter.  The host launches it when the configuration's numbers are
// exactly these (44.1 and 48 kHz with the reference's default 1 ms attack and hold, defaults.py:25-58), the general
// instantiation (-1) otherwise; the results are the same to the bit.
template <int T, int WGS, int HW = -1, int HB = -1, int GR = -1>
__global__ __launch_bounds__(T, WGS * T / 256) void k_limit(LimiterArgs a0) {
    warm_code(CODE_LIMIT, T == 256 ? 0 : 1);
    using LB = LimiterBlock<T>;
    LimiterArgs a = a0;
    if (HW >= 0) {
        a.hw = HW;
        a.hb = HB;
        a.gl = (HW + HB + LB::E - 1) / LB::E;
        a.gw = (HW + LB::E - 1) / LB::E;
        a.gr = GR;
    }
    MGX_LDS;
    float* lds = reinterpret_cast<float*>(mgx_smem);
    int& ticket = *reinterpret_cast<int*>(LB::scalars(lds) + 4);      // dynamic LDS only (16-byte aligned base)
    const int tid = threadIdx.x;
    const bool active = a.active ? (*a.active != 0) : true;
    if (!active) {                       // hyrax.py:83-85: the array passes through, then stages.py:203
        LB::phase_store(tid, blockIdx.x, a, false, lds);
        return;
    }
#ifdef MGX_DEV_LIMITER_PHASES
    const long long dev_entry = wall_clock64();
#endif
    // Which chunk?  The workgroup's own number.  A chunk waits for words of LOWER-numbered chunks only, and the
    // dispatcher walks a grid in the order of the workgroup numbers (the walk may stall on an XCD whose slots are all
    // taken, but whatever it has handed out is lower-numbered than what it has not): the lowest unfinished chunk has
    // always been handed out, all it waits for is finished, so it finishes -- no chunk can wait for ever.  An atomic
    // ticket (a.ticket != null) gives the same guarantee without leaning on the dispatch order, at the price of a
    // returning atomic and a barrier in front of every chunk's loads (entry to loaded 6.2 -> 3.3 us, the kernel
    // 152 -> 134 us: profiles/r05_g_*); a handle falls back to it if a bounded look-back wait ever expires.
    long long chunk = blockIdx.x;
    if (a.ticket) {
        if (tid == 0) ticket = atomicAdd(a.ticket, 1);
        __syncthreads();
        chunk = ticket;
    }
    DEV_LIFE(0, dev_entry);
    if (!LB::full_chunk(chunk, a)) {
        limit_chunk<T, false>(a, chunk, lds);
        DEV_LIFE(2, wall_clock64());
        DEV_LIFE(3, 0ll | ((long long)__builtin_amdgcn_s_getreg((4 << 11) | 4) << 8));
        return;
    }
    // a chunk inside the track: load (the frames stay in registers until it is known whether the chunk is
    // quiet), block maxima, one flag per wave
    typename LB::Reload kept;
    {
        float pm[LB::E / 2];
        LB::phase_load_full(opaque(tid), chunk, a, lds, pm, kept);
        float mine = 0.f;
#pragma unroll
        for (int j = 0; j < LB::E / 2; ++j) {
            mine = fmaxf(mine, pm[j]);
            const float m = dpp_max8(pm[j]);
            if ((tid & 7) == 0) LB::block_max(lds)[LB::block_of(tid, j)] = m;
        }
        const bool wave_busy = __any(mine > 0.f) != 0;
        if ((tid & 63) == 0) LB::edge_sl(lds)[tid >> 6] = wave_busy ? 1.f : 0.f;   // (16 floats; the track's ends use them, not these chunks)
    }
    __syncthreads();
    bool chunk_busy = false;
#pragma unroll
    for (int w = 0; w < LB::WAVES; ++w) chunk_busy = chunk_busy || LB::edge_sl(lds)[w] != 0.f;
    DEV_LIFE(1, wall_clock64());
    if (!chunk_busy && a.quiet_ok) limit_chunk_quiet<T>(a, chunk, lds, kept);
    else limit_chunk<T, true>(a, chunk, lds);
    DEV_LIFE(2, wall_clock64());
    // HW_ID (hwreg 4): cu_id bits 11:8, sh 12, se 15:13; XCC_ID (hwreg 20)
    DEV_LIFE(3, (long long)((!chunk_busy && a.quiet_ok) ? 2 : 1) | ((long long)__builtin_amdgcn_s_getreg((15 << 11) | 4) << 8) |
                    ((long long)__builtin_amdgcn_s_getreg((3 << 11) | 20) << 32));
}

}  // namespace mgx
