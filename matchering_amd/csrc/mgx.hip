// libmgx.so -- C ABI (include/mgx.h) over the gfx950 kernels.  No CPU fallback:
// every entry point that needs a GPU fails with MGX_ERR_NO_DEVICE / MGX_ERR_HIP.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#if __has_include(<rocprofiler-sdk-roctx/roctx.h>) && !defined(MGX_NO_ROCTX)
#include <rocprofiler-sdk-roctx/roctx.h>         // one range per stage for rocprofv3 --marker-trace; optional
#else
static inline int roctxRangePushA(const char*) { return 0; }
static inline int roctxRangePop() { return 0; }
#endif

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/mgx.h"
#include "fir_design.h"
#include "fir_plan.h"
#include "host_params.h"
#include "mgx_kernels.h"

using namespace mgx;

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
static thread_local std::string g_error;
static int fail(int code, const std::string& msg) {
    g_error = msg;
    return code;
}
#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(MGX_ERR_HIP, std::string(#expr) + ": " + hipGetErrorName(e_) + " (" + \
                                         hipGetErrorString(e_) + ")");                        \
    } while (0)
#define MGX_TRY(expr)          \
    do {                       \
        int rc_ = (expr);      \
        if (rc_ != 0) return rc_; \
    } while (0)
#define NCCL_TRY(expr)                                                                       \
    do {                                                                                     \
        ncclResult_t r_ = (expr);                                                            \
        if (r_ != ncclSuccess)                                                               \
            return fail(MGX_ERR_RCCL, std::string(#expr) + ": " + ncclGetErrorString(r_));   \
    } while (0)

// ---------------------------------------------------------------------------
// code sizes of the big kernels, from this library's own device code object (mgx_kernels.h, "code warming")
// ---------------------------------------------------------------------------
// libmgx.so -> section .hip_fatbin -> clang offload bundle -> the gfx950 ELF -> .symtab.  Anything unexpected
// (a compressed bundle, a stripped table) leaves the sizes at zero and the kernels do not warm.
#include <dlfcn.h>
#include <elf.h>

#include <fstream>
#include <iterator>

static const char* const CODE_NAMES[CODE_KERNELS] = {"k_analyzeILi", "k_match_curve", "k_conv_prepILi", "k_convILi",
                                                     "k_correction_round", "k_correction_tail", "k_limitILi"};
static bool elf_ok(const std::vector<char>& f, size_t at) {
    return at + sizeof(Elf64_Ehdr) <= f.size() && std::memcmp(f.data() + at, ELFMAG, SELFMAG) == 0 &&
           f[at + EI_CLASS] == ELFCLASS64;
}
// bytes[family][variant]: variant = the first template argument (log2 of the transform; 256 / 1024 blocks of the
// limiter -> 0 / 1), 0 for plain kernels; the smaller size where two instantiations share a variant
static void code_sizes_from_library(int (&bytes)[CODE_KERNELS][CODE_VARIANTS]) {
    for (auto& row : bytes)
        for (int& b : row) b = 0;
    Dl_info info;
    if (!dladdr(reinterpret_cast<const void*>(&code_sizes_from_library), &info) || !info.dli_fname) return;
    std::ifstream in(info.dli_fname, std::ios::binary);
    if (!in) return;
    const std::vector<char> f((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    if (!elf_ok(f, 0)) return;
    const Elf64_Ehdr* eh = reinterpret_cast<const Elf64_Ehdr*>(f.data());
    if (eh->e_shoff + (size_t)eh->e_shnum * sizeof(Elf64_Shdr) > f.size() || eh->e_shstrndx >= eh->e_shnum) return;
    const Elf64_Shdr* sh = reinterpret_cast<const Elf64_Shdr*>(f.data() + eh->e_shoff);
    const char* names = f.data() + sh[eh->e_shstrndx].sh_offset;
    size_t fat = 0, fat_size = 0;
    for (int i = 0; i < eh->e_shnum; ++i)
        if (std::strcmp(names + sh[i].sh_name, ".hip_fatbin") == 0) { fat = sh[i].sh_offset; fat_size = sh[i].sh_size; }
    static const char MAGIC[] = "__CLANG_OFFLOAD_BUNDLE__";
    if (!fat || fat + fat_size > f.size() || fat_size < 32 || std::memcmp(f.data() + fat, MAGIC, 24) != 0) return;
    uint64_t entries = 0;
    std::memcpy(&entries, f.data() + fat + 24, 8);
    size_t pos = fat + 32, dev = 0;
    for (uint64_t e = 0; e < entries && pos + 24 <= fat + fat_size; ++e) {
        uint64_t off = 0, size = 0, tsize = 0;
        std::memcpy(&off, f.data() + pos, 8);
        std::memcpy(&size, f.data() + pos + 8, 8);
        std::memcpy(&tsize, f.data() + pos + 16, 8);
        if (pos + 24 + tsize > fat + fat_size) return;
        const std::string triple(f.data() + pos + 24, f.data() + pos + 24 + tsize);
        if (triple.find("gfx950") != std::string::npos && fat + off + size <= f.size()) dev = fat + off;
        pos += 24 + tsize;
    }
    if (!dev || !elf_ok(f, dev)) return;
    const Elf64_Ehdr* de = reinterpret_cast<const Elf64_Ehdr*>(f.data() + dev);
    if (dev + de->e_shoff + (size_t)de->e_shnum * sizeof(Elf64_Shdr) > f.size()) return;
    const Elf64_Shdr* ds = reinterpret_cast<const Elf64_Shdr*>(f.data() + dev + de->e_shoff);
    for (int i = 0; i < de->e_shnum; ++i) {
        if (ds[i].sh_type != SHT_SYMTAB || ds[i].sh_link >= de->e_shnum) continue;
        const char* str = f.data() + dev + ds[ds[i].sh_link].sh_offset;
        const size_t count = ds[i].sh_size / sizeof(Elf64_Sym);
        const Elf64_Sym* sym = reinterpret_cast<const Elf64_Sym*>(f.data() + dev + ds[i].sh_offset);
        for (size_t k = 0; k < count; ++k) {
            if (ELF64_ST_TYPE(sym[k].st_info) != STT_FUNC || sym[k].st_size == 0) continue;
            const char* name = str + sym[k].st_name;
            if (std::strstr(name, "k_conv_delayILi")) {              // the delay-line convolution: CODE_CONV's last variant
                bytes[CODE_CONV][CODE_VARIANT_CONV_DELAY] = (int)sym[k].st_size;
                continue;
            }
            if (std::strstr(name, "k_conv_wide_prepILi")) {          // ... and its filter preparation
                bytes[CODE_CONV_PREP][CODE_VARIANT_CONV_WIDE] = (int)sym[k].st_size;
                continue;
            }
            if (std::strstr(name, "k_conv_wideILi")) {               // N = 4F: the slot no k_conv<L> uses
                bytes[CODE_CONV][CODE_VARIANT_CONV_WIDE] = (int)sym[k].st_size;
                continue;
            }
            for (int c = 0; c < CODE_KERNELS; ++c) {
                const char* hit = std::strstr(name, CODE_NAMES[c]);
                if (!hit) continue;
                int variant = 0;
                const size_t len = std::strlen(CODE_NAMES[c]);
                if (len >= 3 && std::strcmp(CODE_NAMES[c] + len - 3, "ILi") == 0) {      // templated: ...ILi<number>E
                    const int number = std::atoi(hit + len);
                    variant = number == 256 ? 0 : number == 1024 ? 1 : number;
                }
                if (variant < 0 || variant >= CODE_VARIANTS) continue;
                int& slot = bytes[c][variant];
                if (slot == 0 || (int)sym[k].st_size < slot) slot = (int)sym[k].st_size;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// handle
// ---------------------------------------------------------------------------
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

struct TrackWork {              // per-track analysis workspace + results (device)
    DevBuf wg_sumsq, wg_peak, wg_spec, stats, rms, loud, avg;   // avg: [2][F/2+1] double
    DevBuf wg_pack;                                              // fft_size 65536: AnalysisQuad's scratch, 408 KB per workgroup
    DevBuf part;                                                 // [SPEC_SLICES][2][F/2+1] partial spectrum sums
    int divisions = 0, segs_per_piece = 0, segs_per_wg = 0, chunks = 0, nwg = 0, is_reference = 0;
    long long piece = 0;
};

struct PlanDev {
    void* blob = nullptr;       // FirPlanHost tables
    double* M = nullptr;        // [bins][bins] raw -> smooth operator
    int2* band = nullptr;       // [bins] columns [x, y) of each row that matter (k_fir_band)
    // the same operator in two packed, banded factors (fft_size >= 16384; mgx_kernels.h, k_fir_apply_a / _b)
    struct Factor {
        double* packed = nullptr;       // the rows' windows, one after the other
        FactorRow* rows = nullptr;      // [rows] window of each row and where it starts in `packed`
        size_t bytes = 0;
    } A, B;
    std::shared_ptr<FirPlanHost> plan;      // keeps the host tables alive as long as the device copy
};
// One copy per (device, Config's design parameters) for the whole process, not per handle: the operator is bins^2
// doubles up to fft_size 8192 (34 MB at 4096), two packed factors of a few MB beyond, and the three device handles a
// batch runs on one GPU (batch.py) would otherwise each build and hold their own.  Entries live as long as the
// process: a handle that is destroyed may leave kernels of its siblings reading them.
static std::mutex g_plan_mu;
static std::map<std::pair<int, const FirPlanHost*>, PlanDev> g_plan_dev;

// Limiter launches of one device never run side by side.  k_limit deals its chunks by workgroup number (no atomic
// ticket): a chunk waits for words of lower-numbered chunks only, and every XCD's dispatcher hands out its share of a
// grid in order, so within ONE launch the lowest unfinished chunk is always resident.  Two limiter launches resident
// together break that: launch A's waiting workgroups can fill the XCD on which launch B's lowest chunk would start
// while B's fill the one A needs (seen with two rank PROCESSES sharing a GPU: the bounded waits expired,
// profiles/r05_u_*).  Handles of one process (the two or three lanes of a batch) therefore chain their limiter
// launches through an event per handle -- a limiter fills the chip by itself, nothing is lost -- while every other
// kernel of the lanes still overlaps freely.  Processes that share a GPU cannot see each other: they set
// MGX_LIMIT_TICKETS=1 (bench.py and batch.py do when ranks outnumber GPUs), and a handle whose wait expires all the
// same falls back to tickets by itself (check_device_error).
constexpr int MAX_DEVICES = 64;
struct LimiterChain {
    std::mutex mu;
    hipEvent_t last = nullptr;      // recorded behind the device's most recent limiter launch
    int live = 0;                   // handles alive on the device
};
static LimiterChain g_limiter_chain[MAX_DEVICES];

struct mgx_handle {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool stage_timing = false;                                    // mgx_stage_timing
    hipEvent_t stage_ev[MGX_STAGE_COUNT][2] = {};
    bool stage_used[MGX_STAGE_COUNT] = {};
    std::map<int, float2*> twiddles;
    TrackWork track[2];
    DevBuf y, mid, block_peak, filt, taps, partial, cstate, scalars, conv_queue;
    DevBuf lim_published, lim_ctrl, lim_weights, round_ctr, tail_gains, band, band_info;
    std::vector<double> lim_weights_host;
    DevBuf peak_words;                      // mgx_peak_count: {bits of the maximum, count}
    DevBuf fir_robust;                      // lowess_it > 0: robustness weights and residuals, [2][2][nlog]
    DevBuf lim_tables;                      // general filter orders: matrix powers and look-back matrices
    std::vector<double> lim_tables_host;
    DevBuf fir_scratch;
    void* pinned = nullptr;
    size_t pinned_bytes = 0;
    // set by a kernel whose bounded wait expired (limiter look-back, level-correction round): one word of
    // page-locked host memory the kernels write through its device address, so that every blocking call
    // can look at it for free once the stream has drained
    bool limiter_ran_by_number = false;      // a limiter launch since the last error check dealt chunks by workgroup number
    int* error_host = nullptr;
    int* error_dev = nullptr;
    int last_taps = 0;
    // the arguments of the last mgx_master call, kept so that it can be queued again when the level-correction
    // tail reports that its workgroups were not resident together (check_device_error)
    struct MasterCall {
        bool valid = false;
        const float* target = nullptr;
        const float* reference = nullptr;
        const float* fir_given = nullptr;
        int64_t n_target = 0, n_reference = 0;
        mgx_config cfg;
        float* out[3] = {nullptr, nullptr, nullptr};
    } last_call;
    bool avoid_tail = false;                // sticky after such a report: rounds 1..K-1 as one launch each
    bool requeued = false;                  // the last check_device_error queued the call again
    // the limiter's chunks are its workgroups' numbers; after a look-back wait has expired once on this handle they are
    // drawn from an atomic ticket instead, which does not lean on the dispatch order (k_limit, mgx_kernels.h)
    bool limiter_tickets = false;
    hipEvent_t lim_done = nullptr;                                // behind this handle's latest limiter launch (LimiterChain)
    int masters_outstanding = 0;            // mgx_master calls queued since the last check of the error words
    int downloads_outstanding = 0;          // device-to-host copies queued behind them
    ncclComm_t comm = nullptr;
    int comm_rank = 0, comm_world = 1;
};

static int ensure(mgx_handle* h, DevBuf& b, size_t bytes) {
    if (b.bytes >= bytes && b.p) return 0;
    if (b.p) {
        HIP_TRY(hipStreamSynchronize(h->stream));
        HIP_TRY(hipFree(b.p));
        b.p = nullptr;
        b.bytes = 0;
    }
    const size_t want = std::max(bytes, (size_t)256);
    HIP_TRY(hipMalloc(&b.p, want));
    b.bytes = want;
    return 0;
}
static int ensure_pinned(mgx_handle* h, size_t bytes) {
    if (h->pinned_bytes >= bytes) return 0;
    if (h->pinned) {
        HIP_TRY(hipStreamSynchronize(h->stream));
        HIP_TRY(hipHostFree(h->pinned));
        h->pinned = nullptr;
        h->pinned_bytes = 0;
    }
    HIP_TRY(hipHostMalloc(&h->pinned, bytes, hipHostMallocDefault));
    h->pinned_bytes = bytes;
    return 0;
}

// HIP-event brackets around the stages of mgx_master (mgx_stage_timing / mgx_stage_times): events on
// the handle's stream, so they cost no synchronisation; off by default.
static void stage_mark(mgx_handle* h, int stage, int end) {
    if (!h->stage_timing) return;
    if (!h->stage_ev[stage][0]) {
        hipEventCreate(&h->stage_ev[stage][0]);
        hipEventCreate(&h->stage_ev[stage][1]);
    }
    hipEventRecord(h->stage_ev[stage][end], h->stream);
    if (end) h->stage_used[stage] = true;
}
// (+ a roctx range per stage: `rocprofv3 --marker-trace` shows which launches belong to which stage of
// stages.py:210-272; a push/pop costs nothing when no profiler is attached)
static const char* const STAGE_NAMES[MGX_STAGE_COUNT] = {"mgx:analyze", "mgx:design_fir", "mgx:filter_spectra",
                                                         "mgx:convolve", "mgx:correct_levels", "mgx:scale_outputs",
                                                         "mgx:limit"};
struct StageScope {
    mgx_handle* h;
    int stage;
    StageScope(mgx_handle* h_, int s) : h(h_), stage(s) {
        roctxRangePushA(STAGE_NAMES[stage]);
        stage_mark(h, stage, 0);
    }
    ~StageScope() {
        stage_mark(h, stage, 1);
        roctxRangePop();
    }
};

// control words shared by the kernels that count arrivals: [0] limiter ticket, [1] limiter error flag,
// [4] correction-round arrivals.  Zeroed when allocated; every user leaves its word at zero.
static int ensure_ctrl(mgx_handle* h);

static int get_twiddles(mgx_handle* h, int log2n, const float2** out) {
    auto it = h->twiddles.find(log2n);
    if (it != h->twiddles.end()) {
        *out = it->second;
        return 0;
    }
    const int n = 1 << log2n;
    std::vector<float2> tw(n);
    const double pi = 3.14159265358979323846;
    for (int k = 0; k < n; ++k)
        tw[k] = make_float2((float)std::cos(2.0 * pi * k / n), (float)-std::sin(2.0 * pi * k / n));
    float2* d = nullptr;
    HIP_TRY(hipMalloc((void**)&d, n * sizeof(float2)));
    HIP_TRY(hipMemcpy(d, tw.data(), n * sizeof(float2), hipMemcpyHostToDevice));
    h->twiddles[log2n] = d;
    *out = d;
    return 0;
}

template <typename K>
static int allow_lds(K kernel, size_t bytes) {
    if (bytes > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return 0;
}

static int ensure_ctrl(mgx_handle* h) {
    if (h->lim_ctrl.p) return 0;
    MGX_TRY(ensure(h, h->lim_ctrl, 64));
    HIP_TRY(hipMemsetAsync(h->lim_ctrl.p, 0, 64, h->stream));
    return 0;
}

static int check_config(const mgx_config* c) {
    if (!c) return fail(MGX_ERR_ARGUMENT, "config is null");
    if (c->internal_sample_rate <= 0) return fail(MGX_ERR_ARGUMENT, "internal_sample_rate must be positive");
    if (ilog2_exact(c->fft_size) < 0) return fail(MGX_ERR_ARGUMENT, "fft_size must be a power of two");
    if (c->fft_size < 8)          /* (defaults.py:110-112 lets 2 and 4 through; match_frequencies.py:45-58 then fails) */
        return fail(MGX_ERR_ARGUMENT, "fft_size below 8: the reference's own cubic interpolation of the matching curve "
                                      "needs at least four points per side and fails there");
    if (c->fft_size > 65536)
        return fail(MGX_ERR_UNSUPPORTED, "fft_size above 65536 is not implemented "
                                         "(an analysis segment is one, two or four transforms that fit one CU's LDS)");
    if (c->rms_correction_steps < 0) return fail(MGX_ERR_ARGUMENT, "rms_correction_steps must not be negative");
    if (c->rms_correction_steps > 4096)      /* (a flag word per round and summing workgroup: 4 MB at 4096) */
        return fail(MGX_ERR_UNSUPPORTED, "more than 4096 rms_correction_steps are not implemented");
    if (c->lowess_it < 0 || c->lowess_it > 64) return fail(MGX_ERR_ARGUMENT, "lowess_it outside [0, 64]");
    if (!(c->threshold > c->min_value && c->threshold < 1.0 && c->min_value > 0.0))
        return fail(MGX_ERR_ARGUMENT, "threshold/min_value out of range (defaults.py:93-99)");
    if (!(c->max_piece_size > c->fft_size)) return fail(MGX_ERR_ARGUMENT, "max_piece_size must exceed fft_size samples");
    return 0;
}

// Kernels address a track through a buffer view with 32-bit byte offsets (mgx_hd.h MemView): 8 bytes
// per frame plus the look-ahead of a convolution block must stay below 4 GiB.  That is 3.3 hours at
// 44.1 kHz; the reference stops at max_length = 15 minutes by default (defaults.py, checker.py:58).
static int check_length(long long n) {
    const long long max_frames = (0xfffffff0ll >> 3) - (1 << 16);
    if (n > max_frames) return fail(MGX_ERR_UNSUPPORTED, "tracks longer than 536 million frames are not implemented");
    return 0;
}

// ---------------------------------------------------------------------------
// stage runners (asynchronous on h->stream)
// ---------------------------------------------------------------------------
// workgroups of k_analyze<log2f> one CU holds (LDS and wave slots)
static int analysis_workgroups_per_cu(int log2f) {
    size_t lds = 0;
    int threads = 64;
    if (log2f < 6) return 8;                                     // k_analyze_small: 256 threads, a few hundred bytes of LDS
    switch (log2f) {
#define CASE(L) case L: lds = analysis_lds_bytes<L>(); threads = Fft2<L>::T; break;
        CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13) CASE(14)
#undef CASE
        case 15:                                                                     // two 16384-point transforms per segment
        case 16: lds = analysis_lds_bytes<14>(); threads = Fft2<14>::T; break;       // four
        default: return 1;
    }
    const int by_lds = (int)((size_t)160 * 1024 / lds), by_waves = 2048 / threads;
    return std::max(1, std::min(MGX_ANALYZE_MAX_WGS, std::min(by_lds, by_waves)));
}

template <int LOG2N>
static int launch_analysis(mgx_handle* h, const AnalysisArgs& a0, const AnalysisArgs& a1, int nwg0, int nwg) {
    const size_t lds = analysis_lds_bytes<LOG2N>();
    MGX_TRY(allow_lds(k_analyze<LOG2N>, lds));
    hipLaunchKernelGGL(k_analyze<LOG2N>, dim3(nwg), dim3(Fft2<LOG2N>::T), lds, h->stream, a0, a1, nwg0);
    HIP_TRY(hipGetLastError());
    return 0;
}

// geometry of a track's analysis (match_levels.py:47-59) and the workspace it fills; chunks per piece
// are chosen by the caller (they depend on what shares the launch)
static int plan_analysis(mgx_handle* h, long long n, const mgx_config* cfg, int is_reference, TrackWork& w) {
    const int f = cfg->fft_size;
    if (n <= f) return fail(MGX_ERR_ARGUMENT, "track must be longer than fft_size frames (core.py:69-74)");
    MGX_TRY(check_length(n));
    piece_geometry(n, cfg->max_piece_size, w.divisions, w.piece);
    w.segs_per_piece = (int)(w.piece / f);
    if (w.segs_per_piece < 1)
        return fail(MGX_ERR_UNSUPPORTED, "analysis pieces shorter than fft_size are not implemented");
    w.is_reference = is_reference;
    return 0;
}

// Every workgroup runs its segments back to back and a CU's residents share its throughput, so the
// kernel lasts about as long as the busiest CU has segments: ceil(workgroups / CUs) * segments per
// workgroup.  Pick the segments per workgroup that minimise it over ALL tracks of the launch (ties:
// more workgroups); a second dispatch wave of a few stragglers would double the kernel's duration.
static int choose_chunks(mgx_handle* h, const mgx_config* cfg, TrackWork* const* tracks, int count) {
    int dev_cus = 256;
    HIP_TRY(hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, h->device));
    const int per_cu = analysis_workgroups_per_cu(ilog2_exact(cfg->fft_size));
    int longest = 1;
    for (int t = 0; t < count; ++t) longest = std::max(longest, tracks[t]->segs_per_piece);
    long long best_cost = -1;
    int best_s = longest;
    for (int s = 1; s <= longest; ++s) {
        long long wgs = 0;
        for (int t = 0; t < count; ++t)
            wgs += (long long)tracks[t]->divisions * ((tracks[t]->segs_per_piece + s - 1) / s);
        const long long deep = (wgs + dev_cus - 1) / dev_cus;
        if (deep > per_cu) continue;
        const long long cost = deep * s;
        if (best_cost < 0 || cost < best_cost) {
            best_cost = cost;
            best_s = s;
        }
    }
    for (int t = 0; t < count; ++t) {
        TrackWork& w = *tracks[t];
        w.chunks = (w.segs_per_piece + best_s - 1) / best_s;
        w.nwg = w.divisions * w.chunks;
        const int half = cfg->fft_size / 2;
        MGX_TRY(ensure(h, w.wg_sumsq, (size_t)w.nwg * sizeof(double)));
        MGX_TRY(ensure(h, w.wg_peak, (size_t)w.nwg * sizeof(float)));
        MGX_TRY(ensure(h, w.wg_spec, (size_t)w.nwg * 2 * (half + 1) * sizeof(float)));
        if (cfg->fft_size == 65536)
            MGX_TRY(ensure(h, w.wg_pack, (size_t)w.nwg * AnalysisQuad<14>::SCRATCH_FLOAT2 * sizeof(float2)));
        MGX_TRY(ensure(h, w.stats, sizeof(TrackStats)));
        MGX_TRY(ensure(h, w.rms, (size_t)w.divisions * sizeof(double)));
        MGX_TRY(ensure(h, w.loud, (size_t)w.divisions * sizeof(int)));
        MGX_TRY(ensure(h, w.avg, (size_t)2 * (half + 1) * sizeof(double)));
    }
    return 0;
}

static int analysis_args(mgx_handle* h, const float* x, long long n, const mgx_config* cfg, const TrackWork& w,
                         AnalysisArgs& a) {
    a.x = reinterpret_cast<const float2*>(x);
    a.n = n;
    a.fft = cfg->fft_size;
    a.piece = w.piece;
    a.divisions = w.divisions;
    a.segs_per_piece = w.segs_per_piece;
    a.chunks_per_piece = w.chunks;
    a.wg_sumsq = (double*)w.wg_sumsq.p;
    a.wg_peak = (float*)w.wg_peak.p;
    a.wg_spec = (float*)w.wg_spec.p;
    a.wg_pack = (float2*)w.wg_pack.p;
    a.tw = nullptr;
    if (cfg->fft_size < 64) return 0;                            // k_analyze_small transforms in registers: no table
    // (fft_size 32768 and 65536 run on 16384-point transforms: AnalysisDouble, AnalysisQuad)
    return get_twiddles(h, std::min(14, ilog2_exact(cfg->fft_size)), &a.tw);
}

// one launch for one track (second == nullptr) or for the target and the reference of a pair
static int run_analysis(mgx_handle* h, const mgx_config* cfg, const float* x0, long long n0, TrackWork& w0,
                        const float* x1 = nullptr, long long n1 = 0, TrackWork* w1 = nullptr) {
    MGX_TRY(plan_analysis(h, n0, cfg, w0.is_reference, w0));
    if (w1) MGX_TRY(plan_analysis(h, n1, cfg, w1->is_reference, *w1));
    TrackWork* tracks[2] = {&w0, w1};
    MGX_TRY(choose_chunks(h, cfg, tracks, w1 ? 2 : 1));
    AnalysisArgs a0, a1;
    MGX_TRY(analysis_args(h, x0, n0, cfg, w0, a0));
    a1 = a0;
    if (w1) MGX_TRY(analysis_args(h, x1, n1, cfg, *w1, a1));
    const int nwg = w0.nwg + (w1 ? w1->nwg : 0);
    switch (ilog2_exact(cfg->fft_size)) {
#define SMALL(L) case L: hipLaunchKernelGGL(k_analyze_small<L>, dim3(nwg), dim3(256), 0, h->stream, a0, a1, w0.nwg); HIP_TRY(hipGetLastError()); break;
        SMALL(3) SMALL(4) SMALL(5)
#undef SMALL
#define CASE(L) case L: MGX_TRY(launch_analysis<L>(h, a0, a1, w0.nwg, nwg)); break;
        CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13) CASE(14)
#undef CASE
        case 15: {
            const size_t lds = analysis_lds_bytes<14>();
            MGX_TRY(allow_lds(k_analyze_double<14>, lds));
            hipLaunchKernelGGL(k_analyze_double<14>, dim3(nwg), dim3(Fft2<14>::T), lds, h->stream, a0, a1, w0.nwg);
            HIP_TRY(hipGetLastError());
            break;
        }
        case 16: {
            const size_t lds = analysis_lds_bytes<14>();
            MGX_TRY(allow_lds(k_analyze_quad<14>, lds));
            hipLaunchKernelGGL(k_analyze_quad<14>, dim3(nwg), dim3(Fft2<14>::T), lds, h->stream, a0, a1, w0.nwg);
            HIP_TRY(hipGetLastError());
            break;
        }
        default: return fail(MGX_ERR_UNSUPPORTED, "fft_size not supported by the analysis kernel");
    }
    return 0;
}

static LevelsArgs levels_args(const TrackWork& w) {
    LevelsArgs a;
    a.wg_sumsq = (const double*)w.wg_sumsq.p;
    a.wg_peak = (const float*)w.wg_peak.p;
    a.chunks_per_piece = w.chunks;
    a.divisions = w.divisions;
    a.piece = w.piece;
    a.is_reference = w.is_reference;
    a.st = (TrackStats*)w.stats.p;
    a.rms = (double*)w.rms.p;
    a.loud = (int*)w.loud.p;
    return a;
}
static SpectraArgs spectra_args(const TrackWork& w) {
    SpectraArgs a;
    a.wg_spec = (const float*)w.wg_spec.p;
    a.loud = (const int*)w.loud.p;
    a.chunks_per_piece = w.chunks;
    a.nwg = w.nwg;
    a.part = (double*)w.part.p;
    return a;
}
// piece statistics -> decisions, then the loud pieces' spectrum sums; one launch each for 1 or 2 tracks
static int run_levels(mgx_handle* h, const mgx_config* cfg, TrackWork* first, TrackWork* second) {
    const int tracks = second ? 2 : 1, half = cfg->fft_size / 2;
    const int max_div = std::max(first->divisions, second ? second->divisions : 0);
    const size_t lds = (size_t)(64 + max_div) * sizeof(double);
    if (lds > 150 * 1024) return fail(MGX_ERR_UNSUPPORTED, "too many analysis pieces");
    MGX_TRY(allow_lds(k_levels, lds));
    for (TrackWork* w : {first, second})
        if (w) MGX_TRY(ensure(h, w->part, (size_t)SPEC_SLICES * 2 * (half + 1) * sizeof(double)));
    const LevelsArgs l0 = levels_args(*first), l1 = second ? levels_args(*second) : l0;
    hipLaunchKernelGGL(k_levels, dim3(tracks), dim3(1024), lds, h->stream, l0, l1, cfg->threshold, cfg->min_value);
    const SpectraArgs s0 = spectra_args(*first), s1 = second ? spectra_args(*second) : s0;
    hipLaunchKernelGGL(k_average_spectra, dim3((half + 1 + 63) / 64, 2, SPEC_SLICES * tracks), dim3(1024), 0, h->stream,
                       s0, s1, half + 1);
    HIP_TRY(hipGetLastError());
    return 0;
}

// device-side FIR design (fir_plan.h): spectra partial sums of both tracks -> h->taps ([2][F] float),
// level gain c0 -> h->scalars[0].  No host synchronisation.  The chain raw -> smooth is one dense
// operator per plan (mgx_kernels.h), built on first use.
static int build_fir_operator(mgx_handle* h, const FirPlanView& pl, double** out, int2** band_out) {
    const size_t per = (size_t)3 * pl.bins + (size_t)3 * pl.nlog + pl.lw.anchors;
    const int batch = std::min(pl.bins, 256);
    double* scratch = nullptr;
    double* M = nullptr;
    HIP_TRY(hipMalloc((void**)&scratch, (size_t)batch * per * sizeof(double)));
    HIP_TRY(hipMalloc((void**)&M, (size_t)pl.bins * pl.bins * sizeof(double)));
    const size_t lds_scan = (size_t)FirDesign::Scan::SCRATCH * sizeof(Affine);
    for (int col0 = 0; col0 < pl.bins; col0 += batch) {
        const int nb = std::min(batch, pl.bins - col0);
        hipLaunchKernelGGL(k_fir_unit_a, dim3(nb), dim3(1024), lds_scan, h->stream, pl, scratch, col0);
        hipLaunchKernelGGL(k_fir_lowess, dim3((pl.lw.anchors + 15) / 16, nb), dim3(1024), 0, h->stream, pl, scratch);
        hipLaunchKernelGGL(k_fir_b, dim3(nb), dim3(1024), lds_scan, h->stream, pl, scratch);
        hipLaunchKernelGGL(k_fir_gather, dim3((nb + 255) / 256, pl.bins), dim3(256), 0, h->stream, pl, scratch, col0,
                           nb, M);
    }
    int2* band = nullptr;
    HIP_TRY(hipMalloc((void**)&band, (size_t)pl.bins * sizeof(int2)));
    hipLaunchKernelGGL(k_fir_band, dim3(pl.bins), dim3(256), 0, h->stream, (const double*)M, pl.bins, band);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipFree(scratch));
    *out = M;
    *band_out = band;
    return 0;
}

// device temporaries of the operator builds: freed on every way out, the failing ones included
struct DevTemp {
    void* p = nullptr;
    ~DevTemp() { if (p) hipFree(p); }
    DevTemp() = default;
    DevTemp(const DevTemp&) = delete;
    DevTemp& operator=(const DevTemp&) = delete;
};
constexpr size_t FIR_FACTOR_DENSE_BUDGET = (size_t)1 << 30;      // 1 GiB for a factor's dense intermediate

// A dense [rows][cols] matrix -> its rows' windows (k_fir_band), packed.  The dense matrix stays the caller's.
static int pack_fir_factor(mgx_handle* h, double* dense, int rows, int cols, PlanDev::Factor& f) {
    DevTemp band_tmp;
    HIP_TRY(hipMalloc(&band_tmp.p, (size_t)rows * sizeof(int2)));
    int2* band_dev = (int2*)band_tmp.p;
    hipLaunchKernelGGL(k_fir_band, dim3(rows), dim3(256), 0, h->stream, (const double*)dense, cols, band_dev);
    HIP_TRY(hipGetLastError());
    std::vector<int2> band(rows);
    HIP_TRY(hipMemcpyAsync(band.data(), band_dev, (size_t)rows * sizeof(int2), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    std::vector<FactorRow> desc(rows);
    long long total = 0;
    for (int r = 0; r < rows; ++r) {
        desc[r] = FactorRow{band[r].x, band[r].y, total};
        total += (band[r].y - band[r].x + 1) & ~1;               // (rows start on 16-byte boundaries)
    }
    f.bytes = (size_t)std::max<long long>(total, 2) * sizeof(double);
    HIP_TRY(hipMalloc((void**)&f.rows, (size_t)rows * sizeof(FactorRow)));
    HIP_TRY(hipMalloc((void**)&f.packed, f.bytes));
    HIP_TRY(hipMemcpyAsync(f.rows, desc.data(), (size_t)rows * sizeof(FactorRow), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemsetAsync(f.packed, 0, f.bytes, h->stream));
    hipLaunchKernelGGL(k_fir_pack, dim3(rows), dim3(256), 0, h->stream, (const double*)dense, cols, (const FactorRow*)f.rows,
                       f.packed);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));
    return 0;
}

// raw -> smooth as B * (A * raw) through the anchors' LOWESS fits (mgx_kernels.h): unit vectors pushed through the two
// halves of the chain, 256 at a time, gathered into dense matrices that live only until their windows are packed
static int build_fir_factors(mgx_handle* h, const FirPlanView& pl, PlanDev& pd) {
    const size_t per = (size_t)3 * pl.bins + (size_t)3 * pl.nlog + pl.lw.anchors;
    const int anchors = pl.lw.anchors, batch = 256;
    DevTemp scratch_tmp, dense_tmp;
    HIP_TRY(hipMalloc(&scratch_tmp.p, (size_t)batch * per * sizeof(double)));
    double* scratch = (double*)scratch_tmp.p;
    const size_t lds_scan = (size_t)FirDesign::Scan::SCRATCH * sizeof(Affine);
    // A: unit raw curves -> the anchors' fits
    HIP_TRY(hipMalloc(&dense_tmp.p, (size_t)anchors * pl.bins * sizeof(double)));
    double* dense = (double*)dense_tmp.p;
    for (int col0 = 0; col0 < pl.bins; col0 += batch) {
        const int nb = std::min(batch, pl.bins - col0);
        hipLaunchKernelGGL(k_fir_unit_a, dim3(nb), dim3(1024), lds_scan, h->stream, pl, scratch, col0);
        hipLaunchKernelGGL(k_fir_lowess, dim3((anchors + 15) / 16, nb), dim3(1024), 0, h->stream, pl, scratch);
        hipLaunchKernelGGL(k_fir_gather_plane, dim3((nb + 255) / 256, anchors), dim3(256), 0, h->stream, pl, scratch, col0,
                           nb, pl.bins, 1, dense);
    }
    HIP_TRY(hipGetLastError());
    MGX_TRY(pack_fir_factor(h, dense, anchors, pl.bins, pd.A));
    // B: unit fits -> the smooth curve on the linear grid (the same bytes: bins x anchors)
    for (int col0 = 0; col0 < anchors; col0 += batch) {
        const int nb = std::min(batch, anchors - col0);
        hipLaunchKernelGGL(k_fir_unit_fit, dim3(nb), dim3(256), 0, h->stream, pl, scratch, col0);
        hipLaunchKernelGGL(k_fir_b, dim3(nb), dim3(1024), lds_scan, h->stream, pl, scratch);
        hipLaunchKernelGGL(k_fir_gather_plane, dim3((nb + 255) / 256, pl.bins), dim3(256), 0, h->stream, pl, scratch, col0,
                           nb, anchors, 0, dense);
    }
    HIP_TRY(hipGetLastError());
    MGX_TRY(pack_fir_factor(h, dense, pl.bins, anchors, pd.B));
    return 0;
}

// fir_given: a FIR pair to use instead of the designed one (album mode), or null
static int run_fir_design(mgx_handle* h, const mgx_config* cfg, const TrackWork& tw, const TrackWork& rw,
                          const float* fir_given) {
    FirDesignParams p{cfg->fft_size, cfg->internal_sample_rate, cfg->lin_log_oversampling, cfg->lowess_frac,
                      cfg->lowess_it, cfg->lowess_delta, cfg->min_value};
    std::shared_ptr<FirPlanHost> plan = FirPlanHost::get(p);
    // no operator when LOWESS is not linear (robustness passes): the chain runs on the curve itself.  Otherwise the
    // dense operator up to fft_size 8192 (134 MB there, of which a product reads a sixth), its two packed factors
    // from 16384 on (MGX_FIR_ROUND4=1: the choices of round 4, for A/B measurements)
    const bool robust = cfg->lowess_it > 0;
    const char* env_old = std::getenv("MGX_FIR_ROUND4");
    const bool round4 = env_old && env_old[0] == '1';            // dense at 16384, the chain itself beyond
    // (the factors are found by pushing unit vectors through the chain into two DENSE anchors x bins matrices before they
    // are packed: with lowess_delta = 0 every point of the log grid is an anchor -- 8.6 GB at fft_size 32768 -- so the
    // factored form needs that intermediate to fit a budget; beyond it the chain runs on the curve itself, as in round 4)
    const size_t dense_bytes = (size_t)plan->anchors() * (size_t)plan->bins() * sizeof(double);
    const bool factored = !robust && plan->bins() > 4097 && !round4 && dense_bytes <= FIR_FACTOR_DENSE_BUDGET;
    const bool direct = robust || (!factored && plan->bins() > 8193);
    PlanDev pd;
    {
        // (the first handle to need an operator builds it on its own stream and waits for it; its siblings
        // wait here and find it complete)
        std::lock_guard<std::mutex> lock(g_plan_mu);
        PlanDev& shared = g_plan_dev[std::make_pair(h->device, (const FirPlanHost*)plan.get())];
        if (!shared.blob) {
            HIP_TRY(hipMalloc(&shared.blob, plan->blob_bytes()));
            HIP_TRY(hipMemcpy(shared.blob, plan->blob(), plan->blob_bytes(), hipMemcpyHostToDevice));
            shared.plan = plan;
        }
        if (factored) {
            if (!shared.A.packed || !shared.B.packed) {
                // (a build that failed half-way leaves nothing behind: both factors or neither)
                for (PlanDev::Factor* f : {&shared.A, &shared.B}) {
                    if (f->rows) hipFree(f->rows);
                    if (f->packed) hipFree(f->packed);
                    *f = PlanDev::Factor();
                }
                MGX_TRY(build_fir_factors(h, plan->view(shared.blob), shared));
            }
        } else if (!direct && !shared.M) {
            MGX_TRY(build_fir_operator(h, plan->view(shared.blob), &shared.M, &shared.band));
        }
        pd = shared;
    }
    const FirPlanView pl = plan->view(pd.blob);
    const size_t per = (size_t)3 * pl.bins + (size_t)3 * pl.nlog + pl.lw.anchors;
    // (two scratch planes, the raw curves, and the partial transforms of the tap synthesis: [2][fft/2] double2)
    MGX_TRY(ensure(h, h->fir_scratch, (2 * per + 2 * (size_t)pl.bins + 2 * (size_t)pl.fft + 2) * sizeof(double)));
    MGX_TRY(ensure(h, h->scalars, 64));
    MGX_TRY(ensure(h, h->taps, (size_t)2 * cfg->fft_size * sizeof(float)));
    FirInputs in;
    in.part_t = (const double*)tw.part.p;
    in.part_r = (const double*)rw.part.p;
    in.st_t = (const TrackStats*)tw.stats.p;
    in.st_r = (const TrackStats*)rw.stats.p;
    in.segs_t = tw.segs_per_piece;
    in.segs_r = rw.segs_per_piece;
    in.eps = cfg->min_value;
    double* scratch = (double*)h->fir_scratch.p;
    double* raw = scratch + 2 * per;
    MGX_TRY(ensure(h, h->cstate, sizeof(CorrectionState)));
    // piece decisions of both tracks, loud-piece spectra and the raw curve: one launch while the piece
    // tables fit a workgroup's LDS (always, short of thousands of pieces), three otherwise
    const int max_div = std::max(tw.divisions, rw.divisions);
    const size_t lds_curve = match_curve_lds_bytes(max_div, tw.nwg + rw.nwg);
    if (lds_curve <= (size_t)150 * 1024) {
        CurveTrack ct{levels_args(tw), (const float*)tw.wg_spec.p, tw.nwg, tw.segs_per_piece};
        CurveTrack cr{levels_args(rw), (const float*)rw.wg_spec.p, rw.nwg, rw.segs_per_piece};
        // tiles of 33 bins where they save a round of workgroups (one workgroup of 1024 threads per CU)
        int dev_cus = 256;
        HIP_TRY(hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, h->device));
        auto rounds = [&](int tile) { return (2 * ((pl.bins + tile - 1) / tile) + dev_cus - 1) / dev_cus; };
        const char* tile32 = std::getenv("MGX_CURVE_TILE32");                    // (A/B: always 32)
        if (rounds(33) < rounds(32) && !(tile32 && tile32[0] == '1')) {
            MGX_TRY(allow_lds(k_match_curve<33>, lds_curve));
            hipLaunchKernelGGL(k_match_curve<33>, dim3((pl.bins + 32) / 33, 2), dim3(1024), lds_curve, h->stream, ct, cr,
                               pl.bins, pl.fft, max_div, cfg->threshold, cfg->min_value, pl.min_value, raw,
                               (double*)h->scalars.p, (CorrectionState*)h->cstate.p, h->error_dev);
        } else {
            MGX_TRY(allow_lds(k_match_curve<32>, lds_curve));
            hipLaunchKernelGGL(k_match_curve<32>, dim3((pl.bins + 31) / 32, 2), dim3(1024), lds_curve, h->stream, ct, cr,
                               pl.bins, pl.fft, max_div, cfg->threshold, cfg->min_value, pl.min_value, raw,
                               (double*)h->scalars.p, (CorrectionState*)h->cstate.p, h->error_dev);
        }
    } else {
        TrackWork& t = const_cast<TrackWork&>(tw);
        TrackWork& r = const_cast<TrackWork&>(rw);
        MGX_TRY(run_levels(h, cfg, &t, &r));
        in.part_t = (const double*)tw.part.p;
        in.part_r = (const double*)rw.part.p;
        hipLaunchKernelGGL(k_fir_raw, dim3((pl.bins + 255) / 256, 2), dim3(256), 0, h->stream, pl, in, raw,
                           (double*)h->scalars.p, (CorrectionState*)h->cstate.p);
    }
    if (fir_given) {             // the levels above are this pair's own; the matching EQ is somebody else's
        if (fir_given != (const float*)h->taps.p)
            HIP_TRY(hipMemcpyAsync(h->taps.p, fir_given, (size_t)2 * cfg->fft_size * sizeof(float),
                                   hipMemcpyDeviceToDevice, h->stream));
        h->last_taps = cfg->fft_size;
        return 0;
    }
    if (direct) {
        const size_t lds_scan = (size_t)FirDesign::Scan::SCRATCH * sizeof(Affine);
        hipLaunchKernelGGL(k_fir_direct_a, dim3(2), dim3(1024), lds_scan, h->stream, pl, scratch, (const double*)raw);
        if (robust) {
            MGX_TRY(ensure(h, h->fir_robust, (size_t)4 * pl.nlog * sizeof(double)));
            hipLaunchKernelGGL(k_fir_lowess_robust, dim3(2), dim3(1024), 0, h->stream, pl, scratch,
                               (double*)h->fir_robust.p, cfg->lowess_it);
        } else {
            hipLaunchKernelGGL(k_fir_lowess, dim3((pl.lw.anchors + 15) / 16, 2), dim3(1024), 0, h->stream, pl, scratch);
        }
        hipLaunchKernelGGL(k_fir_b, dim3(2), dim3(1024), lds_scan, h->stream, pl, scratch);
    } else if (factored) {
        hipLaunchKernelGGL(k_fir_apply_a, dim3(pl.lw.anchors), dim3(256), 0, h->stream, pl, (const double*)pd.A.packed,
                           (const FactorRow*)pd.A.rows, (const double*)raw, scratch);
        hipLaunchKernelGGL(k_fir_apply_b, dim3((pl.bins + 255) / 256), dim3(256), 0, h->stream, pl, (const double*)pd.B.packed,
                           (const FactorRow*)pd.B.rows, (const double*)raw, scratch);
    } else {
        hipLaunchKernelGGL(k_fir_matvec, dim3(pl.bins), dim3(256), 0, h->stream, pl, (const double*)pd.M,
                           (const int2*)pd.band, (const double*)raw, scratch);
    }
    if (pl.fft < 64) {                                          // 8 .. 32 taps: the plain cosine sum, one thread per tap
        hipLaunchKernelGGL(k_fir_taps_direct, dim3(2), dim3(1024), 0, h->stream, pl, (const double*)scratch, (float*)h->taps.p);
        HIP_TRY(hipGetLastError());
        h->last_taps = cfg->fft_size;
        return 0;
    }
    // tap synthesis: the symmetric cosine sum up to 4096 taps (8 us in one launch -- two launches of the split
    // transform cost 10), the split transform from 8192 taps on (13 us against 47 at 16384;
    // profiles/r03_x_tap_synthesis.txt).  MGX_TAPS_BY_COSINE_SUM=1 forces the sum: the A/B switch of that profile.
    if (pl.fft >= TAP_TRANSFORM_FROM && !std::getenv("MGX_TAPS_BY_COSINE_SUM")) {
        // (65536 taps: sixteen sub-transforms of 2048 points -- the longest one instantiated)
        const int split = pl.fft > 32768 ? 2 * TAP_SPLIT : TAP_SPLIT;
        const int m = pl.fft / 2 / split;
        const size_t sub_at = (2 * per + 2 * (size_t)pl.bins + 1) & ~(size_t)1;      // 16-byte aligned
        double2* sub = reinterpret_cast<double2*>(scratch + sub_at);
        const size_t lds_sub = fir_taps_sub_lds_bytes(m);
        switch (m) {
#define MGX_TAPS_SUB(L)                                                                                                \
    case 1 << L:                                                                                                       \
        hipLaunchKernelGGL(k_fir_taps_sub<L>, dim3(split, 2), dim3(TapFft<L>::T), lds_sub, h->stream, pl,              \
                           (const double*)scratch, sub);                                                               \
        break;
            MGX_TAPS_SUB(9) MGX_TAPS_SUB(10) MGX_TAPS_SUB(11)
#undef MGX_TAPS_SUB
            default: return fail(MGX_ERR_UNSUPPORTED, "tap synthesis: transform length not instantiated");
        }
        hipLaunchKernelGGL(k_fir_taps_combine, dim3((pl.fft / 2 + 255) / 256, 2), dim3(256), 0, h->stream, pl,
                           (const double2*)sub, split, (float*)h->taps.p);
    } else {
        const size_t lds_taps = ((size_t)pl.bins + 2048 + 2 * TAP_ROWS) * sizeof(double);
        MGX_TRY(allow_lds(k_fir_taps, lds_taps));
        hipLaunchKernelGGL(k_fir_taps, dim3((pl.fft / 4 + 1 + TAP_ROWS - 1) / TAP_ROWS, 2), dim3(1024), lds_taps, h->stream,
                           pl, (const double*)scratch, (float*)h->taps.p);
    }
    HIP_TRY(hipGetLastError());
    h->last_taps = cfg->fft_size;
    return 0;
}

template <int LOG2N, bool MULTI>
static int launch_conv(mgx_handle* h, Conv2Args a, const float* taps_dev, double gain, const double* gain_ptr) {
    using F = Fft2<LOG2N>;
    const size_t lds = conv_lds_bytes<LOG2N>();
    MGX_TRY((allow_lds(k_conv_prep<LOG2N>, lds)));
    MGX_TRY((allow_lds(k_conv<LOG2N, MULTI>, lds)));
    {
        StageScope scope(h, MGX_STAGE_FILTER_SPECTRA);
        hipLaunchKernelGGL((k_conv_prep<LOG2N>), dim3(2 * a.parts), dim3(F::T), lds, h->stream, taps_dev,
                           a.tw, (float2*)h->filt.p, a.parts, gain_ptr, gain);
    }
    HIP_TRY(hipGetLastError());
    MGX_TRY(ensure(h, h->block_peak, (size_t)a.npairs * sizeof(float)));
    a.pair_peak = (float*)h->block_peak.p;
    if (!h->conv_queue.p) {                  // zero once: every launch leaves the counters at zero
        MGX_TRY(ensure(h, h->conv_queue, 64));
        HIP_TRY(hipMemsetAsync(h->conv_queue.p, 0, 64, h->stream));
    }
    a.queue = (unsigned*)h->conv_queue.p;
    // persistent grid: as many workgroups as fit the chip at once (LDS-limited), a multiple of 8
    int dev_cus = 256;
    HIP_TRY(hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, h->device));
    const int per_cu = std::max(1, std::min(2048 / F::T, (int)((size_t)160 * 1024 / lds)));
    const long long cap = (long long)dev_cus * per_cu;
    const unsigned grid = (unsigned)(((std::min<long long>(a.npairs, cap) + 7) / 8) * 8);
    {
        StageScope scope(h, MGX_STAGE_CONVOLVE);
        hipLaunchKernelGGL((k_conv<LOG2N, MULTI>), dim3(grid), dim3(F::T), lds, h->stream, a);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

// taps = N in two partitions (config #5: 16384 taps on N = 16384 blocks): the frequency-domain delay line of
// conv_delay_kernel.h, a run of consecutive blocks per workgroup (one workgroup per CU: every run costs one extra
// forward transform, so runs are as long as the chip allows).  a.npairs counts blocks on return.
template <int LOG2N>
static int launch_conv_delay(mgx_handle* h, Conv2Args a, const float* taps_dev, double gain, const double* gain_ptr) {
    using F = Fft2<LOG2N>;
    const size_t lds = conv_lds_bytes<LOG2N>();
    MGX_TRY((allow_lds(k_conv_prep<LOG2N>, lds)));
    MGX_TRY((allow_lds(k_conv_delay<LOG2N>, lds)));
    {
        StageScope scope(h, MGX_STAGE_FILTER_SPECTRA);
        hipLaunchKernelGGL((k_conv_prep<LOG2N>), dim3(2 * a.parts), dim3(F::T), lds, h->stream, taps_dev,
                           a.tw, (float2*)h->filt.p, a.parts, gain_ptr, gain);
    }
    HIP_TRY(hipGetLastError());
    a.npairs = (a.n + ConvDelay<LOG2N>::HOP - 1) / ConvDelay<LOG2N>::HOP;
    MGX_TRY(ensure(h, h->block_peak, (size_t)a.npairs * sizeof(float)));
    a.pair_peak = (float*)h->block_peak.p;
    a.queue = nullptr;
    int dev_cus = 256;
    HIP_TRY(hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, h->device));
    const int per_cu = std::max(1, std::min(2048 / F::T, (int)((size_t)160 * 1024 / lds)));
    const long long cap = (long long)dev_cus * per_cu;
    a.run = (int)std::max<long long>(1, (a.npairs + cap - 1) / cap);
    const unsigned grid = (unsigned)((a.npairs + a.run - 1) / a.run);
    {
        StageScope scope(h, MGX_STAGE_CONVOLVE);
        hipLaunchKernelGGL((k_conv_delay<LOG2N>), dim3(grid), dim3(F::T), lds, h->stream, a);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

// F taps on N = 4F blocks (conv_wide_kernel.h; F = 4096 on N = 16384, one workgroup per CU, blocks dealt round-robin).
// a.npairs counts blocks of 3N/4 frames on return.
template <int LOG2N>
static int launch_conv_wide(mgx_handle* h, Conv2Args a, const float* taps_dev, double gain, const double* gain_ptr) {
    using F = Fft2<LOG2N>;
    const size_t lds = conv_lds_bytes<LOG2N>();
    MGX_TRY((allow_lds(k_conv_wide_prep<LOG2N>, lds)));
    MGX_TRY((allow_lds(k_conv_wide<LOG2N>, lds)));
    {
        StageScope scope(h, MGX_STAGE_FILTER_SPECTRA);
        hipLaunchKernelGGL((k_conv_wide_prep<LOG2N>), dim3(2), dim3(F::T), lds, h->stream, taps_dev, a.tw,
                           (float2*)h->filt.p, gain_ptr, gain);
    }
    HIP_TRY(hipGetLastError());
    a.parts = 1;
    a.h_mid = (const float2*)h->filt.p;
    a.h_side = (const float2*)h->filt.p + F::N;
    a.npairs = (a.n + ConvWide<LOG2N>::HOP - 1) / ConvWide<LOG2N>::HOP;
    MGX_TRY(ensure(h, h->block_peak, (size_t)a.npairs * sizeof(float)));
    a.pair_peak = (float*)h->block_peak.p;
    a.queue = nullptr;
    int dev_cus = 256;
    HIP_TRY(hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, h->device));
    const int per_cu = std::max(1, std::min(2048 / F::T, (int)((size_t)160 * 1024 / lds)));
    const unsigned grid = (unsigned)std::min<long long>(a.npairs, (long long)dev_cus * per_cu);
    {
        StageScope scope(h, MGX_STAGE_CONVOLVE);
        hipLaunchKernelGGL((k_conv_wide<LOG2N>), dim3(grid), dim3(F::T), lds, h->stream, a);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

// taps_dev: [2][F] float (mid then side) already on the device.  F <= 8192: one overlap-save block of
// N = 2F per filter; longer filters (config #5: 16 k taps at 96 kHz) are cut into K = F/8192
// partitions on N = 16384 blocks (uniformly partitioned overlap-save), because N = 2F no longer fits
// a CU's LDS.  (Round 2 used K = F/4096 partitions on N = 8192 blocks: K + 1 transforms of 8192 points
// per channel and 8192 output frames, 325 flop per frame and channel; K/2 + 1 of 16384 points per 16384
// frames are 210.)
constexpr int LONG_FIR_LOG2N = 14;
static int run_conv(mgx_handle* h, const float* x, long long n, int taps, const float* taps_dev, double gain,
                    float* y, float* ymid, long long* npairs_out, const double* gain_ptr = nullptr) {
    const int l = ilog2_exact(taps);
    if (l < 0) return fail(MGX_ERR_ARGUMENT, "FIR length must be a power of two");
    MGX_TRY(check_length(n));
    if (taps <= CONV_DIRECT_MAX_TAPS) {                          // 2 .. 32 taps: in the time domain (small_fft_kernels.h)
        const long long tiles = (n + CONV_DIRECT_TILE - 1) / CONV_DIRECT_TILE;
        MGX_TRY(ensure(h, h->block_peak, (size_t)tiles * sizeof(float)));
        if (npairs_out) *npairs_out = tiles;
        StageScope scope(h, MGX_STAGE_CONVOLVE);
        hipLaunchKernelGGL(k_conv_direct, dim3((unsigned)tiles), dim3(256), 0, h->stream, reinterpret_cast<const float2*>(x),
                           (long long)n, taps_dev, taps, gain_ptr, gain, reinterpret_cast<float2*>(y), ymid,
                           (float*)h->block_peak.p);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    {
        // 4096 taps (the reference's default fft_size) on 16384-point blocks, three quarters of a block fresh output
        // (the variable: A/B against N = 2F)
        const char* no_wide = std::getenv("MGX_NO_CONV_WIDE");
        if (taps == 4096 && !(no_wide && no_wide[0] == '1')) {
            constexpr int WIDE = 14;
            MGX_TRY(ensure(h, h->filt, 2 * ((size_t)1 << WIDE) * sizeof(float2)));
            Conv2Args a;
            a.x = reinterpret_cast<const float2*>(x);
            a.n = n;
            a.y = reinterpret_cast<float2*>(y);
            a.ymid = ymid;
            a.run = 0;
            MGX_TRY(get_twiddles(h, WIDE, &a.tw));
            if (npairs_out) *npairs_out = (n + ConvWide<WIDE>::HOP - 1) / ConvWide<WIDE>::HOP;
            return launch_conv_wide<WIDE>(h, a, taps_dev, gain, gain_ptr);
        }
    }
    int log2b = l + 1;
    if (log2b > 14) log2b = LONG_FIR_LOG2N;
    // (N = 4F -- 4096 taps on 16384-point blocks, 187 instead of 260 flop per frame -- was retried in round 4 with the
    // register diet the kernel has had since round 1: 264 against 148 us, profiles/r04_g_conv_n4f.txt)
    const size_t nb = (size_t)1 << log2b;
    const int parts = (int)((size_t)2 * taps / nb);
    const long long pair_frames = (long long)nb;
    MGX_TRY(ensure(h, h->filt, 2 * (size_t)parts * nb * sizeof(float2)));
    Conv2Args a;
    a.x = reinterpret_cast<const float2*>(x);
    a.n = n;
    a.y = reinterpret_cast<float2*>(y);
    a.ymid = ymid;
    a.h_mid = (const float2*)h->filt.p;
    a.h_side = (const float2*)h->filt.p + (size_t)parts * nb;
    a.parts = parts;
    a.npairs = (n + pair_frames - 1) / pair_frames;
    a.pair_peak = nullptr;
    MGX_TRY(get_twiddles(h, log2b, &a.tw));
    a.run = 0;
    const char* no_delay = std::getenv("MGX_NO_CONV_DELAY");
    if (parts == 2 && !(no_delay && no_delay[0] == '1')) {      // (the variable: A/B against the partitioned kernel)
        if (npairs_out) *npairs_out = (n + (long long)nb / 2 - 1) / ((long long)nb / 2);
        return launch_conv_delay<LONG_FIR_LOG2N>(h, a, taps_dev, gain, gain_ptr);
    }
    if (npairs_out) *npairs_out = a.npairs;
    if (parts > 1) return launch_conv<LONG_FIR_LOG2N, true>(h, a, taps_dev, gain, gain_ptr);
    switch (log2b) {
#define CASE(L) case L: return launch_conv<L, false>(h, a, taps_dev, gain, gain_ptr);
        CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13) CASE(14)
#undef CASE
        default: return fail(MGX_ERR_UNSUPPORTED, "FIR length not supported by the convolution kernel");
    }
}

static int clipped_chunks(int divisions) { return std::max(1, std::min(1024, (2048 + divisions - 1) / divisions)); }

static int run_clipped_sumsq(mgx_handle* h, const float* mid, long long piece, int divisions,
                             const double* gain_ptr, double gain_mul, int* chunks_out) {
    const int chunks = clipped_chunks(divisions);
    MGX_TRY(ensure(h, h->partial, (size_t)divisions * chunks * sizeof(double)));
    hipLaunchKernelGGL(k_clipped_sumsq, dim3(divisions * chunks), dim3(256), 0, h->stream, mid, piece, chunks,
                       gain_ptr, gain_mul, (double*)h->partial.p);
    HIP_TRY(hipGetLastError());
    if (chunks_out) *chunks_out = chunks;
    return 0;
}

// look-back words and control block of a limiter launch over n frames (allocated, not initialised)
// blocks per limiter chunk: the rule of host_params.h, unless MGX_LIMIT_THREADS = 256 / 1024 asks otherwise (measurement
// aid; read where the parameters are derived so that an A/B inside one process sees it).  Chunks of 512 blocks were
// built and measured in round 6 for 96 kHz, where the halos eat a quarter of a 256-block chunk: 221 us against 204 (and
// 306 for 1024 blocks), 84 B of scratch at the 128 registers two workgroups per CU leave -- profiles/r06_b_*; removed.
static void limiter_threads_from_environment() {
    const char* e = std::getenv("MGX_LIMIT_THREADS");
    const int v = e ? std::atoi(e) : 0;
    limiter_threads_wish() = (v == 256 || v == 1024) ? v : 0;
}
static int limiter_state(mgx_handle* h, long long n, const mgx_config* cfg, unsigned long long** published,
                         long long* words, int** ticket) {
    LimiterParams lp;
    limiter_threads_from_environment();
    const std::string err = limiter_params(*cfg, lp);
    if (!err.empty()) return fail(MGX_ERR_UNSUPPORTED, err);
    const long long nchunks = (n + lp.geo.chunk - 1) / lp.geo.chunk;
    MGX_TRY(ensure(h, h->lim_published, (size_t)limiter_words(lp, nchunks) * sizeof(unsigned long long)));
    MGX_TRY(ensure_ctrl(h));
    *published = (unsigned long long*)h->lim_published.p;
    *words = limiter_words(lp, nchunks);
    *ticket = (int*)h->lim_ctrl.p;
    return 0;
}

// hold / release filters of order 2 or 3: k_limit_general<K>, its tables uploaded when the parameters change
template <int K>
static int launch_limiter_general(mgx_handle* h, const LimiterArgs& a, const LimiterParams& lp) {
    const std::vector<double> t = general_tables(lp);
    if (h->lim_tables_host != t) {
        MGX_TRY(ensure(h, h->lim_tables, t.size() * sizeof(double)));
        HIP_TRY(hipStreamSynchronize(h->stream));
        HIP_TRY(hipMemcpy(h->lim_tables.p, t.data(), t.size() * sizeof(double), hipMemcpyHostToDevice));
        h->lim_tables_host = t;
    }
    const GeneralArgs<K> g = general_fill<K>(lp, (const double*)h->lim_tables.p, a.published, a.nchunks);
    const size_t lds = LimiterGeneral<K>::LDS_BYTES;
    MGX_TRY((allow_lds(k_limit_general<K>, lds)));
    hipLaunchKernelGGL((k_limit_general<K>), dim3((unsigned)a.nchunks), dim3(256), lds, h->stream, a, g);
    HIP_TRY(hipGetLastError());
    return 0;
}

// 256-block chunks (four workgroups per CU) unless the configured attack / hold times need 1024
// (a persistent grid that fetches a workgroup's next chunk under its current one was built and measured in round 4:
// 187 against 171 us, profiles/r04_c_persistent_limiter.txt)
// the 256-block kernel: the instantiation for the configuration's window geometry when there is one (k_limit in
// mgx_kernels.h; MGX_LIMIT_GENERAL=1: measurement aid, always the general one)
static void launch_limiter_256(const LimiterArgs& a, dim3 grid, hipStream_t stream) {
    const size_t lds = LimiterBlock<256>::LDS_BYTES;
    const char* general = std::getenv("MGX_LIMIT_GENERAL");
    const bool fixed = !(general && general[0] == '1');
    if (fixed && a.hw == 44 && a.hb == 43 && a.gr == 26 && a.gl == 6 && a.gw == 3)                 // 44.1 kHz, 1 ms / 1 ms
        hipLaunchKernelGGL((k_limit<256, 4, 44, 43, 26>), grid, dim3(256), lds, stream, a);
    else if (fixed && a.hw == 48 && a.hb == 47 && a.gr == 28 && a.gl == 6 && a.gw == 3)            // 48 kHz
        hipLaunchKernelGGL((k_limit<256, 4, 48, 47, 28>), grid, dim3(256), lds, stream, a);
    else if (fixed && a.hw == 96 && a.hb == 95 && a.gr == 55 && a.gl == 12 && a.gw == 6)           // 96 kHz (BASELINE config #5)
        hipLaunchKernelGGL((k_limit<256, 4, 96, 95, 55>), grid, dim3(256), lds, stream, a);
    else
        hipLaunchKernelGGL((k_limit<256, 4>), grid, dim3(256), lds, stream, a);
}
static int launch_limiter(mgx_handle* h, const LimiterArgs& a, int threads) {
    const dim3 grid((unsigned)a.nchunks);
    if (threads == 1024) {
        const size_t lds = LimiterBlock<1024>::LDS_BYTES;
        MGX_TRY((allow_lds(k_limit<1024, 1>, lds)));
        hipLaunchKernelGGL((k_limit<1024, 1>), grid, dim3(1024), lds, h->stream, a);
    } else {
        launch_limiter_256(a, grid, h->stream);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

// everything of a limiter launch but the look-back words, ticket and error flag
static int limiter_args(mgx_handle* h, const float* y, long long n, const mgx_config* cfg, const double* gain_dev,
                        const double* post_dev, const int* active_dev, float* out, LimiterArgs& a, int* threads,
                        LimiterParams& lp) {
    limiter_threads_from_environment();
    const std::string err = limiter_params(*cfg, lp);
    if (!err.empty()) return fail(MGX_ERR_UNSUPPORTED, err);
    if (n < 8) return fail(MGX_ERR_ARGUMENT, "limiter input too short");
    limiter_fill(lp, (float)cfg->threshold, a);
    *threads = lp.threads;
    a.y = reinterpret_cast<const float2*>(y);
    a.n = n;
    a.out = reinterpret_cast<float2*>(out);
    a.gain = gain_dev;
    a.post_gain = post_dev;
    a.active = active_dev;
    a.nchunks = (n + lp.geo.chunk - 1) / lp.geo.chunk;
    // look-back weights: uploaded when the parameters change
    std::vector<double> w(lp.w_hold);
    w.insert(w.end(), lp.w_rel.begin(), lp.w_rel.end());
    w.insert(w.end(), lp.w_att.begin(), lp.w_att.end());
    const size_t wbytes = w.size() * sizeof(double);
    if (h->lim_weights_host != w) {
        MGX_TRY(ensure(h, h->lim_weights, wbytes));
        HIP_TRY(hipStreamSynchronize(h->stream));
        HIP_TRY(hipMemcpy(h->lim_weights.p, w.data(), wbytes, hipMemcpyHostToDevice));
        h->lim_weights_host = w;
    }
    a.w_hold = (const double*)h->lim_weights.p;
    a.w_rel = a.w_hold + lp.w_hold.size();
    a.w_att = a.w_rel + lp.w_rel.size();
    return 0;
}

// preset_done: the caller's previous kernel has already preset the look-back words and the ticket
static int run_limiter(mgx_handle* h, const float* y, long long n, const mgx_config* cfg, const double* gain_dev,
                       const double* post_dev, const int* active_dev, float* out, bool preset_done = false) {
    LimiterArgs a;
    LimiterParams lp;
    int threads = 256;
    MGX_TRY(limiter_args(h, y, n, cfg, gain_dev, post_dev, active_dev, out, a, &threads, lp));
    // published words preset to "unpublished", ticket and error zeroed, every launch
    const size_t pub_bytes = (size_t)limiter_words(lp, a.nchunks) * sizeof(unsigned long long);
    MGX_TRY(ensure(h, h->lim_published, pub_bytes));
    MGX_TRY(ensure_ctrl(h));
    a.published = (unsigned long long*)h->lim_published.p;
    const char* force_tickets = std::getenv("MGX_LIMIT_TICKETS");             // measurement aid / tests: "1" = always tickets
    a.ticket = (h->limiter_tickets || (force_tickets && force_tickets[0] == '1')) ? (int*)h->lim_ctrl.p : nullptr;
    h->limiter_ran_by_number = h->limiter_ran_by_number || a.ticket == nullptr;     // (since the last error check)
    a.error = h->error_dev;
    a.gave_up = (int*)h->lim_ctrl.p + 2;                      // ([0] the ticket, [2] "a waiter has given up")
    if (!preset_done) {
        HIP_TRY(hipMemsetAsync(h->lim_published.p, 0xff, pub_bytes, h->stream));
        HIP_TRY(hipMemsetAsync(h->lim_ctrl.p, 0, 16, h->stream));
    }
    // behind the device's previous limiter launch if another handle queued it (LimiterChain); a handle that is alone
    // on its device records nothing: its launches are ordered by its one stream
    LimiterChain& chain = g_limiter_chain[h->device % MAX_DEVICES];
    std::unique_lock<std::mutex> lock(chain.mu);
    const bool shared = chain.live > 1;
    if (shared && chain.last && chain.last != h->lim_done) HIP_TRY(hipStreamWaitEvent(h->stream, chain.last, 0));
    int rc = 0;
    switch (lp.general) {
        case 0: rc = launch_limiter(h, a, threads); break;
        case 2: rc = launch_limiter_general<2>(h, a, lp); break;
        default: rc = launch_limiter_general<3>(h, a, lp); break;
    }
    if (rc == 0 && shared) {
        HIP_TRY(hipEventRecord(h->lim_done, h->stream));
        chain.last = h->lim_done;
    }
    return rc;
}

// A bounded device-side wait expired (never seen in normal operation; the spins are bounded so that a lost word
// cannot hang the GPU, and the audio behind such a wait is wrong).  Called by every entry point that has just
// waited for the stream -- with or without a report -- so that the failure cannot pass silently; the flag is host
// memory, reading it costs nothing.  The counters a timed-out kernel may have left half-counted are put back, so
// the handle is good for the next call.
//   * DEVICE_ERROR_TAIL: k_correction_tail's workgroups (<= 129, spinning on each other's flag words) were not
//     resident together -- something else held the compute units.  Not a lost word: the last mgx_master call is
//     queued again with rounds 1..K-1 as one launch each (they wait for nobody), the handle stays that way, and the
//     call succeeds; mgx_last_error() carries a note.
//   * anything else (a limiter look-back word that never came): MGX_ERR_HIP.
static int queue_master(mgx_handle* h, const mgx_handle::MasterCall& c);
// `may_requeue`: the caller has not yet handed anything of the failed run to the host (master_impl before its report
// is read, mgx_synchronize / mgx_stage_times with no download queued behind the call, the blocking copy that re-issues
// itself).  Everywhere else -- and whenever MORE than one mgx_master call is outstanding since the last
// synchronisation, or a download of its outputs is already queued behind it (only the last call could be run again,
// and a copy would have taken the failed run's frames) -- an expired tail fails like any other expired wait; the handle
// still switches to one launch per round, so the caller's retry succeeds.
static int check_device_error(mgx_handle* h, bool may_requeue = false) {
    h->requeued = false;
    const int outstanding = h->masters_outstanding;
    const int copies = h->downloads_outstanding;
    const bool by_number = h->limiter_ran_by_number;
    h->masters_outstanding = 0;
    h->downloads_outstanding = 0;
    h->limiter_ran_by_number = false;
    if (!h->error_host) return 0;
    volatile int* e = (volatile int*)h->error_host;
    const int what = (e[DEVICE_ERROR_SLOT_LOOKBACK] ? DEVICE_ERROR_LOOKBACK : 0) | (e[DEVICE_ERROR_SLOT_TAIL] ? DEVICE_ERROR_TAIL : 0) |
                     (e[DEVICE_ERROR_SLOT_INPUT] ? DEVICE_ERROR_INPUT : 0);
    if (what == 0) return 0;
    for (int i = 0; i < DEVICE_ERROR_SLOTS; ++i) e[i] = 0;
    if (h->round_ctr.p) HIP_TRY(hipMemsetAsync(h->round_ctr.p, 0, h->round_ctr.bytes, h->stream));
    if (h->lim_ctrl.p) HIP_TRY(hipMemsetAsync(h->lim_ctrl.p, 0, 64, h->stream));
    if (h->conv_queue.p) HIP_TRY(hipMemsetAsync(h->conv_queue.p, 0, 64, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (what & DEVICE_ERROR_INPUT)
        return fail(MGX_ERR_ARGUMENT, "the target or the reference holds samples that are not finite numbers (NaN or infinity): "
                                      "the reference fails on such input too (match_frequencies.py:42)");
    // What a handle can recover from, once each: an expired tail (its workgroups were not resident together -> one launch
    // per round from now on) and an expired limiter look-back while chunks were dealt by workgroup number (-> tickets from
    // now on).  Anything else, or the same again in the safer mode, is a lost word.
    const bool tail_new = (what & DEVICE_ERROR_TAIL) && !h->avoid_tail;
    // (by the mode the failed launches REALLY ran in: under MGX_LIMIT_TICKETS=1 the handle's own switch may still be
    // off while every launch drew tickets -- an expired wait is then a lost word, not something to run again)
    const bool lookback_new = (what & DEVICE_ERROR_LOOKBACK) && !h->limiter_tickets && by_number;
    const bool recoverable = (!(what & DEVICE_ERROR_TAIL) || tail_new) && (!(what & DEVICE_ERROR_LOOKBACK) || lookback_new);
    if (recoverable) {
        if (tail_new) h->avoid_tail = true;                      // whatever happens next, this handle stops using the tail
        if (lookback_new) h->limiter_tickets = true;
        if (may_requeue && h->last_call.valid && outstanding == 1 && copies == 0) {
            const mgx_handle::MasterCall again = h->last_call;
            MGX_TRY(queue_master(h, again));
            HIP_TRY(hipStreamSynchronize(h->stream));
            h->masters_outstanding = 0;
            bool clean = true;
            for (int i = 0; i < DEVICE_ERROR_SLOTS; ++i) clean = clean && e[i] == 0;
            if (clean) {
                h->requeued = true;
                g_error = tail_new ? "note: the level-correction tail kernel's workgroups were not resident together (the GPU is shared); "
                                     "the call was run again with one launch per correction round, and this handle keeps doing so"
                                   : "note: a limiter look-back wait expired with chunks dealt by workgroup number; the call was run again "
                                     "with chunks drawn from an atomic ticket, and this handle keeps doing so";
                return 0;
            }
            for (int i = 0; i < DEVICE_ERROR_SLOTS; ++i) e[i] = 0;
        } else {
            return fail(MGX_ERR_RETRY, tail_new
                ? "the level-correction tail kernel's workgroups were not resident together (the GPU is shared) and "
                  "the results of the mgx_master calls since the last synchronisation are not valid; this handle now "
                  "runs one launch per correction round: call again"
                : "a limiter look-back wait expired with chunks dealt by workgroup number and the results of the calls since the "
                  "last synchronisation are not valid; this handle now draws chunks from an atomic ticket: call again");
        }
    }
    return fail(MGX_ERR_HIP, "a bounded device-side wait expired (limiter look-back or level-correction round): "
                             "the results of the calls since the last synchronisation are not valid");
}

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
extern "C" {

int mgx_version(void) { return 100; }
const char* mgx_last_error(void) { return g_error.c_str(); }

int mgx_device_count(int* count) {
    if (!count) return fail(MGX_ERR_ARGUMENT, "count is null");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return fail(MGX_ERR_NO_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorName(e));
    }
    *count = n;
    return 0;
}

int mgx_device_pci_bus_id(int device, char* out, int32_t capacity) {
    if (!out || capacity < 16) return fail(MGX_ERR_ARGUMENT, "need room for 16 characters");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(MGX_ERR_NO_DEVICE, "no HIP device (this library has no CPU fallback)");
    }
    if (device < 0 || device >= n) return fail(MGX_ERR_ARGUMENT, "no such device");
    HIP_TRY(hipDeviceGetPCIBusId(out, capacity, device));
    return 0;
}

int mgx_config_default(mgx_config* c) {
    if (!c) return fail(MGX_ERR_ARGUMENT, "config is null");
    std::memset(c, 0, sizeof(*c));
    c->internal_sample_rate = 44100;
    c->fft_size = 4096;
    c->lin_log_oversampling = 4;
    c->rms_correction_steps = 4;
    c->max_piece_size = 15.0 * 44100;
    c->threshold = (32768.0 - 61.0) / 32768.0;
    c->min_value = 1e-6;
    c->lowess_frac = 0.0375;
    c->lowess_it = 0;
    c->lowess_delta = 0.001;
    c->attack_ms = 1.0;
    c->hold_ms = 1.0;
    c->release_ms = 3000.0;
    c->attack_filter_coefficient = -2.0;
    c->hold_filter_order = 1;
    c->release_filter_order = 1;
    c->hold_filter_coefficient = 7.0;
    c->release_filter_coefficient = 800.0;
    return 0;
}

int mgx_create(int device, mgx_handle** out) {
    if (!out) return fail(MGX_ERR_ARGUMENT, "out is null");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(MGX_ERR_NO_DEVICE, "no HIP device visible: matchering_amd has no CPU fallback");
    if (device < 0 || device >= n) return fail(MGX_ERR_ARGUMENT, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    mgx_handle* h = new mgx_handle();
    h->device = device;
    HIP_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreate(&h->ev0));
    HIP_TRY(hipEventCreate(&h->ev1));
    HIP_TRY(hipEventCreateWithFlags(&h->lim_done, hipEventDisableTiming));
    {
        // the second handle of a device: whatever limiter the first has in flight was queued without an event
        LimiterChain& chain = g_limiter_chain[device % MAX_DEVICES];
        std::lock_guard<std::mutex> lock(chain.mu);
        if (++chain.live == 2) HIP_TRY(hipDeviceSynchronize());
    }
    HIP_TRY(hipHostMalloc((void**)&h->error_host, 64, hipHostMallocMapped));
    std::memset(h->error_host, 0, 64);
    HIP_TRY(hipHostGetDevicePointer((void**)&h->error_dev, h->error_host, 0));
    {   // code sizes for warm_code, once per device
        static std::mutex mu;
        static std::map<int, bool> done;
        std::lock_guard<std::mutex> lock(mu);
        if (!done[device]) {
            int bytes[CODE_KERNELS][CODE_VARIANTS];
            code_sizes_from_library(bytes);
            if (const char* off = std::getenv("MGX_NO_CODE_WARM"))          // measurement aid (tools/bench_stages.py variants)
                if (off[0] == '1')
                    for (auto& row : bytes)
                        for (int& b : row) b = 0;
            HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(mgx::g_code_bytes), bytes, sizeof(bytes)));
            done[device] = true;
        }
    }
    *out = h;
    return 0;
}

int mgx_destroy(mgx_handle* h) {
    if (!h) return 0;
    hipSetDevice(h->device);
    hipStreamSynchronize(h->stream);
    {
        LimiterChain& chain = g_limiter_chain[h->device % MAX_DEVICES];
        std::lock_guard<std::mutex> lock(chain.mu);
        if (chain.last == h->lim_done) chain.last = nullptr;       // (its limiter has finished: the stream has drained)
        --chain.live;
    }
    if (h->lim_done) hipEventDestroy(h->lim_done);
    if (h->comm) ncclCommDestroy(h->comm);
    DevBuf* bufs[] = {&h->y, &h->mid, &h->block_peak, &h->filt, &h->taps, &h->partial, &h->cstate,
                      &h->scalars, &h->lim_published, &h->lim_ctrl, &h->lim_weights, &h->lim_tables, &h->fir_robust, &h->peak_words, &h->fir_scratch, &h->round_ctr, &h->tail_gains, &h->band, &h->band_info,
                      &h->conv_queue};
    for (DevBuf* b : bufs)
        if (b->p) hipFree(b->p);
    for (TrackWork& w : h->track) {
        DevBuf* tb[] = {&w.wg_sumsq, &w.wg_peak, &w.wg_spec, &w.wg_pack, &w.stats, &w.rms, &w.loud, &w.avg, &w.part};
        for (DevBuf* b : tb)
            if (b->p) hipFree(b->p);
    }
    for (auto& kv : h->twiddles) hipFree(kv.second);
    if (h->pinned) hipHostFree(h->pinned);
    if (h->error_host) hipHostFree(h->error_host);
    hipEventDestroy(h->ev0);
    hipEventDestroy(h->ev1);
    for (auto& pair : h->stage_ev)
        for (hipEvent_t e : pair)
            if (e) hipEventDestroy(e);
    hipStreamDestroy(h->stream);
    delete h;
    return 0;
}

int mgx_malloc(mgx_handle* h, size_t bytes, void** dev) {
    if (!h || !dev) return fail(MGX_ERR_ARGUMENT, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMalloc(dev, bytes ? bytes : 256));
    return 0;
}
int mgx_free(mgx_handle* h, void* dev) {
    if (!h) return fail(MGX_ERR_ARGUMENT, "null handle");
    if (!dev) return 0;
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipFree(dev));
    h->last_call.valid = false;             // (its pointers may be the block that has just gone: never queued again)
    return 0;
}
int mgx_memcpy_h2d(mgx_handle* h, void* dev, const void* host, size_t bytes) {
    if (!h) return fail(MGX_ERR_ARGUMENT, "null handle");
    HIP_TRY(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return 0;
}
int mgx_memcpy_d2h(mgx_handle* h, void* host, const void* dev, size_t bytes) {
    if (!h) return fail(MGX_ERR_ARGUMENT, "null handle");
    HIP_TRY(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    MGX_TRY(check_device_error(h, true));
    if (h->requeued) {                       // the copy above took the failed run's bytes: take them again
        HIP_TRY(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    return 0;
}
// pinned host memory + copies that do not wait: the pieces of an overlapped host <-> HBM pipeline
int mgx_host_alloc(size_t bytes, void** host) {
    if (!host) return fail(MGX_ERR_ARGUMENT, "null argument");
    HIP_TRY(hipHostMalloc(host, bytes ? bytes : 256, hipHostMallocDefault));
    return 0;
}
int mgx_host_free(void* host) {
    if (!host) return 0;
    HIP_TRY(hipHostFree(host));
    return 0;
}
int mgx_memcpy_h2d_async(mgx_handle* h, void* dev, const void* host, size_t bytes) {
    if (!h) return fail(MGX_ERR_ARGUMENT, "null handle");
    HIP_TRY(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, h->stream));
    return 0;
}
int mgx_memcpy_d2h_async(mgx_handle* h, void* host, const void* dev, size_t bytes) {
    if (!h) return fail(MGX_ERR_ARGUMENT, "null handle");
    HIP_TRY(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, h->stream));
    if (h->masters_outstanding > 0) ++h->downloads_outstanding;
    return 0;
}
int mgx_synchronize(mgx_handle* h) {
    if (!h) return fail(MGX_ERR_ARGUMENT, "null handle");
    HIP_TRY(hipStreamSynchronize(h->stream));
    return check_device_error(h, true);
}
int mgx_timer_start(mgx_handle* h) {
    if (!h) return fail(MGX_ERR_ARGUMENT, "null handle");
    HIP_TRY(hipEventRecord(h->ev0, h->stream));
    return 0;
}
int mgx_timer_stop(mgx_handle* h, float* ms) {
    if (!h || !ms) return fail(MGX_ERR_ARGUMENT, "null argument");
    HIP_TRY(hipEventRecord(h->ev1, h->stream));
    HIP_TRY(hipEventSynchronize(h->ev1));
    HIP_TRY(hipEventElapsedTime(ms, h->ev0, h->ev1));
    return check_device_error(h);
}

// ---- stage level -----------------------------------------------------------
int mgx_analyze(mgx_handle* h, const float* x_dev, int64_t n, const mgx_config* cfg, int is_reference,
                double* peak, double* amplitude_coefficient, double* match_rms, int32_t* divisions,
                int64_t* piece_size, double* piece_rms, int32_t* loud, double* avg_mid, double* avg_side) {
    if (!h || !x_dev) return fail(MGX_ERR_ARGUMENT, "null argument");
    MGX_TRY(check_config(cfg));
    HIP_TRY(hipSetDevice(h->device));
    TrackWork& w = h->track[is_reference ? 1 : 0];
    w.is_reference = is_reference;
    MGX_TRY(run_analysis(h, cfg, x_dev, n, w));
    MGX_TRY(run_levels(h, cfg, &w, nullptr));
    {
        const int total = 2 * (cfg->fft_size / 2 + 1);
        hipLaunchKernelGGL(k_finish_spectra, dim3((total + 255) / 256), dim3(256), 0, h->stream,
                           (const double*)w.part.p, (const TrackStats*)w.stats.p, w.segs_per_piece, cfg->fft_size,
                           (double*)w.avg.p);
        HIP_TRY(hipGetLastError());
    }
    TrackStats st;
    HIP_TRY(hipMemcpyAsync(&st, w.stats.p, sizeof(st), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    const int half = cfg->fft_size / 2;
    if (peak) *peak = st.peak;
    if (amplitude_coefficient) *amplitude_coefficient = st.amplitude_c;
    if (match_rms) *match_rms = st.match_rms;
    if (divisions) *divisions = w.divisions;
    if (piece_size) *piece_size = w.piece;
    if (piece_rms) HIP_TRY(hipMemcpy(piece_rms, w.rms.p, w.divisions * sizeof(double), hipMemcpyDeviceToHost));
    if (loud) HIP_TRY(hipMemcpy(loud, w.loud.p, w.divisions * sizeof(int), hipMemcpyDeviceToHost));
    if (avg_mid) HIP_TRY(hipMemcpy(avg_mid, w.avg.p, (half + 1) * sizeof(double), hipMemcpyDeviceToHost));
    if (avg_side)
        HIP_TRY(hipMemcpy(avg_side, (double*)w.avg.p + (half + 1), (half + 1) * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

int mgx_design_fir(const mgx_config* cfg, const double* avg_target, const double* avg_reference, double* taps,
                   double* curve_raw, double* curve_smooth) {
    if (!cfg || !avg_target || !avg_reference || !taps) return fail(MGX_ERR_ARGUMENT, "null argument");
    if (ilog2_exact(cfg->fft_size) < 0 || cfg->fft_size < 8) return fail(MGX_ERR_ARGUMENT, "bad fft_size");
    if (cfg->lowess_it < 0 || cfg->lowess_it > 64) return fail(MGX_ERR_ARGUMENT, "lowess_it outside [0, 64]");
    FirDesignParams p{cfg->fft_size, cfg->internal_sample_rate, cfg->lin_log_oversampling, cfg->lowess_frac,
                      cfg->lowess_it, cfg->lowess_delta, cfg->min_value};
    design_fir(avg_target, avg_reference, p, taps, curve_raw, curve_smooth);
    return 0;
}

static int upload_taps(mgx_handle* h, const double* fir_mid, const double* fir_side, int taps) {
    MGX_TRY(ensure(h, h->taps, (size_t)2 * taps * sizeof(float)));
    MGX_TRY(ensure_pinned(h, std::max((size_t)2 * taps * sizeof(float), (size_t)1 << 16)));
    float* st = (float*)h->pinned;
    for (int i = 0; i < taps; ++i) {
        st[i] = (float)fir_mid[i];
        st[taps + i] = (float)fir_side[i];
    }
    HIP_TRY(hipMemcpyAsync(h->taps.p, st, (size_t)2 * taps * sizeof(float), hipMemcpyHostToDevice, h->stream));
    h->last_taps = taps;
    return 0;
}

int mgx_convolve(mgx_handle* h, const float* x_dev, int64_t n, const double* fir_mid, const double* fir_side,
                 int32_t taps, double gain, float* y_dev, float* y_mid_dev, double* peak) {
    if (!h || !x_dev || !fir_mid || !fir_side || !y_dev) return fail(MGX_ERR_ARGUMENT, "null argument");
    if (n <= 0) return fail(MGX_ERR_ARGUMENT, "empty input");
    HIP_TRY(hipSetDevice(h->device));
    MGX_TRY(upload_taps(h, fir_mid, fir_side, taps));
    long long nblocks = 0;
    MGX_TRY(run_conv(h, x_dev, n, taps, (const float*)h->taps.p, gain, y_dev, y_mid_dev, &nblocks));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (peak) {
        std::vector<float> bp(nblocks);
        HIP_TRY(hipMemcpy(bp.data(), h->block_peak.p, nblocks * sizeof(float), hipMemcpyDeviceToHost));
        float m = 0.f;
        for (float v : bp) m = std::max(m, v);
        *peak = m;
    }
    return 0;
}

int mgx_clipped_piece_sumsq(mgx_handle* h, const float* mid_dev, int64_t n, int64_t piece_size, int32_t divisions,
                            double gain, double* sumsq) {
    if (!h || !mid_dev || !sumsq) return fail(MGX_ERR_ARGUMENT, "null argument");
    if (piece_size <= 0 || divisions <= 0 || piece_size * divisions > n)
        return fail(MGX_ERR_ARGUMENT, "piece grid does not fit the array");
    HIP_TRY(hipSetDevice(h->device));
    int chunks = 0;
    MGX_TRY(run_clipped_sumsq(h, mid_dev, piece_size, divisions, nullptr, gain, &chunks));
    std::vector<double> part((size_t)divisions * chunks);
    HIP_TRY(hipMemcpyAsync(part.data(), h->partial.p, part.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (int d = 0; d < divisions; ++d) {
        double s = 0.0;
        for (int c = 0; c < chunks; ++c) s += part[(size_t)d * chunks + c];
        sumsq[d] = s;
    }
    return 0;
}

int mgx_limit(mgx_handle* h, const float* x_dev, int64_t n, const mgx_config* cfg, double gain, double post_gain,
              float* out_dev, int32_t* active) {
    if (!h || !x_dev || !out_dev) return fail(MGX_ERR_ARGUMENT, "null argument");
    MGX_TRY(check_config(cfg));
    HIP_TRY(hipSetDevice(h->device));
    // peak -> early-out decision, through the same kernels the pipeline uses
    MGX_TRY(ensure(h, h->cstate, sizeof(CorrectionState)));
    MGX_TRY(ensure(h, h->scalars, 64));
    MGX_TRY(ensure_pinned(h, 1 << 16));
    CorrectionState* cs = (CorrectionState*)h->cstate.p;
    hipLaunchKernelGGL(k_correction_init, dim3(1), dim3(1), 0, h->stream, cs, gain);
    // per-block peaks of x via the scale kernel's sibling: reuse k_finalize on a peak pass
    const long long nb = (n + 4095) / 4096;
    MGX_TRY(ensure(h, h->block_peak, (size_t)nb * sizeof(float)));
    hipLaunchKernelGGL(k_frame_peaks, dim3((unsigned)nb), dim3(256), 0, h->stream, (const float2*)x_dev, (long long)n,
                       (float*)h->block_peak.p);
    hipLaunchKernelGGL(k_finalize_scalars, dim3(1), dim3(256), 0, h->stream, (const float*)h->block_peak.p, nb,
                       cfg->threshold, cfg->min_value, cs);
    double* post = (double*)h->pinned;
    *post = post_gain;
    HIP_TRY(hipMemcpyAsync(h->scalars.p, post, sizeof(double), hipMemcpyHostToDevice, h->stream));
    CorrectionState host_cs;
    // Error words that earlier asynchronous mgx_master calls may have left are THEIRS: looked at (and reported) before
    // this call queues anything, so that the retry below can only ever answer for this call's own limiter.
    if (h->masters_outstanding > 0) {
        HIP_TRY(hipStreamSynchronize(h->stream));
        MGX_TRY(check_device_error(h));
    }
    for (int attempt = 0;; ++attempt) {
        const bool tickets_before = h->limiter_tickets;
        MGX_TRY(run_limiter(h, x_dev, n, cfg, &cs->gain, (const double*)h->scalars.p, &cs->limiter_active, out_dev));
        HIP_TRY(hipMemcpyAsync(&host_cs, cs, sizeof(host_cs), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        const int rc = check_device_error(h);
        if (rc == 0) break;
        if (attempt > 0 || tickets_before || !h->limiter_tickets) return rc;       // (once more when the handle has just switched to tickets)
    }
    if (active) *active = host_cs.limiter_active;
    return 0;
}

int mgx_scale(mgx_handle* h, const float* x_dev, int64_t n, double gain, float* out_dev) {
    if (!h || !x_dev || !out_dev) return fail(MGX_ERR_ARGUMENT, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    const unsigned grid = (unsigned)std::min<long long>((n + 255) / 256, 8192);
    hipLaunchKernelGGL(k_scale_outputs, dim3(grid), dim3(256), 0, h->stream, (const float2*)x_dev, (long long)n,
                       (const double*)nullptr, gain, (const double*)nullptr, (float2*)out_dev, (float2*)nullptr);
    HIP_TRY(hipGetLastError());
    return 0;
}

int mgx_peak_count(mgx_handle* h, const float* x_dev, int64_t samples, double* peak, int64_t* count) {
    if (!h || !x_dev || !peak || !count) return fail(MGX_ERR_ARGUMENT, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    MGX_TRY(ensure(h, h->peak_words, 64));
    MGX_TRY(ensure_pinned(h, (size_t)1 << 16));
    unsigned long long* words = (unsigned long long*)h->peak_words.p;
    HIP_TRY(hipMemsetAsync(words, 0, 16, h->stream));
    if (samples > 0) {
        const unsigned grid = (unsigned)std::min<long long>((samples / 4 + 255) / 256 + 1, 4096);
        hipLaunchKernelGGL(k_peak_max, dim3(grid), dim3(256), 0, h->stream, x_dev, (long long)samples, words);
        hipLaunchKernelGGL(k_peak_count, dim3(grid), dim3(256), 0, h->stream, x_dev, (long long)samples, words);
        HIP_TRY(hipGetLastError());
    }
    unsigned long long* host = (unsigned long long*)h->pinned;
    HIP_TRY(hipMemcpyAsync(host, words, 16, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    const unsigned bits = (unsigned)host[0];
    float f;
    std::memcpy(&f, &bits, 4);
    *peak = (double)f;
    *count = (int64_t)host[1];
    return 0;
}

int mgx_window_energy(mgx_handle* h, const float* x_dev, int64_t n, int64_t size, int64_t step, double* energy,
                      int64_t capacity, int64_t* count) {
    if (!h || !x_dev || !energy || !count || n < 1 || size < 1 || step < 1)
        return fail(MGX_ERR_ARGUMENT, "bad window energy arguments");
    HIP_TRY(hipSetDevice(h->device));
    if (size > n) size = n;                                   // dsp.py:131-132: the whole array is the only window
    const int64_t windows = (n - size) / step + 1;
    *count = windows;
    if (windows > capacity) return fail(MGX_ERR_ARGUMENT, "energy array too small for the number of windows");
    const int chunks = (int)std::max<int64_t>(1, std::min<int64_t>(64, size / 16384));
    const size_t bytes = (size_t)windows * chunks * sizeof(double);
    MGX_TRY(ensure(h, h->partial, bytes));
    MGX_TRY(ensure_pinned(h, std::max(bytes, (size_t)1 << 16)));
    for (int64_t w0 = 0; w0 < windows; w0 += 65535) {          // (a grid's y extent is 16 bits; the reference has no limit)
        const unsigned rows = (unsigned)std::min<int64_t>(65535, windows - w0);
        hipLaunchKernelGGL(k_window_energy, dim3(chunks, rows), dim3(256), 0, h->stream, (const float2*)x_dev,
                           (long long)size, (long long)step, chunks, (double*)h->partial.p, (long long)w0);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(h->pinned, h->partial.p, bytes, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    MGX_TRY(check_device_error(h));
    const double* part = (const double*)h->pinned;
    for (int64_t w = 0; w < windows; ++w) {
        double s = 0.0;
        for (int c = 0; c < chunks; ++c) s += part[(size_t)w * chunks + c];
        energy[w] = s;
    }
    return 0;
}

int mgx_preview_cut(mgx_handle* h, const float* x_dev, int64_t n, int64_t begin, int64_t size, int64_t fade,
                    double clip_limit, float* out_dev) {
    if (!h || !x_dev || !out_dev || begin < 0 || size < 1 || begin + size > n || fade < 0 || fade > size)
        return fail(MGX_ERR_ARGUMENT, "bad preview cut arguments");
    HIP_TRY(hipSetDevice(h->device));
    const unsigned grid = (unsigned)std::min<int64_t>((size + 255) / 256, 4096);
    hipLaunchKernelGGL(k_preview_cut, dim3(grid), dim3(256), 0, h->stream, (const float2*)x_dev, (long long)begin,
                       (long long)size, (long long)fade, clip_limit, (float2*)out_dev);
    HIP_TRY(hipGetLastError());
    return 0;
}

int mgx_pcm_decode(mgx_handle* h, const void* pcm_dev, int64_t samples, int32_t bits, float* out_dev) {
    if (!h || !pcm_dev || !out_dev) return fail(MGX_ERR_ARGUMENT, "null argument");
    if (bits != 16 && bits != 24 && bits != 32) return fail(MGX_ERR_ARGUMENT, "PCM width must be 16, 24 or 32 bits");
    if (samples <= 0) return 0;
    HIP_TRY(hipSetDevice(h->device));
    const unsigned grid = (unsigned)std::min<long long>((samples / 4 + 255) / 256 + 1, 8192);
    hipLaunchKernelGGL(k_pcm_decode, dim3(grid), dim3(256), 0, h->stream, pcm_dev, (long long)samples, bits, out_dev);
    HIP_TRY(hipGetLastError());
    return 0;
}

int mgx_pcm_encode(mgx_handle* h, const float* x_dev, int64_t samples, int32_t bits, void* pcm_dev) {
    if (!h || !pcm_dev || !x_dev) return fail(MGX_ERR_ARGUMENT, "null argument");
    if (bits != 16 && bits != 24 && bits != 32) return fail(MGX_ERR_ARGUMENT, "PCM width must be 16, 24 or 32 bits");
    if (samples <= 0) return 0;
    HIP_TRY(hipSetDevice(h->device));
    const unsigned grid = (unsigned)std::min<long long>((samples / 4 + 255) / 256 + 1, 8192);
    hipLaunchKernelGGL(k_pcm_encode, dim3(grid), dim3(256), 0, h->stream, x_dev, (long long)samples, bits, pcm_dev);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---- the boundary: stages.main ----------------------------------------------
} // extern "C"
static int master_impl(mgx_handle* h, const float* target_dev, int64_t n_target, const float* reference_dev,
                       int64_t n_reference, const mgx_config* cfg, const float* fir_given, float* result_dev,
                       float* result_no_limiter_dev, float* result_no_limiter_normalized_dev, mgx_report* report);
// every launch of one stages.main, queued on the handle's stream (no host round trip)
static int dev_repeat_default(const char* name, int fallback) {
    const char* e = std::getenv(name);
    return e ? std::max(1, std::atoi(e)) : fallback;
}
static int dev_repeat(const char* name) { return dev_repeat_default(name, 1); }
static int queue_master(mgx_handle* h, const mgx_handle::MasterCall& c) {
    const float* target_dev = c.target;
    const float* reference_dev = c.reference;
    const float* fir_given = c.fir_given;
    const int64_t n_target = c.n_target, n_reference = c.n_reference;
    const mgx_config* cfg = &c.cfg;
    float* result_dev = c.out[0];
    float* result_no_limiter_dev = c.out[1];
    float* result_no_limiter_normalized_dev = c.out[2];
    const int f = cfg->fft_size;
    if (result_dev) {               // validate limiter parameters before any work is queued
        LimiterParams lp;
        const std::string err = limiter_params(*cfg, lp);
        if (!err.empty()) return fail(MGX_ERR_UNSUPPORTED, err);
    }
    // stage 1 (stages.py:38-104): both tracks analysed in one pass each
    TrackWork& tw = h->track[0];
    TrackWork& rw = h->track[1];
    // (analysing the reference first, so that the target is the fresher track in the Infinity Cache when
    // the convolution reads it, was measured: no difference)
    {
        StageScope scope(h, MGX_STAGE_ANALYZE);
        tw.is_reference = 0;
        rw.is_reference = 1;
        MGX_TRY(run_analysis(h, cfg, target_dev, n_target, tw, reference_dev, n_reference, &rw));
    }
    // stage 2 (stages.py:107-135): FIR design on the device, then the overlap-save convolution with
    // the level gain of stages.py:80-88 (a device scalar) folded into the filter spectra
    {
        StageScope scope(h, MGX_STAGE_DESIGN_FIR);
        MGX_TRY(run_fir_design(h, cfg, tw, rw, fir_given));
    }
    MGX_TRY(ensure(h, h->y, (size_t)n_target * sizeof(float2)));
    MGX_TRY(ensure(h, h->mid, (size_t)n_target * sizeof(float)));
    long long nblocks = 0;
    // (MGX_DEV_REPEAT_CONV / MGX_DEV_REPEAT_LIMIT = n: measurement aid, the stage's launches n times in a row -- the
    // difference between n = 2 and n = 1 is the stage with its code already in the instruction caches)
    for (int rep = dev_repeat("MGX_DEV_REPEAT_CONV"); rep > 0; --rep)
        MGX_TRY(run_conv(h, target_dev, n_target, f, (const float*)h->taps.p, 1.0, (float*)h->y.p, (float*)h->mid.p,
                         &nblocks, (const double*)h->scalars.p));
    // stage 3 (stages.py:138-170): scalar feedback stays on the device; one launch per round, the last
    // round also derives the peak / early-out / normalisation scalars (the state was reset by k_fir_raw)
    CorrectionState* cs = (CorrectionState*)h->cstate.p;
    bool limiter_preset = false;
    {
        StageScope scope(h, MGX_STAGE_CORRECT_LEVELS);
        RoundArgs ra;
        ra.mid = (const float*)h->mid.p;
        ra.piece = tw.piece;
        ra.divisions = tw.divisions;
        ra.chunks = std::max(1, dev_repeat_default("MGX_ROUND_WGS", 1024) / tw.divisions);     // ~1000 workgroups: each pays one publish + ticket
        // (round 0's partial sums, and behind them the peak words of k_correction_tail's workgroups)
        MGX_TRY(ensure(h, h->partial, (size_t)2 * ra.divisions * ra.chunks * sizeof(double)));
        ra.partial = (double*)h->partial.p;
        // arrival counters [1 + divisions], zero between launches; the 16 gain words of k_correction_tail
        // live in their own buffer (a layout that moved with `divisions` would leave one call's preset
        // gain words where the next call counts arrivals)
        const size_t ctr_bytes = (size_t)(1 + ra.divisions) * sizeof(unsigned);
        // at most ~128 workgroups in k_correction_tail, at most 64 chunks (the lanes of a wave) per workgroup
        const int tail_groups = std::max((ra.chunks + 63) / 64, std::max(1, std::min(ra.chunks, 128 / ra.divisions)));
        // (MGX_NO_TAIL=1: measurement aid, one launch per correction round for this call)
        const char* no_tail = std::getenv("MGX_NO_TAIL");
        const bool use_tail = cfg->rms_correction_steps > 1 && !h->avoid_tail && !(no_tail && no_tail[0] == '1');
        const int tail_total = use_tail ? ra.divisions * tail_groups : 0;
        const int tail_rounds = use_tail ? cfg->rms_correction_steps - 1 : 0;
        MGX_TRY(ensure(h, h->tail_gains, ((size_t)tail_rounds + 1 + (size_t)tail_rounds * tail_total) * sizeof(unsigned long long)));
        ra.tail_total = tail_total;
        ra.tail_rounds = tail_rounds;
        if (h->round_ctr.bytes < ctr_bytes) {                 // zeroed when (re)allocated, reset by each launch
            MGX_TRY(ensure(h, h->round_ctr, std::max(ctr_bytes, (size_t)4096)));
            HIP_TRY(hipMemsetAsync(h->round_ctr.p, 0, h->round_ctr.bytes, h->stream));
        }
        ra.arrivals = (unsigned*)h->round_ctr.p;
        const size_t wgs = (size_t)ra.divisions * ra.chunks;
        MGX_TRY(ensure(h, h->band, ((size_t)n_target + wgs * BAND_SLACK) * sizeof(float)));
        MGX_TRY(ensure(h, h->band_info, wgs * sizeof(BandInfo)));
        ra.band = (float*)h->band.p;
        ra.info = (BandInfo*)h->band_info.p;
        ra.reference_match_rms = &((const TrackStats*)rw.stats.p)->match_rms;
        ra.eps = cfg->min_value;
        ra.threshold = cfg->threshold;
        ra.cs = cs;
        ra.npeaks = nblocks;
        MGX_TRY(ensure_ctrl(h));
        ra.error = h->error_dev;
        const size_t lds_step = (size_t)(64 + tw.divisions + (size_t)ra.divisions * ra.chunks) * sizeof(double);
        if (lds_step > (size_t)150 * 1024)
            return fail(MGX_ERR_UNSUPPORTED, "too many analysis pieces for the level-correction kernel's LDS");
        MGX_TRY(allow_lds(k_correction_round, lds_step));
        const int rounds = cfg->rms_correction_steps;
        ra.lim_published = nullptr;
        ra.lim_words = 0;
        ra.lim_ticket = nullptr;
        ra.tail_gains = use_tail ? (unsigned long long*)h->tail_gains.p : nullptr;
        auto with_final = [&](RoundArgs& r) -> int {          // the launch that runs the last round
            r.final_peaks = (const float*)h->block_peak.p;
            return 0;
        };
        if (rounds >= 1) {                                    // round 0 streams the mid plane and builds the band lists
            RoundArgs r0 = ra;
            r0.final_peaks = nullptr;
            r0.build_band = 1;
            r0.step = 0;
            if (result_dev) {     // the limiter's look-back words are preset by this grid: a thousand workgroups, two words a thread
                MGX_TRY(limiter_state(h, n_target, cfg, &r0.lim_published, &r0.lim_words, &r0.lim_ticket));
                limiter_preset = true;
            }
            if (rounds == 1) MGX_TRY(with_final(r0));
            hipLaunchKernelGGL(k_correction_round, dim3(ra.divisions * ra.chunks), dim3(256), lds_step, h->stream, r0);
        }
        if (rounds > 1 && !use_tail) {
            // one launch per round, the last arriver of each decides (k_correction_round without a tail): what a handle
            // falls back to after its tail kernel found the GPU shared (check_device_error), and MGX_NO_TAIL=1
            for (int r = 1; r < rounds; ++r) {
                RoundArgs rr = ra;
                rr.build_band = 0;
                rr.step = r;
                rr.final_peaks = nullptr;
                if (r == rounds - 1) MGX_TRY(with_final(rr));
                hipLaunchKernelGGL(k_correction_round, dim3(ra.divisions * ra.chunks), dim3(256), lds_step, h->stream, rr);
            }
        }
        if (use_tail) {                                       // every further round inside one small resident grid
            RoundArgs rt = ra;
            rt.build_band = 0;
            rt.step = 1;
            MGX_TRY(with_final(rt));
            const int groups = tail_groups;
            const size_t lds_tail = correction_tail_lds_bytes(ra.divisions, groups, ra.chunks);
            if (lds_tail > (size_t)150 * 1024)
                return fail(MGX_ERR_UNSUPPORTED, "too many analysis pieces for the level-correction kernel's LDS");
            MGX_TRY(allow_lds(k_correction_tail, lds_tail));
            // (+ 1: the deciding workgroup)
            hipLaunchKernelGGL(k_correction_tail, dim3(ra.divisions * groups + 1), dim3(256), lds_tail, h->stream, rt, groups,
                               rounds - 1);
        }
        if (rounds == 0)
            hipLaunchKernelGGL(k_finalize_scalars, dim3(1), dim3(256), 0, h->stream, (const float*)h->block_peak.p,
                               nblocks, cfg->threshold, cfg->min_value, cs);
        HIP_TRY(hipGetLastError());
    }
    // stage 4 (stages.py:173-207)
    if (result_no_limiter_dev || result_no_limiter_normalized_dev) {
        StageScope scope(h, MGX_STAGE_SCALE_OUTPUTS);
        const unsigned grid = (unsigned)std::min<long long>((n_target + 255) / 256, 8192);
        hipLaunchKernelGGL(k_scale_outputs, dim3(grid), dim3(256), 0, h->stream, (const float2*)h->y.p,
                           (long long)n_target, (const double*)&cs->gain, 1.0, (const double*)&cs->normalize_c,
                           (float2*)result_no_limiter_dev, (float2*)result_no_limiter_normalized_dev);
        HIP_TRY(hipGetLastError());
    }
    if (result_dev) {
        StageScope scope(h, MGX_STAGE_LIMIT);
        const double* post = &((const TrackStats*)rw.stats.p)->amplitude_c;
        MGX_TRY(run_limiter(h, (const float*)h->y.p, n_target, cfg, &cs->gain, post, &cs->limiter_active, result_dev,
                            limiter_preset));
        for (int rep = dev_repeat("MGX_DEV_REPEAT_LIMIT"); rep > 1; --rep)
            MGX_TRY(run_limiter(h, (const float*)h->y.p, n_target, cfg, &cs->gain, post, &cs->limiter_active, result_dev, false));
    }
    return 0;
}

static int master_impl(mgx_handle* h, const float* target_dev, int64_t n_target, const float* reference_dev,
                       int64_t n_reference, const mgx_config* cfg, const float* fir_given, float* result_dev,
                       float* result_no_limiter_dev, float* result_no_limiter_normalized_dev, mgx_report* report) {
    if (!h || !target_dev || !reference_dev) return fail(MGX_ERR_ARGUMENT, "null argument");
    MGX_TRY(check_config(cfg));
    HIP_TRY(hipSetDevice(h->device));
    if (result_dev) {               // validate limiter parameters before any work is queued
        LimiterParams lp;
        const std::string err = limiter_params(*cfg, lp);
        if (!err.empty()) return fail(MGX_ERR_UNSUPPORTED, err);
    }
    mgx_handle::MasterCall& call = h->last_call;
    call.valid = false;
    call.target = target_dev;
    call.reference = reference_dev;
    call.fir_given = fir_given;
    call.n_target = n_target;
    call.n_reference = n_reference;
    call.cfg = *cfg;
    call.out[0] = result_dev;
    call.out[1] = result_no_limiter_dev;
    call.out[2] = result_no_limiter_normalized_dev;
    MGX_TRY(queue_master(h, call));
    call.valid = true;
    ++h->masters_outstanding;
    TrackWork& tw = h->track[0];
    TrackWork& rw = h->track[1];
    CorrectionState* cs = (CorrectionState*)h->cstate.p;
    if (report) {
        MGX_TRY(ensure_pinned(h, 1 << 16));
        char* pin = (char*)h->pinned;
        TrackStats* st_t = (TrackStats*)pin;
        TrackStats* st_r = (TrackStats*)(pin + 256);
        CorrectionState* hc = (CorrectionState*)(pin + 512);
        double* c0 = (double*)(pin + 1024);
        for (int attempt = 0; attempt < 2; ++attempt) {      // (a second time when the check queued the call again)
            HIP_TRY(hipMemcpyAsync(st_t, tw.stats.p, sizeof(TrackStats), hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(hipMemcpyAsync(st_r, rw.stats.p, sizeof(TrackStats), hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(hipMemcpyAsync(hc, cs, sizeof(CorrectionState), hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(hipMemcpyAsync(c0, h->scalars.p, sizeof(double), hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(hipStreamSynchronize(h->stream));
            MGX_TRY(check_device_error(h, true));                 // (nothing of the run has been handed out yet)
            if (!h->requeued) break;
        }
        std::memset(report, 0, sizeof(*report));
        report->final_amplitude_coefficient = st_r->amplitude_c;
        report->target_match_rms = st_t->match_rms;
        report->reference_match_rms = st_r->match_rms;
        report->rms_coefficient = *c0;
        for (int i = 0; i < 16; ++i) report->correction_coefficients[i] = hc->coeffs[i];
        report->normalize_coefficient = result_no_limiter_normalized_dev ? hc->normalize_c : 0.0;
        report->result_peak = hc->result_peak;
        report->target_divisions = st_t->divisions;
        report->reference_divisions = st_r->divisions;
        report->target_piece = st_t->piece;
        report->reference_piece = st_r->piece;
        report->target_loud_count = st_t->loud_count;
        report->reference_loud_count = st_r->loud_count;
        report->limiter_active = hc->limiter_active;
    }
    return 0;
}

extern "C" {
int mgx_master(mgx_handle* h, const float* target_dev, int64_t n_target, const float* reference_dev,
               int64_t n_reference, const mgx_config* cfg, float* result_dev, float* result_no_limiter_dev,
               float* result_no_limiter_normalized_dev, mgx_report* report) {
    return master_impl(h, target_dev, n_target, reference_dev, n_reference, cfg, nullptr, result_dev,
                       result_no_limiter_dev, result_no_limiter_normalized_dev, report);
}
int mgx_master_with_fir(mgx_handle* h, const float* target_dev, int64_t n_target, const float* reference_dev,
                        int64_t n_reference, const mgx_config* cfg, const float* fir_dev, float* result_dev,
                        float* result_no_limiter_dev, float* result_no_limiter_normalized_dev, mgx_report* report) {
    if (!fir_dev) return fail(MGX_ERR_ARGUMENT, "null FIR");
    return master_impl(h, target_dev, n_target, reference_dev, n_reference, cfg, fir_dev, result_dev,
                       result_no_limiter_dev, result_no_limiter_normalized_dev, report);
}

int mgx_stage_timing(mgx_handle* h, int32_t enable) {
    if (!h) return fail(MGX_ERR_ARGUMENT, "null handle");
    h->stage_timing = enable != 0;
    for (bool& u : h->stage_used) u = false;
    return 0;
}
#ifdef MGX_DEV_CONV_PHASES           // development builds only (tools/conv_delay_phases.py)
int mgx_dev_conv_ticks_read(unsigned* out, int blocks) {        // out: [blocks][8]
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(mgx::mgx_dev_conv_ticks), (size_t)blocks * 8 * sizeof(unsigned)) != hipSuccess;
}
#endif
#ifdef MGX_DEV_LIMITER_PHASES        // development builds only (tools/limiter_phases.py)
int mgx_dev_chunk_life_read(long long* out, int chunks) {       // out: [chunks][8]
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(mgx::mgx_dev_chunk_life), (size_t)chunks * 8 * sizeof(long long)) != hipSuccess;
}
int mgx_dev_phase_ticks_read(unsigned* out, int chunks) {       // out: [chunks][16]
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(mgx::mgx_dev_phase_ticks), (size_t)chunks * 16 * sizeof(unsigned)) != hipSuccess;
}
#endif
int mgx_stage_times(mgx_handle* h, float* ms) {
    if (!h || !ms) return fail(MGX_ERR_ARGUMENT, "null argument");
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (int s = 0; s < MGX_STAGE_COUNT; ++s) {
        ms[s] = -1.f;
        if (!h->stage_used[s]) continue;
        HIP_TRY(hipEventElapsedTime(&ms[s], h->stage_ev[s][0], h->stage_ev[s][1]));
        h->stage_used[s] = false;
    }
    return check_device_error(h);
}

}  // extern "C"
extern "C" {
int mgx_code_bytes(int32_t* bytes, int32_t capacity) {
    if (!bytes || capacity < CODE_KERNELS * CODE_VARIANTS) return fail(MGX_ERR_ARGUMENT, "need room for 7 x 16 sizes");
    int found[CODE_KERNELS][CODE_VARIANTS];
    code_sizes_from_library(found);
    for (int c = 0; c < CODE_KERNELS; ++c)
        for (int v = 0; v < CODE_VARIANTS; ++v) bytes[c * CODE_VARIANTS + v] = found[c][v];
    return CODE_KERNELS;
}

int mgx_last_fir(mgx_handle* h, void** taps_dev, int32_t* taps) {
    if (!h || !taps_dev || !taps) return fail(MGX_ERR_ARGUMENT, "null argument");
    if (!h->taps.p || h->last_taps <= 0) return fail(MGX_ERR_ARGUMENT, "no FIR has been designed on this handle yet");
    *taps_dev = h->taps.p;
    *taps = h->last_taps;
    return 0;
}

// ---- RCCL ----------------------------------------------------------------------
// The ranks of a job are the GPUs of ONE node, the data path between them is xGMI and the bootstrap needs nothing but
// loop-back -- but telling RCCL so (NCCL_SOCKET_IFNAME=lo, NCCL_IB_DISABLE=1) changes the environment of the HOST
// process, which is the host's decision and not thread-safe against the getenv calls of a running library: the Python
// front end does it where a job sets itself up (matchering_amd/ranks.py single_node_rccl_defaults), a C / C++ host sets
// the two variables itself before its first mgx_comm_* call.  Nothing here calls setenv.
int mgx_comm_unique_id(void* id128) {
    if (!id128) return fail(MGX_ERR_ARGUMENT, "null argument");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    NCCL_TRY(ncclGetUniqueId(&id));
    std::memcpy(id128, &id, sizeof(id));
    return 0;
}
int mgx_comm_init(mgx_handle* h, const void* id128, int rank, int world) {
    if (!h || !id128) return fail(MGX_ERR_ARGUMENT, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    NCCL_TRY(ncclCommInitRank(&h->comm, world, id, rank));
    h->comm_rank = rank;
    h->comm_world = world;
    return 0;
}
int mgx_comm_count(mgx_handle* h, int32_t* ranks) {
    if (!h || !ranks) return fail(MGX_ERR_ARGUMENT, "null argument");
    if (!h->comm) return fail(MGX_ERR_ARGUMENT, "communicator not initialised");
    int n = 0;
    NCCL_TRY(ncclCommCount(h->comm, &n));
    *ranks = n;
    return 0;
}
int mgx_comm_broadcast_f32(mgx_handle* h, float* dev, int64_t count, int root) {
    if (!h || !h->comm) return fail(MGX_ERR_ARGUMENT, "communicator not initialised");
    NCCL_TRY(ncclBroadcast(dev, dev, (size_t)count, ncclFloat, root, h->comm, h->stream));
    return 0;
}
int mgx_comm_allgather_f32(mgx_handle* h, const float* send_dev, float* recv_dev, int64_t count) {
    if (!h || !h->comm) return fail(MGX_ERR_ARGUMENT, "communicator not initialised");
    NCCL_TRY(ncclAllGather(send_dev, recv_dev, (size_t)count, ncclFloat, h->comm, h->stream));
    return 0;
}
int mgx_comm_destroy(mgx_handle* h) {
    if (!h || !h->comm) return 0;
    NCCL_TRY(ncclCommDestroy(h->comm));
    h->comm = nullptr;
    return 0;
}

}  // extern "C"


#ifdef MGX_TAIL_TRACE
// experiments only (tools/tail_trace.py): the 100 MHz phase stamps of the last k_correction_round / k_correction_tail
extern "C" int mgx_debug_tail_trace(unsigned long long* tail /* [160*32] */, unsigned long long* round /* [8] */) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(tail, HIP_SYMBOL(mgx::g_tail_trace), sizeof(unsigned long long) * 160 * 32));
    HIP_TRY(hipMemcpyFromSymbol(round, HIP_SYMBOL(mgx::g_round_trace), sizeof(unsigned long long) * 8));
    return 0;
}
#endif
