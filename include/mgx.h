/* mgx -- C ABI of the MI355X-native mastering core.
 *
 * sergree/matchering is pure Python: it has no FFI of its own.  The functions
 * below are the boundary a maintainer would bind (ctypes stub in INTEGRATION.md)
 * to replace the body of matchering/stages.py:210-272 (`main`) and, one level
 * down, the stage helpers it calls.  Each entry point cites the reference
 * interface it replaces.  Plain pointers and sizes only; no framework types.
 *
 * Conventions
 *  - audio is float32, interleaved stereo frames (n,2) -- numpy C order, the
 *    layout soundfile hands to matchering/loader.py:35.
 *  - "dev" pointers are device (HBM) addresses obtained from mgx_malloc; "host"
 *    pointers are ordinary memory.  Nothing is freed or retained across calls
 *    except through the handle.
 *  - every function returns 0 on success, a negative mgx_status otherwise;
 *    mgx_last_error() gives the message (thread-local).
 *  - a handle is bound to one GPU and one HIP stream; calls on one handle are
 *    serialised by the caller, different handles are independent.
 */
#ifndef MGX_H
#define MGX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mgx_handle mgx_handle;

enum mgx_status {
    MGX_OK = 0,
    MGX_ERR_ARGUMENT = -1,     /* bad size / null pointer / unsupported parameter */
    MGX_ERR_HIP = -2,          /* HIP runtime error (message has the hipError name) */
    MGX_ERR_NO_DEVICE = -3,    /* no usable GPU: there is NO CPU fallback */
    MGX_ERR_UNSUPPORTED = -4,  /* valid for the reference but not implemented here */
    MGX_ERR_RCCL = -5,
    MGX_ERR_RETRY = -6         /* a bounded device-side wait expired (the GPU is shared with somebody else's kernels); the
                                  outputs of the calls since the last synchronisation are not valid, the handle has
                                  switched to the mode that cannot wait for ever, and the same call succeeds when made
                                  again.  Blocking calls retry by themselves where nothing of the failed run has
                                  reached the host; this code is for the asynchronous ones. */
};

/* matchering/defaults.py:25-58 LimiterConfig + :61-155 Config, the fields the
 * hot path consumes.  max_piece_size is in SAMPLES (defaults.py:109 already
 * multiplied it by the sample rate). */
typedef struct mgx_config {
    int32_t internal_sample_rate;
    int32_t fft_size;
    int32_t lin_log_oversampling;
    int32_t rms_correction_steps;
    double max_piece_size;
    double threshold;
    double min_value;
    double lowess_frac;
    int32_t lowess_it;
    int32_t reserved0;
    double lowess_delta;
    /* limiter */
    double attack_ms, hold_ms, release_ms;
    double attack_filter_coefficient;
    int32_t hold_filter_order;
    int32_t release_filter_order;
    double hold_filter_coefficient;
    double release_filter_coefficient;
} mgx_config;

/* Scalars that flow between the stages of stages.py:210-272 (what the reference
 * logs through debug()).  Filled by mgx_master / the stage calls. */
typedef struct mgx_report {
    double final_amplitude_coefficient;  /* match_levels.py:29-44 */
    double target_match_rms, reference_match_rms;
    double rms_coefficient;              /* stages.py:80-88 */
    double correction_coefficients[16];  /* stages.py:149-168, first rms_correction_steps entries */
    double normalize_coefficient;        /* stages.py:186-191 (0 when not requested) */
    double result_peak;                  /* max |result_no_limiter| */
    int32_t target_divisions, reference_divisions;
    int64_t target_piece, reference_piece;
    int32_t target_loud_count, reference_loud_count;
    int32_t limiter_active;              /* 0 = hyrax.py:83-85 early-out */
    int32_t reserved;
} mgx_report;

/* ---- library / device ---------------------------------------------------- */
int mgx_version(void);
const char* mgx_last_error(void);
int mgx_device_count(int* count);
/* PCI address of GPU `device` as the driver prints it ("0000:0d:00.0", NUL-terminated, capacity >= 16): which physical
 * GPU a rank really sits on -- the batch front end and bench.py put it beside every rank's numbers (core.py:32-121 has
 * no notion of devices; this is part of the new surface, like the handle). */
int mgx_device_pci_bus_id(int device, char* out, int32_t capacity);
int mgx_create(int device, mgx_handle** out);
int mgx_destroy(mgx_handle* h);
int mgx_config_default(mgx_config* cfg);         /* Config() defaults, defaults.py:61-84 */

/* device memory + transfers (the Python host has no other way to hold HBM) */
int mgx_malloc(mgx_handle* h, size_t bytes, void** dev);
int mgx_free(mgx_handle* h, void* dev);
int mgx_memcpy_h2d(mgx_handle* h, void* dev, const void* host, size_t bytes);
int mgx_memcpy_d2h(mgx_handle* h, void* host, const void* dev, size_t bytes);
int mgx_synchronize(mgx_handle* h);
/* Page-locked host memory and copies that return at once (ordered on the handle's stream; the host
 * buffer must stay valid and untouched until mgx_synchronize): what a host needs to overlap the
 * upload of pair k+1 and the download of pair k-1 with the kernels of pair k (loader.py:30-47 and
 * saver.py:27-33 sit on either side of stages.main; PCIe is the bound of the whole process()). */
int mgx_host_alloc(size_t bytes, void** host);
int mgx_host_free(void* host);
int mgx_memcpy_h2d_async(mgx_handle* h, void* dev, const void* host, size_t bytes);
int mgx_memcpy_d2h_async(mgx_handle* h, void* host, const void* dev, size_t bytes);
/* HIP-event timing on the handle's stream: bracket any sequence of calls */
int mgx_timer_start(mgx_handle* h);
int mgx_timer_stop(mgx_handle* h, float* milliseconds);

/* ---- the drop-in boundary: stages.main ------------------------------------ */
/* Replaces matchering/stages.py:210-272 `main(target, reference, config,
 * need_default, need_no_limiter, need_no_limiter_normalized)`.  target_dev /
 * reference_dev: (n,2) float32 in HBM.  Each non-null output receives (n_target,2)
 * float32 in HBM; a null output = the corresponding need_* flag False.  Inputs
 * are not modified.  Everything, the FIR design included, is queued on the
 * handle's stream without a host round trip: with report == NULL the call returns
 * before the GPU has finished (mgx_synchronize to wait); with a report it waits
 * itself and fills it.
 * Limits (MGX_ERR_UNSUPPORTED, never a silent approximation): fft_size in
 * [8, 65536]; lowess_it in [0, 64]; tracks up to 536 million frames (32-bit byte
 * offsets; 3.3 hours at 44.1 kHz); limiter hold / release filters (defaults.py:48-56
 * accepts any positive order, hyrax.py:61-73 runs it) of order 1 to 3 whose
 * transfer-function form is well conditioned: a filter of order n at fc Hz is refused
 * when the rounding noise of the reference's own float64 recursion, about
 * 1.1e-16 / (2 pi fc / fs)^(n - 1/2) of full scale, exceeds 1e-6 -- there is then no
 * well-defined output to be within 1e-5 of (at the default cut-offs: hold order 3
 * runs, release order 3 does not; tests/test_limiter_order3_conditioning.py). */
int mgx_master(mgx_handle* h, const float* target_dev, int64_t n_target,
               const float* reference_dev, int64_t n_reference, const mgx_config* cfg,
               float* result_dev, float* result_no_limiter_dev,
               float* result_no_limiter_normalized_dev, mgx_report* report);

/* ---- stage-level entry points (parity tests, custom pipelines) ------------- */
/* match_levels.py:134-161 analyze_levels (+ dsp.py:93-100 peak for the reference,
 * match_levels.py:29-44) and match_frequencies.py:30-42 __average_fft of the loud
 * pieces, in one pass.  Host outputs (any may be null): piece_rms[divisions],
 * loud[divisions] (0/1), avg_mid/avg_side[fft_size/2+1] = mean |rfft|/F over the
 * loud pieces of the (peak-normalised, if is_reference) track. */
int mgx_analyze(mgx_handle* h, const float* x_dev, int64_t n, const mgx_config* cfg,
                int is_reference, double* peak, double* amplitude_coefficient,
                double* match_rms, int32_t* divisions, int64_t* piece_size,
                double* piece_rms, int32_t* loud, double* avg_mid, double* avg_side);

/* match_frequencies.py:78-101 get_fir from averaged spectra (host, float64).
 * avg_target must already include the level gain of stages.py:90-91.  A host-side
 * cross-check of the design (a direct float64 evaluation, no precomputed operator)
 * for tests and tools: it needs no GPU and is NOT what mgx_master runs -- there the
 * same design happens on the device (k_fir_raw / k_fir_matvec / k_fir_taps). */
int mgx_design_fir(const mgx_config* cfg, const double* avg_target, const double* avg_reference,
                   double* taps, double* curve_raw, double* curve_smooth);

/* match_frequencies.py:104-119 convolve (fftconvolve "same" on mid and side,
 * then ms_to_lr): y = L/R result (n,2), y_mid (n) optional.  taps are host
 * float64 arrays of fft_size entries; `gain` scales both (stages.py:80-88 folded in). */
int mgx_convolve(mgx_handle* h, const float* x_dev, int64_t n, const double* fir_mid,
                 const double* fir_side, int32_t taps, double gain, float* y_dev, float* y_mid_dev,
                 double* peak);

/* One round of stages.py:149-160: piece RMS of clip(gain*mid, -1, 1) over the
 * target's piece grid.  Host output sumsq[divisions] = sum of squares per piece. */
int mgx_clipped_piece_sumsq(mgx_handle* h, const float* mid_dev, int64_t n, int64_t piece_size,
                            int32_t divisions, double gain, double* sumsq);

/* limiter/hyrax.py:78-99 limit(array*gain) * post_gain, float32 in/out in HBM.
 * active (host, may be null) receives 0 when the limiter early-outs. */
int mgx_limit(mgx_handle* h, const float* x_dev, int64_t n, const mgx_config* cfg, double gain,
              double post_gain, float* out_dev, int32_t* active);

/* dsp.py:89-90 amplify on interleaved frames: out = x * gain */
int mgx_scale(mgx_handle* h, const float* x_dev, int64_t n, double gain, float* out_dev);

/* Integer PCM at the boundary.  The reference reads and writes files through soundfile
 * (loader.py:35 sf.read, saver.py:27-33 sf.write), i.e. libsndfile converts between the file's integer
 * samples and floats on the host.  These two entry points do that conversion in HBM, so that the integer
 * samples -- half the bytes of float32 at 16 bits -- are what crosses PCIe: `samples` counts single
 * samples (2 per stereo frame), interleaved as in the file; bits = 16 (int16), 24 (three bytes per
 * sample, little-endian, packed) or 32 (int32).  Scaling as libsndfile: decode x = v / 2^(bits-1);
 * encode v = rint(x * (2^(bits-1) - 1)), clipped to the integer range, evaluated in float64.  Queued on
 * the handle's stream. */
/* dsp.py:49-54 count_max_peaks on interleaved float32 samples in HBM: the largest magnitude and the number
 * of samples numpy.isclose (rtol 1e-5, atol 1e-8) puts on it, either sign -- what checker.py:118-130 looks
 * at to warn about clipped or already limited targets.  Waits for the result. */
int mgx_peak_count(mgx_handle* h, const float* x_dev, int64_t samples, double* peak, int64_t* count);
int mgx_pcm_decode(mgx_handle* h, const void* pcm_dev, int64_t samples, int32_t bits, float* out_dev);
int mgx_pcm_encode(mgx_handle* h, const float* x_dev, int64_t samples, int32_t bits, void* pcm_dev);

/* Album mode (the use of the FIR broadcast): stages.main with the matching-EQ FIR
 * GIVEN instead of designed from this pair's spectra -- `fir_dev` = [2][fft_size] float32 in HBM, mid
 * taps then side taps, e.g. the table mgx_last_fir returns on the rank that designed it, after
 * mgx_comm_broadcast_f32 brought it here.  Levels are still matched per track (stages.py:80-91,
 * 138-170 unchanged); only match_frequencies.py:78-101 is replaced. */
int mgx_master_with_fir(mgx_handle* h, const float* target_dev, int64_t n_target, const float* reference_dev,
                        int64_t n_reference, const mgx_config* cfg, const float* fir_dev, float* result_dev,
                        float* result_no_limiter_dev, float* result_no_limiter_normalized_dev,
                        mgx_report* report);

/* A/B previews (matchering/preview_creator.py:30-94) on frames that are still in HBM.
 * mgx_window_energy: dsp.py:128-143 (strided_app_2d + batch_rms_2d): sum of squares over both channels of
 * every window of `size` frames taken every `step` frames (`size` > n: the whole track is the one window);
 * energy[w] for w < *count (host array of `capacity` doubles); the loudest window is the argmax (the square
 * root and the mean of dsp.py:80-86 are monotone).  Waits for the stream.
 * mgx_preview_cut: frames [begin, begin + size) clipped to +-clip_limit (dsp.py:109-110; <= 0: not clipped)
 * and faded in and out over `fade` frames (dsp.py:146-152, numpy.linspace(0, 1, fade)), written to
 * out_dev [size][2]; queued on the handle's stream. */
int mgx_window_energy(mgx_handle* h, const float* x_dev, int64_t n, int64_t size, int64_t step, double* energy,
                      int64_t capacity, int64_t* count);
int mgx_preview_cut(mgx_handle* h, const float* x_dev, int64_t n, int64_t begin, int64_t size, int64_t fade,
                    double clip_limit, float* out_dev);

/* Measurement aid (bench.py): with timing enabled, mgx_master brackets each of
 * its stages with HIP events recorded on the handle's own stream (no synchronisation is added);
 * mgx_stage_times waits for the stream and returns the device time of every stage of the LAST
 * mgx_master call in milliseconds, -1 for a stage that did not run.  The stages are those of
 * stages.py:210-272 with `convolve` (match_frequencies.py:104-119) and the limiter
 * (hyrax.py:78-99) on their own, since each is a single kernel launch. */
enum mgx_stage {
    MGX_STAGE_ANALYZE = 0,          /* match_levels.py:134-161 + match_frequencies.py:30-42, target AND
                                       reference: ONE launch of k_analyze */
    MGX_STAGE_DESIGN_FIR = 1,       /* match_levels.py:62-71, match_frequencies.py:45-101 */
    MGX_STAGE_FILTER_SPECTRA = 2,   /* transforms of the FIR pair the convolution multiplies by */
    MGX_STAGE_CONVOLVE = 3,         /* match_frequencies.py:104-119: ONE launch of k_conv */
    MGX_STAGE_CORRECT_LEVELS = 4,   /* stages.py:138-170 */
    MGX_STAGE_SCALE_OUTPUTS = 5,    /* stages.py:185-191 */
    MGX_STAGE_LIMIT = 6,            /* hyrax.py:78-99: ONE launch of the limiter kernel */
    MGX_STAGE_COUNT = 7
};
int mgx_stage_timing(mgx_handle* h, int32_t enable);
int mgx_stage_times(mgx_handle* h, float* ms /* [MGX_STAGE_COUNT] */);

/* Code bytes of the seven big kernel families (analyze, match_curve, conv_prep, conv, correction_round,
 * correction_tail, limit), bytes[family * 16 + variant] with variant = log2 of the transform size (0 / 1 for the
 * 256 / 1024-block limiter, 0 for the untemplated kernels, 15 in the conv family for the two-partition
 * delay-line kernel, 6 in the conv and conv_prep families for the N = 4F kernel of 4096 taps), as read from this
 * library's own device code
 * object -- what the kernels' first workgroups read as data to put their code into the L2 ahead of the
 * instruction cache ("fast and slow boxes").  Returns the number of families; needs no
 * GPU.  Zeros mean the code object could not be read and the kernels do not warm. */
int mgx_code_bytes(int32_t* bytes, int32_t capacity);

/* Device address of the FIR pair ([2][fft_size] float32: mid taps then side taps, level gain
 * not included) designed by the last mgx_master / uploaded by the last mgx_convolve on this
 * handle -- the payload of the RCCL exchange below. */
int mgx_last_fir(mgx_handle* h, void** taps_dev, int32_t* taps);

/* ---- multi-GPU: one process per GPU, FIR taps over RCCL/xGMI ---------------- */
int mgx_comm_unique_id(void* id128);                                  /* ncclGetUniqueId, 128 bytes */
int mgx_comm_init(mgx_handle* h, const void* id128, int rank, int world);
int mgx_comm_count(mgx_handle* h, int32_t* ranks);                    /* ncclCommCount: the ranks RCCL itself sees */
int mgx_comm_broadcast_f32(mgx_handle* h, float* dev, int64_t count, int root);
int mgx_comm_allgather_f32(mgx_handle* h, const float* send_dev, float* recv_dev, int64_t count);
int mgx_comm_destroy(mgx_handle* h);

#ifdef __cplusplus
}
#endif
#endif /* MGX_H */
