/*
 * stripenn_hip.h -- C ABI of libstripenn_hip.so, the MI355X (gfx950) implementation of
 * Stripenn's `compute` / `score` hot path.
 *
 * The reference (ysora/stripenn, pure Python) has no FFI; the drop-in boundary is the Python
 * class stripenn.getStripe.getStripe (src/stripenn/getStripe.py:17-1232).  Each entry point
 * below replaces the arithmetic of the reference method(s) cited next to it; the Python facade
 * stripenn_amd/getStripe.py keeps the reference's method names / arguments and calls these
 * through ctypes (see INTEGRATION.md for the binding a maintainer would add).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes.  Return 0 (STP_OK) or a negative STP_E_* code;
 *     stp_last_error(ctx) gives a message valid until the next call on that ctx.
 *   - Host pointers are caller-owned and only read/written during the call (calls are
 *     synchronous).  Device memory is library-owned behind opaque handles.
 *   - A ctx binds one HIP device and one stream; not thread-safe; one ctx per host thread /
 *     process (multi-GPU model: one process per GPU).
 *   - There is NO CPU fallback: every compute entry point fails with STP_E_HIP when no
 *     gfx950 device is usable.
 *   - All image stages are bit-exact with the reference; floating-point score outputs follow
 *     the tolerance stated per function.
 */
#ifndef STRIPENN_HIP_H
#define STRIPENN_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STP_ABI_VERSION 1
#define STP_FRAME_MAX 400   /* frames are at most 400 x 400 (getStripe.py:794-799) */
#define STP_NDIAG 400       /* diagonals kept for expected values / background (getStripe.py:219,305) */

enum {
    STP_OK = 0,
    STP_E_ARG = -1,
    STP_E_CAPACITY = -2,
    STP_E_HIP = -3,
    STP_E_NOMEM = -4,
    STP_E_UNSUPPORTED = -5
};

typedef struct stp_ctx stp_ctx;
typedef struct stp_band stp_band;
typedef struct stp_frames stp_frames;

int stp_version(void);
int stp_ctx_create(int device_ordinal, stp_ctx** out);
void stp_ctx_destroy(stp_ctx* ctx);
const char* stp_last_error(const stp_ctx* ctx);
/* Use an existing HIP stream (e.g. torch's current stream) for all launches; NULL = own stream. */
int stp_ctx_set_stream(stp_ctx* ctx, void* hip_stream);
int stp_ctx_synchronize(stp_ctx* ctx);

/* ---- contact-matrix band -------------------------------------------------------------
 * One chromosome as a dense diagonal band resident in HBM:
 *     band[i * (2*hw) + (d + hw)] = M[i][i + d],  d in [-hw, hw),  NaN preserved,
 *     entries outside the chromosome = 0.
 * Replaces the per-frame / per-stripe `self.unbalLib.fetch(...)` dense fetches
 * (getStripe.py:193,326,432,518,560,684,690,696,808): the host fetches each strip once.
 * halfwidth must be a multiple of 64 and >= 448 + 2*bs (512 covers 5 kb and 1 kb). */
int stp_band_upload(stp_ctx* ctx, const double* band_host, int64_t nrows, int32_t halfwidth, stp_band** out);
/* Adopt a band that already lives in device memory (caller keeps ownership of dptr). */
int stp_band_wrap_device(stp_ctx* ctx, const void* dptr, int64_t nrows, int32_t halfwidth, stp_band** out);
void stp_band_free(stp_ctx* ctx, stp_band* band);

/* ---- frames ---------------------------------------------------------------------------
 * search_frame's window + zero-column removal (getStripe.py:808-821) and StripeSearch's
 * medpixel (getStripe.py:885) for a batch of frames [start[f], end[f]] (inclusive bin indices,
 * end - start + 1 <= 400).  S[f] = number of kept columns, or 0 when <= 10 remain (:818).
 * nz[f*400 + k] = offset (0..399) of kept column k inside the frame. */
int stp_frames_create(stp_ctx* ctx, const stp_band* band, const int32_t* start, const int32_t* end,
                      int32_t nframes, stp_frames** out);
int stp_frames_info(stp_ctx* ctx, const stp_frames* fr, int32_t* S_out, int16_t* nz_out /* nframes*400 */,
                    double* medpixel_out /* nframes, may be NULL */);
void stp_frames_free(stp_ctx* ctx, stp_frames* fr);

/* ---- StripeSearch ------------------------------------------------------------------------
 * getStripe.StripeSearch (getStripe.py:864-1104) for every frame x maxpixel level x brightness:
 * image build (:889-895), ImageProcessing.imBrightness3D (ImageProcessing.py:8-41), cv.filter2D
 * mean blur + cv.cvtColor grey (:907-913), skimage.feature.canny (:917), ImageProcessing.verticalLine
 * (ImageProcessing.py:61-83), ImageProcessing.block per column (:924-940, ImageProcessing.py:100-195),
 * line joining (:942-1078) and the x/y/w/h/total of each candidate (:1081-1104).
 * Records come out in the reference's row order: frame, level, brightness, ud (1 then 2), X order. */
typedef struct {
    int32_t minH;        /* --minL  (getStripe.py:933) */
    int32_t maxW;        /* --maxW  (getStripe.py:1055) */
    int32_t bfilter;     /* --bfilter, odd, 1..7 (getStripe.py:907-909) */
    int32_t n_bright;    /* len(np.arange(0.5, 1.01, 0.1)) = 6 (getStripe.py:898), <= 8 */
    const double* bright;   /* the arange values, computed by the host with numpy */
    int32_t gauss_radius;   /* int(4*sigma + 0.5)  (scipy gaussian_filter1d), <= 12 */
    const double* gauss_w;  /* 2*radius+1 weights of scipy's _gaussian_kernel1d(sigma, 0, radius) */
} stp_search_params;

typedef struct {
    int32_t frame;       /* index into the frame batch */
    int32_t level;       /* index into M_levels */
    int32_t b_index;     /* brightness index */
    int32_t ud;          /* 1 = up, 2 = down (getStripe.py:937-940) */
    int32_t x, y, w, h;  /* compacted frame coordinates (getStripe.py:1082-1085) */
    double total;        /* submat[y:y+h, x:x+w].sum() (getStripe.py:1094) */
} stp_stripe_rec;

int stp_stripe_search(stp_ctx* ctx, const stp_frames* fr, const stp_search_params* prm,
                      const double* M_levels /* nframes_levels: M for [level] */, int32_t n_levels,
                      stp_stripe_rec* out, int64_t out_capacity, int64_t* out_count);

/* ---- stage-level entry points (parity tests; same kernels as stp_stripe_search) ----------
 * All operate on ONE image: frame `f` of `fr`, level value M, brightness index bi.
 * Buffers are S x S row-major (S = S[f]).  Any output pointer may be NULL. */
int stp_dbg_stages(stp_ctx* ctx, const stp_frames* fr, const stp_search_params* prm, int32_t f, double M,
                   int32_t bi, float* gray, uint8_t* cls /* 0,1 low,2 high */, uint8_t* edges,
                   uint8_t* vert, int32_t* col_t, int32_t* col_end, int32_t* col_ud,
                   uint8_t* testmat_ud1, uint8_t* testmat_ud2);

/* ---- statistics / profiling ---------------------------------------------------------------
 * When profiling is on, every kernel launch is bracketed by HIP events on the ctx stream. */
typedef struct {
    char name[32];
    int64_t launches;
    double ms_total;     /* sum of HIP-event elapsed times */
    double alg_bytes;    /* algorithmic bytes moved by those launches */
} stp_kernel_stat;
int stp_set_profiling(stp_ctx* ctx, int on);
int stp_get_stats(stp_ctx* ctx, stp_kernel_stat* out, int32_t capacity, int32_t* count);
int stp_reset_stats(stp_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif
