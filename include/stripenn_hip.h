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

#define STP_ABI_VERSION 3
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
/* Build the band on the device straight from cooler's pixel table (the data format one step before the
 * path; replaces `cooler.Cooler(cool).matrix(balance=norm)` + dense `.fetch`, stripenn.py:80-118):
 *   bin1_id <= bin2_id (global bin ids), count; pixels with either bin outside [bin_lo, bin_lo + nrows) or
 *   with |bin2 - bin1| > halfwidth are skipped; value = count * (bias[bin1] * bias[bin2]) (cooler's dense
 *   read multiplies the count block by np.outer(bias1, bias2); NaN gives NaN), or (double)count when
 *   weight == NULL; `weight` is the MULTIPLICATIVE bias: the caller passes 1 / w for cooler's divisive
 *   columns (KR, VC, SQRT_VC); each pixel is written at (i, j) and mirrored at (j, i); a cell without a
 *   stored pixel is the count 0 times the same product -- 0, or NaN along the whole row and column of a bin
 *   whose bias is NaN (cooler multiplies the DENSE block, so an unbalanced bin is NaN everywhere) -- and 0
 *   outside the chromosome.  weight has nbins_total entries (global bin ids).  No dense intermediate exists. */
int stp_band_pack(stp_ctx* ctx, const int64_t* bin1_id, const int64_t* bin2_id, const int32_t* count, int64_t npix,
                  const double* weight, int64_t nbins_total, int64_t bin_lo, int64_t nrows, int32_t halfwidth,
                  stp_band** out);
/* stp_band_pack that ALSO appends the chromosome's balanced pixel values to an order-statistic select (see
 * stp_select_append_pixels below; `sel` may be NULL): the maxpixel quantile (getStripe.py:160-176) and the band then
 * share one trip of the pixel table over PCIe.  Pass every cis pixel of the chromosome. */
typedef struct stp_select stp_select;
#define STP_COUNT_I32 0     /* pixels/count as int32 (cooler's default) */
#define STP_COUNT_F64 1     /* ... as float64: coolers written with --count-as-float, merged or scaled ones */
int stp_band_pack_select(stp_ctx* ctx, const int64_t* bin1_id, const int64_t* bin2_id, const void* count, int32_t count_type,
                         int64_t npix, const double* weight, int64_t nbins_total, int64_t bin_lo, int64_t nrows,
                         int32_t halfwidth, stp_select* sel, stp_band** out);
/* The same with the table's columns in their narrow form (round 5): bin1_id is sorted, so cooler keeps it as a CSR index too
 * (`indexes/bin1_offset`), and a bin id fits 32 bits.  bin1_offset[r], r = 0 .. nrows: position -- among the npix pixels
 * handed over -- of the first pixel of bin lo + r (bin1_offset[0] = 0, bin1_offset[nrows] = npix, non-decreasing); bin2_id as
 * int64 (STP_ID_I64) or int32 (STP_ID_I32).  The bin1_id column (8 of a pixel's 20 bytes) never crosses PCIe: the device expands
 * the index.  Every pixel's bin1 must lie in [lo, lo + nrows) by construction; pixels whose bin2 does not are skipped as in
 * stp_band_pack.  Same band, nearest-pixel table and select contents as stp_band_pack_select on the expanded columns. */
#define STP_ID_I64 0
#define STP_ID_I32 1
int stp_band_pack_csr(stp_ctx* ctx, const int64_t* bin1_offset, const void* bin2_id, int32_t bin2_type, const void* count, int32_t count_type,
                      int64_t npix, const double* weight, int64_t nbins_total, int64_t lo, int64_t nrows, int32_t hw, stp_select* select,
                      stp_band** out);

/* For a band built by stp_band_pack: the distance from every bin to the nearest stored pixel with a positive
 * value in its row of the symmetric matrix, to the right (column >= row; 0 = a positive diagonal pixel) and to
 * the left (column < row); INT32_MAX where there is none.  Every cis pixel handed to stp_band_pack takes part,
 * also those beyond the halfwidth.  With it the host answers "which rows of this block have a non-zero sum"
 * (the pools of getStripe.nulldist, getStripe.py:262-273, 329-331: `np.sum(mat, axis=1) != 0` over a dense
 * fetch) for any block whose columns cover its rows, without fetching anything: row i of the block
 * rows x [c0, c1) is non-empty iff i + right[i] < c1 or i - left[i] >= c0 (all values being >= 0).
 * STP_E_UNSUPPORTED for uploaded / wrapped bands. */
int stp_band_nearest(stp_ctx* ctx, const stp_band* band, int32_t* right_out /* nrows */, int32_t* left_out /* nrows */);
/* Copy a band back to the host (nrows x 2*halfwidth doubles; parity tests, debugging). */
int stp_band_download(stp_ctx* ctx, const stp_band* band, double* out_host);
/* Adopt a band that already lives in device memory (caller keeps ownership of dptr).  The memory must stay UNCHANGED for the life
 * of the handle: the library establishes once per handle whether the band is symmetric bit for bit (k_band_symcheck) and then
 * reads stripe patches through the symmetry (stp_score, stp_pvalue, stp_stripiness) and takes the Canny class maps of the
 * tiles below an image's diagonal from the tiles above it (stp_stripe_search); a band rewritten behind the handle would be read
 * under a stale verdict.  To search other data, free the handle and wrap again. */
int stp_band_wrap_device(stp_ctx* ctx, const void* dptr, int64_t nrows, int32_t halfwidth, stp_band** out);
void stp_band_free(stp_ctx* ctx, stp_band* band);

/* ---- frames ---------------------------------------------------------------------------
 * search_frame's window + zero-column removal (getStripe.py:808-821) and StripeSearch's
 * medpixel (getStripe.py:885) for a batch of frames [start[f], end[f]] (inclusive bin indices,
 * end - start + 1 <= 400).  S[f] = number of kept columns, or 0 when <= 10 remain (:818).
 * nz[f*400 + k] = offset (0..399) of kept column k inside the frame.
 * medpixel[f] = np.quantile(submat[submat > 0], 0.5): exact order statistics by radix select on
 * the device, numpy's linear interpolation applied by the library on the host. */
int stp_frames_create(stp_ctx* ctx, const stp_band* band, const int32_t* start, const int32_t* end,
                      int32_t nframes, stp_frames** out);
/* flags: STP_FRAMES_KEEP_ALL keeps every column of every frame (no zero-column removal, no "more than 10 columns"
 * rule): the frame IS the matrix a caller hands to getStripe.StripeSearch (getStripe.py:864), whose columns
 * search_frame has already compacted. */
#define STP_FRAMES_KEEP_ALL 1
int stp_frames_create_ex(stp_ctx* ctx, const stp_band* band, const int32_t* start, const int32_t* end,
                         int32_t nframes, int32_t flags, stp_frames** out);
int stp_frames_info(stp_ctx* ctx, const stp_frames* fr, int32_t* S_out, int16_t* nz_out /* nframes*400 */,
                    double* medpixel_out /* nframes, may be NULL */);
/* Frame overlap (round 6).  search_frame's windows advance by 200 bins and are 400 wide (getStripe.py:794-799), so the trailing block
 * of frame i is the leading block of frame i + 1 and the reference computes the images of that block twice.  shift_out[i] = index
 * (in frame i's compacted coordinates) where the block shared with frame i + 1 starts when both compacted frames keep exactly
 * the same bins of it, else -1: the search then takes the class maps of the block's interior (Gaussian radius + 3 pixels from
 * its border) for frame i from frame i + 1 instead of computing them (STP_REUSE=0 in the environment: always computed). */
int stp_frames_overlap(stp_ctx* ctx, const stp_frames* fr, int32_t* shift_out /* nframes */);
void stp_frames_free(stp_ctx* ctx, stp_frames* fr);

/* ---- StripeSearch ------------------------------------------------------------------------
 * getStripe.StripeSearch (getStripe.py:864-1104) for every frame x maxpixel level x brightness:
 * image build (:889-895), ImageProcessing.imBrightness3D (ImageProcessing.py:8-41), cv.filter2D
 * mean blur + cv.cvtColor grey (:907-913), skimage.feature.canny (:917), ImageProcessing.verticalLine
 * (ImageProcessing.py:61-83), ImageProcessing.block per column (:924-940, ImageProcessing.py:100-195),
 * line joining (:942-1078) and the x/y/w/h/total of each candidate (:1081-1104).
 * Records come out in the reference's row order: frame, level, brightness, ud (1 then 2), X order.
 * What the reference computes twice is computed once, with its results: in a band that is symmetric bit for
 * bit the Canny class maps of the tiles below an image's diagonal are the transposes of the tiles above it (proved for the f32
 * verdicts, settled per position for the undecidable pixels); the block of bins two consecutive frames of the batch share is
 * taken from the later frame where both keep the same bins of it (stp_frames_overlap).  Environment: STP_SYM=0 / STP_REUSE=0
 * switch these off, STP_CANNY=exact / STP_GRAY=exact select the kernels that perform every operation of the reference. */
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

/* The same search without a host round trip per call: `begin` enqueues every kernel of the batch on the ctx stream and
 * returns at once (the frames, and the band behind them, must stay alive until the search is fetched or cancelled);
 * `count` blocks until the batch is done and gives the number of records; `fetch` copies them out (capacity >= count)
 * and releases the search, also on error.  Several searches may be in flight on one context: they run in order on its
 * stream and share its image workspace, so the host can prepare / consume one batch while the device works on the
 * next (the reference runs frames strictly one after the other, getStripe.py:829-842). */
typedef struct stp_search stp_search;
int stp_stripe_search_begin(stp_ctx* ctx, const stp_frames* fr, const stp_search_params* prm, const double* M_levels,
                            int32_t n_levels, stp_search** out);
int stp_stripe_search_count(stp_ctx* ctx, stp_search* search, int64_t* out_count);
int stp_stripe_search_fetch(stp_ctx* ctx, stp_search* search, stp_stripe_rec* out, int64_t out_capacity);
void stp_stripe_search_cancel(stp_ctx* ctx, stp_search* search);

/* Parity tests: the sweep grants every image 128 record slots and re-runs a chunk of images with 400 slots (one
 * per possible column pair) when some image needs more -- which no contact map has been seen to do.  This shrinks
 * the first pass to `slots` (1..128) so that the re-run path executes on ordinary data; results must not change. */
int stp_dbg_set_sweep_slots(stp_ctx* ctx, int32_t slots);

/* ---- stage-level entry points (parity tests; same kernels as stp_stripe_search) ----------
 * All operate on ONE image: frame `f` of `fr`, level value M, brightness index bi.
 * Buffers are S x S row-major (S = S[f]).  Any output pointer may be NULL. */
int stp_dbg_stages(stp_ctx* ctx, const stp_frames* fr, const stp_search_params* prm, int32_t f, double M,
                   int32_t bi, float* gray, uint8_t* cls /* 0,1 low,2 high */, uint8_t* edges,
                   uint8_t* vert, int32_t* col_t, int32_t* col_end, int32_t* col_ud,
                   uint8_t* testmat_ud1, uint8_t* testmat_ud2);

/* Device-side evidence for the f32 Canny kernel's error budget (k_canny_f32, stp_canny32.h; replaces nothing of the
 * reference: skimage _canny.py:53-297 is what the kernel's classes must equal): what the tiles of image (f, M, bi) computed in
 * f32.  planes: 6 x S x S floats -- smoothed value, the two Sobel sums, magnitude, the grey scale g and the Sobel budget E_G
 * (in units of 2^-24 g) the pixel's tile used; NaN where the tile was skipped as flat.  counts: candidates (magnitude above
 * the lowered low threshold), pixels sent to the exact f64 resolver, tile-images handed to the all-f64 kernel. */
int stp_dbg_canny_f32(stp_ctx* ctx, const stp_frames* fr, const stp_search_params* prm, int32_t f, double M, int32_t bi,
                      float* planes, int64_t* counts);

/* ---- expected values: getStripe.mpmean (getStripe.py:178-235) --------------------------------
 * For every 400-row frame f of the chromosome and diagonal j < 400:
 *   part_sum[f*400 + j] = sum_i M[i][i+j] over the frame's rows i with i + j < nrows (NaN -> 0),
 *                         accumulated in row order exactly like the reference's inner loop (:198-207)
 *   part_cnt[f*400 + j] = number of terms.
 * The host adds the frames in order and divides (:225-233).  n400 = ceil(nrows / 400). */
int stp_diag_sums(stp_ctx* ctx, const stp_band* band, double* part_sum, int64_t* part_cnt, int32_t n400);

/* ---- background windows: getStripe.nulldist (getStripe.py:347-378, 389-414, 447-477) ---------
 * One sample = one randomly chosen row x of the unit matrix mat = M[row0:row0+nrow, col0:col0+ncol]
 * (the reference's fetch at :326, NaN -> 0).  For j = 0..399 the six bs x bs window means around
 * columns x -/+ j (+ yoff) are taken with Python slice semantics on mat and numpy's np.mean
 * summation order, and the four "centre - flank" tables are written: T[j*n + i], T in
 * {left_up, right_up, left_down, right_down}.  Sampling (random.Random) stays on the host. */
typedef struct {
    int32_t row0, nrow;   /* mat rows    = bins [row0, row0 + nrow) */
    int32_t col0, ncol;   /* mat columns = bins [col0, col0 + ncol) */
    int32_t x;            /* sampled row index inside mat */
    int32_t yoff;         /* 400 for units > 0, else 0 (getStripe.py:358-360) */
} stp_null_sample;
/* unit_matrix (may be NULL): the host-fetched mat itself (nrow x ncol, NaN preserved), to be given
 * when some window's Python slice wraps around (negative start: the reference's top-up branch has
 * no 410-row margin, getStripe.py:438-440, and at 1 kb the 20-row margin is too small) -- such a
 * window reads columns far outside the diagonal band.  With NULL the resident band is read. */
int stp_null_windows(stp_ctx* ctx, const stp_band* band, const double* unit_matrix, const stp_null_sample* samples,
                     int32_t n, int32_t bs, double* left_up, double* right_up, double* left_down, double* right_down);

/* ---- background tables resident on the device ---------------------------------------------- */
typedef struct stp_background stp_background;
int stp_background_upload(stp_ctx* ctx, const double* left_up, const double* right_up, const double* left_down,
                          const double* right_down, int32_t ncol, stp_background** out); /* each 400 x ncol */
void stp_background_free(stp_ctx* ctx, stp_background* bg);

/* ---- per-stripe p-value: getStripe.pvalue (getStripe.py:536-606) ----------------------------
 * mat = M[row0:row1, col0:col1] (the fetch at :560); centre / left / right = mat[:, bs:-bs],
 * mat[:, :bs], mat[:, -bs:]; row means in numpy order; rank against the background rows; median.
 * The host decides the direction exactly like :584-597, including the inherited ("fixed")
 * background rows when a stripe touches neither end of the diagonal. */
typedef struct {
    int32_t row0, row1, col0, col1;
    int32_t mode;       /* 0 down: table *_down, row j; 1 up: table *_up, row upbase-j-1; 2 fixed */
    int32_t upbase;     /* y2 - y1 (getStripe.py:592) */
    int32_t fixed_row;  /* mode 2 */
    int32_t fixed_tab;  /* mode 2: 0 = *_up tables, 1 = *_down tables */
} stp_pv_stripe;
int stp_pvalue(stp_ctx* ctx, const stp_band* band, const stp_background* bg, int32_t bs,
               const stp_pv_stripe* stripes, int64_t n, double* out_p);

/* ---- Stripiness: getStripe.scoringstripes + stats.elementwise_product_sum --------------------
 * (getStripe.py:661-759, stats.py:184-199).  Block b = 0 centre, 1 left, 2 right.
 * Output tolerance: 1e-4 relative (floating-point statistics; order of additions may differ). */
typedef struct {
    int32_t row0, row1;           /* observed rows [row0, row1) = y_coord fetch            */
    int32_t col0[3], col1[3];     /* observed columns of each block                         */
    int32_t ex0[3];               /* first x index of each block's expected matrix (:685,691,697) */
    int32_t ey0;                  /* y_start_index                                          */
    int32_t mirror;               /* 0 if xs == ys else 1 (row deletion rule :716-731)       */
    int32_t mcol0[3], mcol1[3];   /* masked relative column range [lo, hi] per block; lo > hi: none */
    int32_t mrow0, mrow1;         /* masked relative row range                               */
} stp_score_stripe;
/* out_status (may be NULL): 1 when an all-NaN column maps to a row index outside the stripe -- the
 * reference's np.delete raises IndexError there (getStripe.py:735); the facade re-raises it. */
int stp_stripiness(stp_ctx* ctx, const stp_band* band, const double* exval400, const stp_score_stripe* stripes,
                   int64_t n, double* out_g, double* out_oe_mean, double* out_oe_total, int32_t* out_status);

/* ---- p-value AND Stripiness of the same stripes in one call (one upload, one launch, one download): what
 * score.getScore (score.py:52-55) and a driver that scores candidates straight from the search do back to back.
 * pv_stripes[i] / sc_stripes[i] describe stripe i as stp_pvalue / stp_stripiness take it; outputs as theirs.  Rows whose two
 * descriptors are byte-identical to an earlier row's are not scored again: they receive that row's results (the outputs are a
 * function of the descriptors alone; a search returns the same rectangle for several brightness images, levels and frames). */
int stp_score(stp_ctx* ctx, const stp_band* band, const stp_background* bg, int32_t bs, const double* exval400,
              const stp_pv_stripe* pv_stripes, const stp_score_stripe* sc_stripes, int64_t n, double* out_p, double* out_g,
              double* out_oe_mean, double* out_oe_total, int32_t* out_status);

/* ---- observed mean / sum: getStripe.getMean (getStripe.py:501-534) -------------------------- */
typedef struct {
    int32_t row0, row1, col0, col1;
} stp_rect;
int stp_stripe_mean(stp_ctx* ctx, const stp_band* band, const stp_rect* rects, int64_t n, double* out_mean,
                    double* out_sum);

/* ---- heat-map plane of a window: seeimage (seeimage.py:74-85) -----------------------------------
 * out[r * ncols + c] = clip((255 * (M - A[r][c]) / M) / 255, 0, 1) for the window A = M[row0:row0+nrows, col0:col0+ncols]
 * of the resident band -- the image-build arithmetic of StripeSearch (getStripe.py:889-895) -- NaN pixels stay NaN.
 * The caller stacks it as the green and blue channels under a constant red 1.  The window must lie inside the band. */
int stp_window_plane(stp_ctx* ctx, const stp_band* band, int64_t row0, int32_t nrows, int64_t col0, int32_t ncols, double M,
                     double* out);

/* ---- order statistics of the positive pixels: getStripe.getQuantile_original ------------------
 * (getStripe.py:160-176: `np.quantile(mat[mat > 0], quantile)` over the whole chromosome.)
 * The host streams the chromosome in row strips (`stp_select_append`; non-positive and NaN entries
 * are ignored, exactly like `mat[mat > 0]`), so the dense chromosome (12 GB for chr1 at 5 kb) never
 * exists; `stp_select_ranks` returns the exact order statistics a[rank] (0-based, ascending) by
 * radix select -- all ranks of a call descend together: one sweep over the values per 11-bit digit, one host round
 * trip per 16 ranks; numpy's interpolation between them is applied by the caller (stripenn_amd/getStripe.py). */
int stp_select_create(stp_ctx* ctx, stp_select** out);
int stp_select_append(stp_ctx* ctx, stp_select* sel, const double* values_host, int64_t n);
/* Append the balanced values of cooler pixels (bin1_id <= bin2_id, count; value = count * (bias[bin1] * bias[bin2]), or
 * (double)count when weight == NULL; `weight` is the multiplicative bias as in stp_band_pack) as the dense symmetric matrix would hold them: an off-diagonal pixel counts
 * twice (getStripe.py:160-176 takes the quantile over the full square matrix).  The values are formed on the
 * device from the table columns; nothing dense and no host-side product array exists. */
int stp_select_append_pixels(stp_ctx* ctx, stp_select* sel, const int64_t* bin1_id, const int64_t* bin2_id,
                             const int32_t* count, int64_t npix, const double* weight, int64_t nbins_total);
int stp_select_append_pixels_ex(stp_ctx* ctx, stp_select* sel, const int64_t* bin1_id, const int64_t* bin2_id,
                                const void* count, int32_t count_type /* STP_COUNT_* */, int64_t npix, const double* weight,
                                int64_t nbins_total);
int stp_select_count(stp_ctx* ctx, stp_select* sel, int64_t* n_positive);
int stp_select_ranks(stp_ctx* ctx, stp_select* sel, const int64_t* ranks, int32_t nranks, double* out);
void stp_select_free(stp_ctx* ctx, stp_select* sel);

/* ---- redundancy filter: getStripe.RemoveRedundant (getStripe.py:1116-1196) ---------------------
 * For every pair of rows of one chromosome whose frame numbers differ by at most one (the reference
 * groups rows with num in {n, n+1}, :1119-1121), with A the row that comes first in the table:
 *   s_x = |[max(x0), min(x1)]| / min(A.x1 - A.x0, B.x1 - B.x0),  s_y likewise;  if both > 0.2:
 *   by 0 'size'  : drop A if A.h/A.w <= B.h/B.w else B          (:1147-1150)
 *   by 1 'score' : drop A if A.key   <= B.key   else B          (:1152-1155)
 *   by 2 'pvalue': drop A if A.key   >  B.key   else B          (:1157-1160)
 * Deletions are only collected (a dropped row still knocks out others), so the pair tests are
 * independent.  order[] = row indices sorted by (chromosome, num) (stable); b0/b1/b2[i] = start of
 * row i's (chr, num) bucket, its end = start of the (chr, num+1) bucket, and that bucket's end, as
 * positions in order[] (b1 == b2 when the next frame number is absent).  keep[i] = 0/1. */
int stp_remove_redundant(stp_ctx* ctx, int64_t n, const int64_t* pos1, const int64_t* pos2, const int64_t* pos3,
                         const int64_t* pos4, const int32_t* h, const int32_t* w, const double* key, int32_t by,
                         const int32_t* order, const int32_t* b0, const int32_t* b1, const int32_t* b2,
                         uint8_t* keep);

/* ---- result tables as text: `df.to_csv(path, sep='\t', header=True, index=False)` (stripenn.py:156-157, score.py:60) ----
 * Formats nrows rows of ncols columns as tab-separated lines ('\n' behind every row; no header) into out[0 .. cap).
 * kind[c] = STP_COL_I64: data[c] is const int64_t[nrows], written in decimal;
 *           STP_COL_F64: const double[nrows], written as pandas writes a float64 column -- Python's repr(float): the shortest
 *                        digits that round-trip, exponent form below 1e-4 and from 1e16 on, ".0" behind an integral value --
 *                        NaN as the empty field;
 *           STP_COL_STR: const int32_t[nrows] codes 0 .. nstr[c]-1 into the column's string table: string k is the bytes
 *                        strtab[c][stroff[c][k] .. stroff[c][k+1]) (the caller has checked that none needs quoting).
 * strtab / stroff / nstr are read for STP_COL_STR columns only.  *out_len = bytes written, or, with STP_E_CAPACITY, nothing
 * usable.  Host code only (no context, no device): `stripenn_amd.stripenn.write_tsv` calls it and keeps a Python writer for
 * tables it does not describe (mixed cells, fields that need quoting). */
enum { STP_COL_I64 = 0, STP_COL_F64 = 1, STP_COL_STR = 2 };
int stp_format_tsv(int32_t ncols, const int32_t* kind, const void* const* data, const char* const* strtab,
                   const int64_t* const* stroff, const int32_t* nstr, int64_t nrows, char* out, int64_t cap, int64_t* out_len);

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
