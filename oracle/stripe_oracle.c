/*
 * stripe_oracle.c -- CPU restatement of Stripenn's `compute`/`score` hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity checker for the HIP library
 * (stripenn_amd/csrc).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; the product path never does.
 *
 * Every function cites the reference lines it restates (paths relative to
 * /root/reference/src/stripenn/) or the pinned third-party routine whose published
 * algorithm it restates (scikit-image 0.18.3 feature/_canny.py, scipy 1.7.1
 * ndimage/src/ni_filters.c NI_Correlate1D, scipy.signal.convolve2d).
 * Pinning: tests/test_oracle_golden.py checks each stage against vectors produced by
 * the unmodified reference (oracle/refharness/gen_golden.py).  The three OpenCV calls are
 * "parity unpinned": cv2 is absent here; oracle/refharness/standins/cv2.py is the spec.
 *
 * Plain C99, scalar, one rounding per source-level operation: build with
 * -O2 -ffp-contract=off (see oracle/Makefile).  Row-major S x S images.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define SO_API __attribute__((visibility("default")))

typedef struct {
    int32_t b_index;  /* brightness level index 0..nb-1            */
    int32_t ud;       /* 1 = upward stripe, 2 = downward            */
    int32_t x, y, w, h; /* compacted frame coordinates (getStripe.py:1082-1085) */
    double total;     /* submat[y:y+h, x:x+w].sum() (getStripe.py:1094) */
} so_rec;

/* ------------------------------------------------------------------ */
/* a-3 image build: getStripe.py:889-895                               */
/*   blue = 255*(M-D)/M ; blue<0 -> 0 ; g = clip(blue/255, 0, 1)       */
SO_API void so_gplane(const double* D, int64_t n, double M, double* g)
{
    for (int64_t i = 0; i < n; i++) {
        double t = 255.0 * (M - D[i]);
        t = t / M;
        if (t < 0.0) t = 0.0;
        double v = t / 255.0;
        /* np.clip(img, 0, 1) == minimum(maximum(v, 0), 1) */
        if (v < 0.0) v = 0.0;
        if (v > 1.0) v = 1.0;
        g[i] = v;
    }
}

/* a-4 ImageProcessing.imBrightness3D, ImageProcessing.py:15-31, for the G/B channel with
 * In=(0, b), Out=(0, 1): k = (1.0-0.0)/(b-0.0); <=0 -> 0 ; >b -> 1 ; else k*(v-0.0)+0.0.
 * A NaN pixel matches no branch and keeps imgOut's initial 0 (ImageProcessing.py:16). */
static inline double so_bright(double v, double b, double k)
{
    if (v <= 0.0) return 0.0;
    if (v > b) return 1.0;
    if (v > 0.0 && v <= b) return k * (v - 0.0) + 0.0;
    return 0.0;
}

static inline int so_reflect101(int i, int n)
{
    /* BORDER_REFLECT_101: gfedcb|abcdefgh|gfedcba ; valid for |overshoot| < n */
    if (n == 1) return 0;
    while (i < 0 || i >= n) {
        if (i < 0) i = -i;
        if (i >= n) i = 2 * (n - 1) - i;
    }
    return i;
}

/* a-4 + a-5: brightness -> bf x bf mean filter (cv.filter2D, getStripe.py:907-910) ->
 * clip -> float32 -> RGB2GRAY (getStripe.py:913).  R plane is the constant 1 pushed
 * through the same arithmetic; G == B. */
SO_API void so_gray(const double* g, int S, double b, int bf, float* gray)
{
    double k = (1.0 - 0.0) / (b - 0.0);
    double kv = 1.0 / (double)(bf * bf);  /* np.ones((bf,bf)) / (bf*bf) */
    int a = bf / 2;
    double* adj = (double*)malloc(sizeof(double) * (size_t)S * S);
    for (int64_t i = 0; i < (int64_t)S * S; i++) adj[i] = so_bright(g[i], b, k);
    /* red channel: imgconvert(1.0; low 0, high 1): mid branch k_r*(1-0)+0 with k_r = 1 */
    double radj = 1.0 * (1.0 - 0.0) + 0.0;
    double rb = 0.0;
    for (int t = 0; t < bf * bf; t++) rb = rb + kv * radj;
    if (rb < 0.0) rb = 0.0;
    if (rb > 1.0) rb = 1.0;
    float r32 = (float)rb;
    for (int y = 0; y < S; y++) {
        for (int x = 0; x < S; x++) {
            double acc = 0.0;
            for (int ky = 0; ky < bf; ky++) {
                int yy = so_reflect101(y + ky - a, S);
                for (int kx = 0; kx < bf; kx++) {
                    int xx = so_reflect101(x + kx - a, S);
                    acc = acc + kv * adj[(int64_t)yy * S + xx];
                }
            }
            if (acc < 0.0) acc = 0.0;
            if (acc > 1.0) acc = 1.0;
            float g32 = (float)acc;
            float v = r32 * 0.299f;
            v = v + g32 * 0.587f;
            v = v + g32 * 0.114f;
            gray[(int64_t)y * S + x] = v;
        }
    }
    free(adj);
}

/* so_gray with the DETERMINISTIC alternatives a real OpenCV build may use for the two unpinned calls (cv2 is absent;
 * oracle/refharness/standins/cv2.py is the specification) -- for tools/cv2_ambiguity.py only, never a parity target:
 *   box   != 0: cv.filter2D accumulates with a fused multiply-add, acc = fma(k, px, acc) (an -mfma / AVX2 build);
 *   order == 1: RGB2GRAY as fma(B, cb, fma(G, cg, R * cr));  order == 2: fma(R, cr, fma(G, cg, B * cb)). */
SO_API void so_gray_alt(const double* g, int S, double b, int bf, int box, int order, float* gray)
{
    double k = (1.0 - 0.0) / (b - 0.0);
    double kv = 1.0 / (double)(bf * bf);
    int a = bf / 2;
    double* adj = (double*)malloc(sizeof(double) * (size_t)S * S);
    for (int64_t i = 0; i < (int64_t)S * S; i++) adj[i] = so_bright(g[i], b, k);
    double radj = 1.0 * (1.0 - 0.0) + 0.0;
    double rb = 0.0;
    for (int t = 0; t < bf * bf; t++) rb = box ? fma(kv, radj, rb) : rb + kv * radj;
    if (rb < 0.0) rb = 0.0;
    if (rb > 1.0) rb = 1.0;
    float r32 = (float)rb;
    for (int y = 0; y < S; y++) {
        for (int x = 0; x < S; x++) {
            double acc = 0.0;
            for (int ky = 0; ky < bf; ky++) {
                int yy = so_reflect101(y + ky - a, S);
                for (int kx = 0; kx < bf; kx++) {
                    int xx = so_reflect101(x + kx - a, S);
                    acc = box ? fma(kv, adj[(int64_t)yy * S + xx], acc) : acc + kv * adj[(int64_t)yy * S + xx];
                }
            }
            if (acc < 0.0) acc = 0.0;
            if (acc > 1.0) acc = 1.0;
            float g32 = (float)acc, v;
            if (order == 1) v = fmaf(g32, 0.114f, fmaf(g32, 0.587f, r32 * 0.299f));
            else if (order == 2) v = fmaf(r32, 0.299f, fmaf(g32, 0.587f, g32 * 0.114f));
            else { v = r32 * 0.299f; v = v + g32 * 0.587f; v = v + g32 * 0.114f; }
            gray[(int64_t)y * S + x] = v;
        }
    }
    free(adj);
}

/* ------------------------------------------------------------------ */
/* scipy 1.7.1 NI_Correlate1D, symmetric odd kernel, mode='constant' cval=0:
 *   o = x[0]*w[0]; for k = r..1: o += (x[-k] + x[+k]) * w[k]      (outermost tap first) */
static inline double so_corr_sym(const double* line /* centre */, const double* w, int r)
{
    double o = line[0] * w[r];
    for (int k = r; k >= 1; k--) o += (line[-k] + line[k]) * w[r - k];
    return o;
}

/* gaussian_filter(sigma, mode='constant', truncate=4): axis 0 then axis 1, double line
 * buffers; result of each axis stored in the array dtype (float32 here when f32out). */
static void so_gauss2d(const double* in, int S, const double* w, int r, int f32out, double* out)
{
    double* buf = (double*)calloc((size_t)S + 2 * r, sizeof(double));
    double* tmp = (double*)malloc(sizeof(double) * (size_t)S * S);
    for (int x = 0; x < S; x++) {           /* axis 0: lines are columns */
        for (int y = 0; y < S; y++) buf[r + y] = in[(int64_t)y * S + x];
        for (int y = 0; y < S; y++) {
            double o = so_corr_sym(buf + r + y, w, r);
            tmp[(int64_t)y * S + x] = f32out ? (double)(float)o : o;
        }
    }
    for (int y = 0; y < S; y++) {           /* axis 1: lines are rows */
        for (int x = 0; x < S; x++) buf[r + x] = tmp[(int64_t)y * S + x];
        for (int x = 0; x < S; x++) {
            double o = so_corr_sym(buf + r + x, w, r);
            out[(int64_t)y * S + x] = f32out ? (double)(float)o : o;
        }
    }
    free(buf);
    free(tmp);
}

static inline int so_reflect(int i, int n)
{
    /* scipy 'reflect' (d c b a | a b c d | d c b a): only +-1 overshoot needed */
    if (i < 0) return -i - 1;
    if (i >= n) return 2 * n - 1 - i;
    return i;
}

/* ndi.sobel(a, axis): correlate1d([-1,0,1], axis) then correlate1d([1,2,1]) on the other
 * axis, mode='reflect'.  Antisymmetric branch: o = x[0]*0 + (x[-1]-x[1])*(-1);
 * symmetric branch: o = x[0]*2 + (x[-1]+x[1])*1. */
static void so_sobel(const double* a, int S, int axis, double* out)
{
    double* d = (double*)malloc(sizeof(double) * (size_t)S * S);
    for (int y = 0; y < S; y++)
        for (int x = 0; x < S; x++) {
            double m1, p1, c = a[(int64_t)y * S + x];
            if (axis == 1) {
                m1 = a[(int64_t)y * S + so_reflect(x - 1, S)];
                p1 = a[(int64_t)y * S + so_reflect(x + 1, S)];
            } else {
                m1 = a[(int64_t)so_reflect(y - 1, S) * S + x];
                p1 = a[(int64_t)so_reflect(y + 1, S) * S + x];
            }
            double o = c * 0.0;
            o += (m1 - p1) * -1.0;
            d[(int64_t)y * S + x] = o;
        }
    for (int y = 0; y < S; y++)
        for (int x = 0; x < S; x++) {
            double m1, p1, c = d[(int64_t)y * S + x];
            if (axis == 1) {  /* smooth along axis 0 */
                m1 = d[(int64_t)so_reflect(y - 1, S) * S + x];
                p1 = d[(int64_t)so_reflect(y + 1, S) * S + x];
            } else {
                m1 = d[(int64_t)y * S + so_reflect(x - 1, S)];
                p1 = d[(int64_t)y * S + so_reflect(x + 1, S)];
            }
            double o = c * 2.0;
            o += (m1 + p1) * 1.0;
            out[(int64_t)y * S + x] = o;
        }
    free(d);
}

/* skimage 0.18.3 feature.canny(gray_f32, sigma) with mask=None, thresholds 0.1 / 0.2
 * (_canny.py:53-297).  `w` = scipy _gaussian_kernel1d(sigma, 0, r) computed by the caller
 * with numpy exactly as scipy does (host side, not restated here).
 * Optional debug outputs (may be NULL): smoothed, isobel, jsobel, magnitude (f64 SxS),
 * cls (0 none / 1 low / 2 high local maxima). */
SO_API void so_canny(const float* gray, int S, const double* w, int r, uint8_t* edges,
                     double* o_smoothed, double* o_isobel, double* o_jsobel, double* o_mag,
                     uint8_t* o_cls)
{
    int64_t n = (int64_t)S * S;
    double* img = (double*)malloc(sizeof(double) * n);
    double* ones = (double*)malloc(sizeof(double) * n);
    double* sm = (double*)malloc(sizeof(double) * n);
    double* bleed = (double*)malloc(sizeof(double) * n);
    double* is = (double*)malloc(sizeof(double) * n);
    double* js = (double*)malloc(sizeof(double) * n);
    double* mag = (double*)malloc(sizeof(double) * n);
    uint8_t* cls = (uint8_t*)calloc(n, 1);
    for (int64_t i = 0; i < n; i++) { img[i] = (double)gray[i]; ones[i] = 1.0; }
    so_gauss2d(img, S, w, r, 1, sm);      /* float32 image: rounded to f32 after each axis */
    so_gauss2d(ones, S, w, r, 0, bleed);  /* mask.astype(float): float64 throughout */
    for (int64_t i = 0; i < n; i++) sm[i] = sm[i] / (bleed[i] + DBL_EPSILON);
    so_sobel(sm, S, 1, js);
    so_sobel(sm, S, 0, is);
    for (int64_t i = 0; i < n; i++) mag[i] = hypot(is[i], js[i]);

    for (int y = 1; y < S - 1; y++)
        for (int x = 1; x < S - 1; x++) {
            int64_t p = (int64_t)y * S + x;
            double m = mag[p];
            if (!(m > 0.0)) continue;           /* eroded_mask & (magnitude > 0) */
            double gi = is[p], gj = js[p], ai = fabs(gi), aj = fabs(gj);
            int same = (gi >= 0 && gj >= 0) || (gi <= 0 && gj <= 0);
            int opp = (gi <= 0 && gj >= 0) || (gi >= 0 && gj <= 0);
            int lm = 0;
            double c1, c2, wq;
            int cp, cm;
            if (same && ai >= aj) {             /* 0-45 deg */
                wq = aj / ai;
                c1 = mag[p + S]; c2 = mag[p + S + 1];
                cp = (c2 * wq + c1 * (1 - wq)) <= m;
                c1 = mag[p - S]; c2 = mag[p - S - 1];
                cm = (c2 * wq + c1 * (1 - wq)) <= m;
                lm = cp && cm;
            }
            if (same && ai <= aj) {             /* 45-90 deg (overrides) */
                wq = ai / aj;
                c1 = mag[p + 1]; c2 = mag[p + S + 1];
                cp = (c2 * wq + c1 * (1 - wq)) <= m;
                c1 = mag[p - 1]; c2 = mag[p - S - 1];
                cm = (c2 * wq + c1 * (1 - wq)) <= m;
                lm = cp && cm;
            }
            if (opp && ai <= aj) {              /* 90-135 deg */
                wq = ai / aj;
                c1 = mag[p + 1]; c2 = mag[p - S + 1];
                cp = (c2 * wq + c1 * (1.0 - wq)) <= m;
                c1 = mag[p - 1]; c2 = mag[p + S - 1];
                cm = (c2 * wq + c1 * (1.0 - wq)) <= m;
                lm = cp && cm;
            }
            if (opp && ai >= aj) {              /* 135-180 deg */
                wq = aj / ai;
                c1 = mag[p - S]; c2 = mag[p - S + 1];
                cp = (c2 * wq + c1 * (1 - wq)) <= m;
                c1 = mag[p + S]; c2 = mag[p + S - 1];
                cm = (c2 * wq + c1 * (1 - wq)) <= m;
                lm = cp && cm;
            }
            if (lm) cls[p] = (m >= 0.2) ? 2 : ((m >= 0.1) ? 1 : 0);
        }

    /* hysteresis: 8-connected components of low (cls>=1) that contain a high (cls==2) */
    memset(edges, 0, n);
    int64_t* stack = (int64_t*)malloc(sizeof(int64_t) * n);
    for (int64_t s0 = 0; s0 < n; s0++) {
        if (cls[s0] != 2 || edges[s0]) continue;
        int64_t sp = 0;
        stack[sp++] = s0;
        edges[s0] = 1;
        while (sp) {
            int64_t p = stack[--sp];
            int y = (int)(p / S), x = (int)(p % S);
            for (int dy = -1; dy <= 1; dy++)
                for (int dx = -1; dx <= 1; dx++) {
                    int yy = y + dy, xx = x + dx;
                    if (yy < 0 || yy >= S || xx < 0 || xx >= S) continue;
                    int64_t q = (int64_t)yy * S + xx;
                    if (cls[q] && !edges[q]) { edges[q] = 1; stack[sp++] = q; }
                }
        }
    }
    if (o_smoothed) memcpy(o_smoothed, sm, sizeof(double) * n);
    if (o_isobel) memcpy(o_isobel, is, sizeof(double) * n);
    if (o_jsobel) memcpy(o_jsobel, js, sizeof(double) * n);
    if (o_mag) memcpy(o_mag, mag, sizeof(double) * n);
    if (o_cls) memcpy(o_cls, cls, n);
    free(stack); free(img); free(ones); free(sm); free(bleed); free(is); free(js); free(mag); free(cls);
}

/* ------------------------------------------------------------------ */
/* a-7 ImageProcessing.verticalLine(edges, 60, 120), ImageProcessing.py:61-83.
 * Fx = conv2d(E, Gx) = (1,2,1)^T . (E[:, j-1] - E[:, j+1]); Fy = (1,2,1) . (E[i+1,:] - E[i-1,:])
 * (zero fill); 60 < atan2(Fx, Fy) in degrees < 120  <=>  Fx > 0 and 3*Fy^2 < Fx^2 (integers,
 * equality impossible); hit written to column j-1, column 0 wraps to S-1 (:78). */
SO_API void so_vertical_line(const uint8_t* E, int S, uint8_t* vert)
{
    memset(vert, 0, (size_t)S * S);
#define EE(y, x) (((y) < 0 || (y) >= S || (x) < 0 || (x) >= S) ? 0 : (int)E[(int64_t)(y) * S + (x)])
    for (int i = 0; i < S; i++)
        for (int j = 0; j < S; j++) {
            int fx = (EE(i - 1, j - 1) - EE(i - 1, j + 1)) + 2 * (EE(i, j - 1) - EE(i, j + 1)) +
                     (EE(i + 1, j - 1) - EE(i + 1, j + 1));
            int fy = (EE(i + 1, j - 1) - EE(i - 1, j - 1)) + 2 * (EE(i + 1, j) - EE(i - 1, j)) +
                     (EE(i + 1, j + 1) - EE(i - 1, j + 1));
            if (fx > 0 && 3 * fy * fy < fx * fx) {
                int jj = j - 1;
                if (jj < 0) jj += S;
                vert[(int64_t)i * S + jj] = 1;
            }
        }
#undef EE
}

/* a-8 ImageProcessing.block(vert, c), ImageProcessing.py:100-195 */
SO_API void so_block(const uint8_t* vert, int S, int c, int* t_out, int* end_out)
{
    int count = 0, MAX = 0, END = 0, J = 0, buffer = 0;
    int c0 = c - 1 < 0 ? 0 : c - 1, c1 = c + 2 > S ? S : c + 2;
    for (int i = 0; i < S; i++) {
        int v = 0;
        for (int x = c0; x < c1; x++) v |= vert[(int64_t)i * S + x];
        if (v == 1) { count++; J = i; }
        else if (buffer < 5) buffer++;
        else {
            if (count > MAX) { MAX = count; END = J; }
            count = 0; buffer = 0;
        }
    }
    if (count > MAX) { MAX = count; END = J; }
    int t = MAX;
    if (END < c) END = END - t + 1;
    *t_out = t; *end_out = END;
}

/* a-8 caller loop getStripe.py:924-940 -> per column keep flag, END, updown */
SO_API void so_columns(const uint8_t* vert, int S, int minH, int32_t* t_arr, int32_t* end_arr,
                       int32_t* ud_arr /* 0 = not kept, 1 up, 2 down */)
{
    for (int c = 0; c < S; c++) {
        int t, END;
        so_block(vert, S, c, &t, &END);
        t_arr[c] = t; end_arr[c] = END; ud_arr[c] = 0;
        int above = c < END ? c : END, bottom = c > END ? c : END;
        int sum = 0;
        for (int y = above; y <= bottom; y++) {
            int yy = y < 0 ? y + S : y;          /* numpy negative index wrap (never hit: END >= 0) */
            if (yy >= 0 && yy < S) sum += vert[(int64_t)yy * S + c];
        }
        if (t > minH && sum != 0) ud_arr[c] = (END > c) ? 2 : 1;
    }
}

/* a-9 line joining for one `ud`, getStripe.py:946-1078.  testmat (S x S, u8) is an output
 * for debugging.  Returns the number of pairs appended to (px, pw, py, ph). */
static int so_join_ud(const uint8_t* edges, const uint8_t* vert, int S, int ud, const int32_t* end_arr,
                      const int32_t* ud_arr, int maxW, uint8_t* testmat, int32_t* ox, int32_t* oy,
                      int32_t* ow, int32_t* oh, int cap)
{
    memset(testmat, 0, (size_t)S * S);
    for (int c = 0; c < S; c++) {                       /* :948-955 */
        if (ud_arr[c] != ud) continue;
        int st = c, en = end_arr[c];
        if (ud == 1) { int tmp = st; st = en; en = tmp; }
        /* python slice st:en with possibly negative st (never: END>=0) */
        if (st < 0) st = 0;
        if (en > S) en = S;
        for (int y = st; y < en; y++) testmat[(int64_t)y * S + c] = 1;
    }
    int* rs = (int*)malloc(sizeof(int) * (S + 2));
    int* re = (int*)malloc(sizeof(int) * (S + 2));
    for (int r = 0; r < S; r++) {                       /* :957-978 */
        uint8_t* vec = testmat + (int64_t)r * S;
        int nl = 0, ne = 0;
        if (vec[0] == 1) rs[nl++] = 0;
        for (int i = 0; i + 1 < S; i++) {
            if (vec[i + 1] > vec[i]) rs[nl++] = i + 1;
            if (vec[i + 1] < vec[i]) re[ne++] = i;
        }
        if (vec[S - 1] == 1) re[ne++] = S - 1;
        for (int L = 0; L < nl; L++) {
            int st = rs[L], en = re[L], sum = 0;
            for (int x = st; x <= en; x++) sum += edges[(int64_t)r * S + x];
            if (sum > 0) {
                for (int x = st; x < en; x++) vec[x] = vert[(int64_t)r * S + x];
            } else {
                /* MED = int(np.round(median([st+en]) / 2)) : round-half-even */
                int med = (int)nearbyint((double)(st + en) / 2.0);
                for (int x = st; x < en; x++) vec[x] = 0;
                vec[med] = 1;
            }
        }
    }
    /* column counts, keep >= 3 (:981-994) */
    int* cidx = (int*)malloc(sizeof(int) * (S + 1));
    int* clen = (int*)malloc(sizeof(int) * (S + 1));
    int nrow = 0;
    for (int c = 0; c < S; c++) {
        int cnt = 0;
        for (int y = 0; y < S; y++) cnt += (testmat[(int64_t)y * S + c] == 1);
        if (cnt >= 3) { cidx[nrow] = c; clen[nrow] = cnt; nrow++; }
    }
    /* grouping -> meanX (:996-1028), including the stale-[Current] quirk */
    double* meanX = (double*)malloc(sizeof(double) * (S + 2));
    int nmean = 0;
    int* cont = (int*)malloc(sizeof(int) * (S + 2));
    int* len = (int*)malloc(sizeof(int) * (S + 2));
    int ncont = 0, isContinue = 0;
#define FLUSH() do { \
        long ssum = 0; for (int q = 0; q < ncont; q++) ssum += len[q]; \
        double temp = 0.0; \
        for (int q = 0; q < ncont; q++) temp = temp + (double)cont[q] * ((double)len[q] / (double)ssum); \
        meanX[nmean++] = nearbyint(temp); } while (0)
    for (int c = 0; c + 1 < nrow; c++) {
        int Current = cidx[c], Next = cidx[c + 1];
        if (Next - Current == 1 && isContinue) {
            cont[ncont] = Next; len[ncont] = clen[c + 1]; ncont++;
        } else if (Next - Current == 1 && !isContinue) {
            cont[0] = Current; cont[1] = Next; len[0] = clen[c]; len[1] = clen[c + 1]; ncont = 2;
            isContinue = 1;
        } else if (Next - Current != 1 && !isContinue) {
            cont[0] = Current; len[0] = clen[c]; ncont = 1;
            isContinue = 0;
            FLUSH();
            /* Len is now the normalised [1.0]; a later flush re-normalises 1.0/1.0: same value */
        } else {
            FLUSH();
            cont[0] = Current; len[0] = clen[c]; ncont = 1;
            isContinue = 0;
        }
    }
    if (ncont == 0) meanX[nmean++] = 0.0;   /* sum([]) = 0 -> np.round(0) */
    else FLUSH();
#undef FLUSH
    /* X = sorted(set(meanX)) */
    for (int i = 1; i < nmean; i++) {
        double v = meanX[i]; int j = i - 1;
        while (j >= 0 && meanX[j] > v) { meanX[j + 1] = meanX[j]; j--; }
        meanX[j + 1] = v;
    }
    int nx = 0;
    for (int i = 0; i < nmean; i++) if (nx == 0 || meanX[i] != meanX[nx - 1]) meanX[nx++] = meanX[i];
    int np_ = 0;
    for (int c = 0; c + 1 < nx; c++) {                   /* :1034-1078 */
        int n = (int)meanX[c], m = (int)meanX[c + 1];
        int gap = abs(m - n);
        if (gap > 1 && gap <= maxW) {
            int p1 = (gap > 4) ? m - 2 : m;
            int MIN = S, MAX = -1;
            int lo = n - 1 < 0 ? 0 : n - 1, hi = n + 2 > S ? S : n + 2;
            for (int y = 0; y < S; y++) for (int x = lo; x < hi; x++)
                if (testmat[(int64_t)y * S + x] == 1) { if (y < MIN) MIN = y; if (y > MAX) MAX = y; }
            lo = m - 1 < 0 ? 0 : m - 1; hi = m + 2 > S ? S : m + 2;
            for (int y = 0; y < S; y++) for (int x = lo; x < hi; x++)
                if (testmat[(int64_t)y * S + x] == 1) { if (y < MIN) MIN = y; if (y > MAX) MAX = y; }
            if (ud == 1) MAX = p1; else MIN = n;
            if (np_ < cap) {
                ox[np_] = n; oy[np_] = MIN; ow[np_] = p1 - n + 1; oh[np_] = MAX - MIN + 1;
            }
            np_++;
        }
    }
    free(rs); free(re); free(cidx); free(clen); free(meanX); free(cont); free(len);
    return np_;
}

/* Full StripeSearch body for one compacted frame (getStripe.py:864-1104, without the
 * per-frame RemoveRedundant and without medpixel).  D: S x S float64, NaN already 0.
 * bvals = np.arange(0.5, 1.01, 0.1) from the caller.  Returns number of records (may exceed
 * cap; only the first cap are written). */
SO_API int so_stripe_search(const double* D, int S, double M, const double* bvals, int nb, int bf,
                            const double* gw, int gr, int minH, int maxW, so_rec* out, int cap)
{
    int64_t n = (int64_t)S * S;
    double* g = (double*)malloc(sizeof(double) * n);
    float* gray = (float*)malloc(sizeof(float) * n);
    uint8_t* edges = (uint8_t*)malloc(n);
    uint8_t* vert = (uint8_t*)malloc(n);
    uint8_t* testmat = (uint8_t*)malloc(n);
    int32_t* t_arr = (int32_t*)malloc(sizeof(int32_t) * S);
    int32_t* end_arr = (int32_t*)malloc(sizeof(int32_t) * S);
    int32_t* ud_arr = (int32_t*)malloc(sizeof(int32_t) * S);
    int32_t ox[512], oy[512], ow[512], oh[512];
    int nrec = 0;
    so_gplane(D, n, M, g);
    for (int bi = 0; bi < nb; bi++) {
        so_gray(g, S, bvals[bi], bf, gray);
        so_canny(gray, S, gw, gr, edges, 0, 0, 0, 0, 0);
        so_vertical_line(edges, S, vert);
        so_columns(vert, S, minH, t_arr, end_arr, ud_arr);
        for (int ud = 1; ud <= 2; ud++) {
            int np_ = so_join_ud(edges, vert, S, ud, end_arr, ud_arr, maxW, testmat, ox, oy, ow, oh, 512);
            if (np_ > 512) np_ = 512;
            for (int k = 0; k < np_; k++) {
                if (nrec < cap) {
                    so_rec* r = &out[nrec];
                    r->b_index = bi; r->ud = ud; r->x = ox[k]; r->y = oy[k]; r->w = ow[k]; r->h = oh[k];
                    /* numpy sum of the 2-D slice: per row sequential (w < 8), rows added in order */
                    double tot = 0.0;
                    for (int y = oy[k]; y < oy[k] + oh[k] && y < S; y++) {
                        double rsum = 0.0;
                        for (int x = ox[k]; x < ox[k] + ow[k] && x < S; x++) rsum += D[(int64_t)y * S + x];
                        tot += rsum;
                    }
                    r->total = tot;
                }
                nrec++;
            }
        }
    }
    free(g); free(gray); free(edges); free(vert); free(testmat); free(t_arr); free(end_arr); free(ud_arr);
    return nrec;
}

/* debug: testmat for one ud given edges / vert */
SO_API int so_join_dbg(const uint8_t* edges, const uint8_t* vert, int S, int ud, int minH, int maxW,
                       uint8_t* testmat, int32_t* ox, int32_t* oy, int32_t* ow, int32_t* oh, int cap)
{
    int32_t* t_arr = (int32_t*)malloc(sizeof(int32_t) * S);
    int32_t* end_arr = (int32_t*)malloc(sizeof(int32_t) * S);
    int32_t* ud_arr = (int32_t*)malloc(sizeof(int32_t) * S);
    so_columns(vert, S, minH, t_arr, end_arr, ud_arr);
    int n = so_join_ud(edges, vert, S, ud, end_arr, ud_arr, maxW, testmat, ox, oy, ow, oh, cap);
    free(t_arr); free(end_arr); free(ud_arr);
    return n;
}
