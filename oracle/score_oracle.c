/*
 * score_oracle.c -- CPU restatement of the background-window arithmetic of getStripe.nulldist
 * (TEST INFRASTRUCTURE ONLY, see stripe_oracle.c).  The window means are np.mean over 2-D
 * slices; numpy 1.26/2.2 reduce such a (small, strided) slice with ONE pairwise-sum inner loop
 * (numpy/core/src/umath/loops_utils.h.src, @TYPE@_pairwise_sum) over the buffered, row-major
 * flattened block, then one division by the element count.  Restated here and pinned against the
 * reference's own tables (tests/golden/e2e_*.npz).
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>

#define SO_API __attribute__((visibility("default")))

/* numpy DOUBLE_pairwise_sum for a contiguous run (PW_BLOCKSIZE 128) */
SO_API double so_pairwise(const double* a, int64_t n)
{
    if (n < 8) {
        double res = 0.;
        for (int64_t i = 0; i < n; i++) res += a[i];
        return res;
    } else if (n <= 128) {
        double r[8], res;
        int64_t i;
        for (i = 0; i < 8; i++) r[i] = a[i];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; k++) r[k] += a[i + k];
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        int64_t n2 = n / 2;
        n2 -= n2 % 8;
        return so_pairwise(a, n2) + so_pairwise(a + n2, n - n2);
    }
}

/* python slice a:b on an axis of length n -> [lo, hi) */
static void py_slice(int64_t a, int64_t b, int64_t n, int64_t* lo, int64_t* hi)
{
    if (a < 0) { a += n; if (a < 0) a = 0; }
    if (a > n) a = n;
    if (b < 0) { b += n; if (b < 0) b = 0; }
    if (b > n) b = n;
    if (b < a) b = a;
    *lo = a; *hi = b;
}

/* np.mean(mat[r0:r1, c0:c1]) with NaN already replaced by 0 in mat.  numpy's buffered reduction
 * copies a strided 2-D slice of <= 8192 elements (its default buffer size) into one contiguous
 * buffer and runs ONE pairwise inner loop over it, i.e. pairwise over the row-major flattened
 * block (verified against numpy 1.26.4 and 2.2.6 for 10x10 and 50x50 blocks). */
static double so_pw_block(const double* base, int64_t ncol, int64_t w, int64_t o, int64_t n)
{
#define BEL(k) base[((k) / w) * ncol + ((k) % w)]
    if (n < 8) {
        double res = 0.;
        for (int64_t i = 0; i < n; i++) res += BEL(o + i);
        return res;
    } else if (n <= 128) {
        double r[8], res;
        int64_t i;
        for (i = 0; i < 8; i++) r[i] = BEL(o + i);
        for (i = 8; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; k++) r[k] += BEL(o + i + k);
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += BEL(o + i);
        return res;
    } else {
        int64_t n2 = n / 2;
        n2 -= n2 % 8;
        return so_pw_block(base, ncol, w, o, n2) + so_pw_block(base, ncol, w, o + n2, n - n2);
    }
#undef BEL
}

SO_API double so_block_mean(const double* mat, int64_t nrow, int64_t ncol, int64_t r0, int64_t r1, int64_t c0,
                            int64_t c1)
{
    int64_t rl, rh, cl, ch;
    py_slice(r0, r1, nrow, &rl, &rh);
    py_slice(c0, c1, ncol, &cl, &ch);
    int64_t cnt = (rh - rl) * (ch - cl);
    double acc = 0.0;
    /* blocks of more than 8192 elements (bs > 90, i.e. bins below 556 bp): the buffered iterator hands the inner
     * loop one buffer (8192 elements of the row-major flattened block) at a time; each buffer is pairwise-summed
     * and the partial sums are added in order (checked against numpy 1.26.4 and 2.2.6 for blocks up to 200 x 200,
     * tests/test_oracle_golden.py::test_block_mean_is_numpys_mean) */
    for (int64_t o = 0; o < cnt; o += 8192) {
        double part = so_pw_block(mat + rl * ncol + cl, ncol, ch - cl, o, cnt - o < 8192 ? cnt - o : 8192);
        acc = (o == 0) ? part : acc + part;
    }
    return acc / (double)cnt;   /* 0/0 -> NaN like np.mean of an empty slice */
}

/* getStripe.py:347-378 (and :389-414, :447-477): for sampled rows xs[0..n) of `mat` and offsets
 * j = 0..399, the six window means and the four "centre minus flank" tables.
 * out: 4 tables [400][n] in the order left_up, right_up, left_down, right_down. */
SO_API void so_null_windows(const double* mat, int64_t nrow, int64_t ncol, const int64_t* xs, int64_t n, int bs,
                            int yoff, double* lu, double* ru, double* ld, double* rd)
{
    int up = bs / 2, down = bs - up;   /* floor(bs/2), bs - up (getStripe.py:288-291) */
    for (int64_t i = 0; i < n; i++) {
        int64_t x = xs[i];
        for (int j = 0; j < 400; j++) {
            int64_t yd = x + j + yoff, yu = x - j + yoff;
            double l_u = so_block_mean(mat, nrow, ncol, x - up - bs, x - up, yu - up, yu + down);
            double c_u = so_block_mean(mat, nrow, ncol, x - up, x + down, yu - up, yu + down);
            double r_u = so_block_mean(mat, nrow, ncol, x + down, x + down + bs, yu - up, yu + down);
            double l_d = so_block_mean(mat, nrow, ncol, x - up - bs, x - up, yd - up, yd + down);
            double c_d = so_block_mean(mat, nrow, ncol, x - up, x + down, yd - up, yd + down);
            double r_d = so_block_mean(mat, nrow, ncol, x + down, x + down + bs, yd - up, yd + down);
            lu[(int64_t)j * n + i] = c_u - l_u;
            ru[(int64_t)j * n + i] = c_u - r_u;
            ld[(int64_t)j * n + i] = c_d - l_d;
            rd[(int64_t)j * n + i] = c_d - r_d;
        }
    }
}
