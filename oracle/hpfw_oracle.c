/*
 * hpfw_oracle.c -- CPU restatement of the hpfw index()/search() hot path.  See hpfw_oracle.h:
 * TEST INFRASTRUCTURE ONLY, PARITY UNPINNED (no reference fixtures exist, reference unbuildable
 * here).  Build: see oracle/Makefile (-O3 -mfma -ffp-contract=off, no fast-math: every rounding
 * below is explicit; fmaf() is a single-rounding fused multiply-add).
 *
 * Reference map (all paths relative to /root/reference):
 *   include/hpfw/spectrum/cqt.h:36-84          spectrogram(): MonoLoader + NSGConstantQ + |.| + /3
 *   include/hpfw/spectrum/convert.h:7-25       power_to_db / amplitude_to_db
 *   include/hpfw/core/hashprint_handle.h:79-142 calc_frames, calc_fingerprint, bool_col_to_num
 *   include/hpfw/core/parallel_collector.h:54-59 calc_hashprint (filters * frames)
 *   include/hpfw/audioproblems/live-song-id/storage.h:27-64 MemoryStorage::find
 *   examples/python/liveid.ipynb cell 9        top-10 by (distance, label)
 * NSGConstantQ itself lives in essentia (un-vendored, version unpinned; CMakeLists.txt:36);
 * its published algorithm (Holighaus, Doerfler, Velasco, Grill: "A framework for invertible,
 * real-time constant-Q transforms", IEEE TASLP 2013; essentia nsgconstantq.cpp) is restated in
 * make_bands() / hpfw_oracle_cqmag() and in DESIGN.md appendix A.
 */
#include "hpfw_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    float r, i;
} cf;

/* ------------------------------------------------------------------------------------------ */
/* complex helpers with a fixed rounding order (DESIGN.md "Arithmetic specification" S1)       */
/* ------------------------------------------------------------------------------------------ */
static inline cf c_add(cf a, cf b) { cf o = {a.r + b.r, a.i + b.i}; return o; }
static inline cf c_sub(cf a, cf b) { cf o = {a.r - b.r, a.i - b.i}; return o; }
/* a * w */
static inline cf c_mul(cf a, cf w)
{
    float p = a.i * w.i;
    float q = a.i * w.r;
    cf o = {fmaf(a.r, w.r, -p), fmaf(a.r, w.i, q)};
    return o;
}
/* a * conj(w) */
static inline cf c_mulc(cf a, cf w)
{
    float p = a.i * w.i;
    float q = a.r * w.i;
    cf o = {fmaf(a.r, w.r, p), fmaf(a.i, w.r, -q)};
    return o;
}
/* -i * a */
static inline cf c_mulmi(cf a) { cf o = {a.i, -a.r}; return o; }
/* real scalar times complex, then fused add: s * a + b */
static inline cf c_fma_s(float s, cf a, cf b) { cf o = {fmaf(s, a.r, b.r), fmaf(s, a.i, b.i)}; return o; }
static inline cf c_scale(float s, cf a) { cf o = {s * a.r, s * a.i}; return o; }

/* ------------------------------------------------------------------------------------------ */
/* S2: twiddles.  e^{-2 pi i m/n} evaluated in double with exact octant symmetry, then rounded  */
/* ------------------------------------------------------------------------------------------ */
static void twiddle_d(int64_t m, int64_t n, double *re, double *im)
{
    m %= n;
    if (m < 0) m += n;
    int64_t a = 8 * m;
    int oct = (int)(a / n);
    int64_t r = a - (int64_t)oct * n;
    int64_t t = (oct & 1) ? (n - r) : r;
    double alpha = M_PI * (double)t / (double)(4 * n);
    double ca, sa, c, s;
    sincos(alpha, &sa, &ca); /* glibc's sincos here and in plan.cpp: cos() and sin() differ from it in the last bit for 0.1 % of arguments */
    switch (oct) {
    case 0: c = ca; s = sa; break;
    case 1: c = sa; s = ca; break;
    case 2: c = -sa; s = ca; break;
    case 3: c = -ca; s = sa; break;
    case 4: c = -ca; s = -sa; break;
    case 5: c = -sa; s = -ca; break;
    case 6: c = sa; s = -ca; break;
    default: c = ca; s = -sa; break;
    }
    *re = c;
    *im = -s;
}

void hpfw_oracle_twiddle(int64_t m, int64_t n, float *re, float *im)
{
    double c, s;
    twiddle_d(m, n, &c, &s);
    *re = (float)c;
    *im = (float)s;
}

static cf *make_twiddle_table(int64_t n)
{
    cf *t = (cf *)malloc(sizeof(cf) * (size_t)n);
    for (int64_t m = 0; m < n; ++m) hpfw_oracle_twiddle(m, n, &t[m].r, &t[m].i);
    return t;
}

/* ------------------------------------------------------------------------------------------ */
/* S3: small forward DFT codelets, radix 2, 3, 4, 5, 7 (sign -).  In place on u[0..r).          */
/* ------------------------------------------------------------------------------------------ */
#define K3_S 0.86602540378443864676f /* sin(2pi/3) */
#define K5_C1 0.30901699437494742410f
#define K5_C2 (-0.80901699437494742410f)
#define K5_S1 0.95105651629515357212f
#define K5_S2 0.58778525229247312917f
#define K7_C1 0.62348980185873353053f
#define K7_C2 (-0.22252093395631440429f)
#define K7_C3 (-0.90096886790241912624f)
#define K7_S1 0.78183148246802980871f
#define K7_S2 0.97492791218182360702f
#define K7_S3 0.43388373911755812048f

static inline void dft2(cf *u)
{
    cf a = u[0], b = u[1];
    u[0] = c_add(a, b);
    u[1] = c_sub(a, b);
}

static inline void dft3(cf *u)
{
    cf t1 = c_add(u[1], u[2]);
    cf d = c_sub(u[1], u[2]);
    cf m1 = c_fma_s(-0.5f, t1, u[0]);
    cf jd = {K3_S * d.i, -(K3_S * d.r)}; /* -i * s * d */
    u[0] = c_add(u[0], t1);
    u[1] = c_add(m1, jd);
    u[2] = c_sub(m1, jd);
}

static inline void dft4(cf *u)
{
    cf t0 = c_add(u[0], u[2]);
    cf t1 = c_sub(u[0], u[2]);
    cf t2 = c_add(u[1], u[3]);
    cf t3 = c_mulmi(c_sub(u[1], u[3]));
    u[0] = c_add(t0, t2);
    u[2] = c_sub(t0, t2);
    u[1] = c_add(t1, t3);
    u[3] = c_sub(t1, t3);
}

static inline void dft5(cf *u)
{
    cf a1 = c_add(u[1], u[4]), a2 = c_add(u[2], u[3]);
    cf b1 = c_sub(u[1], u[4]), b2 = c_sub(u[2], u[3]);
    cf p1 = c_fma_s(K5_C2, a2, c_fma_s(K5_C1, a1, u[0]));
    cf p2 = c_fma_s(K5_C1, a2, c_fma_s(K5_C2, a1, u[0]));
    cf q1 = c_fma_s(K5_S2, b2, c_scale(K5_S1, b1));
    cf q2 = c_fma_s(-K5_S1, b2, c_scale(K5_S2, b1));
    cf jq1 = c_mulmi(q1), jq2 = c_mulmi(q2); /* -i q */
    u[0] = c_add(c_add(u[0], a1), a2);
    u[1] = c_add(p1, jq1);
    u[4] = c_sub(p1, jq1);
    u[2] = c_add(p2, jq2);
    u[3] = c_sub(p2, jq2);
}

static inline void dft7(cf *u)
{
    cf a1 = c_add(u[1], u[6]), a2 = c_add(u[2], u[5]), a3 = c_add(u[3], u[4]);
    cf b1 = c_sub(u[1], u[6]), b2 = c_sub(u[2], u[5]), b3 = c_sub(u[3], u[4]);
    /* s = 1: cos(1,2,3) sin(1,2,3); s = 2: cos(2,3,1) sin(2,-3,-1); s = 3: cos(3,1,2) sin(3,-1,2) */
    cf p1 = c_fma_s(K7_C3, a3, c_fma_s(K7_C2, a2, c_fma_s(K7_C1, a1, u[0])));
    cf p2 = c_fma_s(K7_C1, a3, c_fma_s(K7_C3, a2, c_fma_s(K7_C2, a1, u[0])));
    cf p3 = c_fma_s(K7_C2, a3, c_fma_s(K7_C1, a2, c_fma_s(K7_C3, a1, u[0])));
    cf q1 = c_fma_s(K7_S3, b3, c_fma_s(K7_S2, b2, c_scale(K7_S1, b1)));
    cf q2 = c_fma_s(-K7_S1, b3, c_fma_s(-K7_S3, b2, c_scale(K7_S2, b1)));
    cf q3 = c_fma_s(K7_S2, b3, c_fma_s(-K7_S1, b2, c_scale(K7_S3, b1)));
    cf jq1 = c_mulmi(q1), jq2 = c_mulmi(q2), jq3 = c_mulmi(q3);
    u[0] = c_add(c_add(c_add(u[0], a1), a2), a3);
    u[1] = c_add(p1, jq1);
    u[6] = c_sub(p1, jq1);
    u[2] = c_add(p2, jq2);
    u[5] = c_sub(p2, jq2);
    u[3] = c_add(p3, jq3);
    u[4] = c_sub(p3, jq3);
}

static inline void dft_r(cf *u, int r)
{
    switch (r) {
    case 2: dft2(u); break;
    case 3: dft3(u); break;
    case 4: dft4(u); break;
    case 5: dft5(u); break;
    default: dft7(u); break;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* S4: in-place decimation-in-frequency FFT (natural in, digit-reversed out) and the inverse    */
/*     decimation-in-time FFT (digit-reversed in, natural out, unnormalised, sign +).           */
/* ------------------------------------------------------------------------------------------ */
static void fft_dif(cf *a, int64_t n, const int32_t *radix, int nr, const cf *tw)
{
    int64_t len = n;
    for (int p = 0; p < nr; ++p) {
        int r = radix[p];
        int64_t m = len / r, ts = n / len;
        for (int64_t base = 0; base < n; base += len) {
            for (int64_t j = 0; j < m; ++j) {
                cf u[7];
                for (int q = 0; q < r; ++q) u[q] = a[base + j + q * m];
                dft_r(u, r);
                a[base + j] = u[0];
                for (int s = 1; s < r; ++s) a[base + j + s * m] = c_mul(u[s], tw[ts * j * s]);
            }
        }
        len = m;
    }
}

static void fft_idit(cf *a, int64_t n, const int32_t *radix, int nr, const cf *tw)
{
    int64_t m = 1;
    for (int p = nr - 1; p >= 0; --p) {
        int r = radix[p];
        int64_t len = m * r, ts = n / len;
        for (int64_t base = 0; base < n; base += len) {
            for (int64_t j = 0; j < m; ++j) {
                cf u[7];
                u[0] = a[base + j];
                for (int q = 1; q < r; ++q) u[q] = c_mulc(a[base + j + q * m], tw[ts * j * q]);
                /* inverse codelet = swap(forward codelet(swap(.))) */
                for (int q = 0; q < r; ++q) { float t = u[q].r; u[q].r = u[q].i; u[q].i = t; }
                dft_r(u, r);
                for (int s = 0; s < r; ++s) { cf o = {u[s].i, u[s].r}; a[base + j + s * m] = o; }
            }
        }
        m = len;
    }
}

int64_t hpfw_oracle_digit_pos(int64_t k, int64_t n, const int32_t *radix, int nr)
{
    int64_t pos = 0, len = n;
    for (int p = 0; p < nr; ++p) {
        int r = radix[p];
        len /= r;
        pos += (k % r) * len;
        k /= r;
    }
    return pos;
}

void hpfw_oracle_fft_dif(float *a, int64_t n, const int32_t *radix, int nr)
{
    cf *tw = make_twiddle_table(n);
    fft_dif((cf *)a, n, radix, nr, tw);
    free(tw);
}

void hpfw_oracle_fft_idit(float *a, int64_t n, const int32_t *radix, int nr)
{
    cf *tw = make_twiddle_table(n);
    fft_idit((cf *)a, n, radix, nr, tw);
    free(tw);
}

/* radix list of a 7-smooth n: primes descending, pairs of 2 merged into 4 ([.. 4 .. 3 .. 2]).
 * Returns the number of passes or -1. */
static int make_radix_list(int64_t n, int32_t *radix)
{
    int c2 = 0, c3 = 0, c5 = 0, c7 = 0, nr = 0;
    while (n % 7 == 0) { n /= 7; ++c7; }
    while (n % 5 == 0) { n /= 5; ++c5; }
    while (n % 3 == 0) { n /= 3; ++c3; }
    while (n % 2 == 0) { n /= 2; ++c2; }
    if (n != 1) return -1;
    if (c7 + c5 + c2 / 2 + c3 + (c2 & 1) > HPFW_O_MAXRADIX) return -1;
    for (int i = 0; i < c7; ++i) radix[nr++] = 7;
    for (int i = 0; i < c5; ++i) radix[nr++] = 5;
    for (int i = 0; i < c2 / 2; ++i) radix[nr++] = 4;
    for (int i = 0; i < c3; ++i) radix[nr++] = 3;
    if (c2 & 1) radix[nr++] = 2;
    return nr;
}

/* pass order of the length-n2 row transform: the descending list taken alternately from its front
 * and its back ([7,7,5,5,4,3] -> [7,3,7,4,5,5]), so that neighbouring passes have small products */
static int make_rows_radix_list(int64_t n, int32_t *radix)
{
    int32_t d[HPFW_O_MAXRADIX];
    int nr = make_radix_list(n, d);
    if (nr < 0) return nr;
    int lo = 0, hi = nr - 1, w = 0;
    while (lo <= hi) {
        radix[w++] = d[lo++];
        if (lo <= hi) radix[w++] = d[hi--];
    }
    return nr;
}

/* ------------------------------------------------------------------------------------------ */
/* plan                                                                                        */
/* ------------------------------------------------------------------------------------------ */
#define N2_MAX 6826 /* complex f32 elements of one in-core FFT: with half of its twiddle table, two fit a CU's LDS */
#define SAMPLE_RATE 44100.0
#define MIN_FREQ 130.81  /* cqt.h:60 */
#define MAX_FREQ 4186.01 /* cqt.h:61 */
#define BINS_PER_OCTAVE 24
#define MIN_WINDOW 96 /* cqt.h:58: "minimumWindow", int(HopLength) */
#define DOWNSAMPLE 3  /* cqt.h:22 */

typedef struct {
    int64_t p;        /* transform length (power of two)                         */
    int32_t nr;
    int32_t radix[HPFW_O_MAXRADIX];
    cf *tw;           /* T_p                                                      */
    cf *vrev;         /* DFT_p(chirp), stored at digit-reversed positions         */
} bluestein_class;

struct hpfw_oracle_plan {
    hpfw_oracle_plan_info info;
    int32_t start[HPFW_O_BINS], lg[HPFW_O_BINS], psize[HPFW_O_BINS], cls[HPFW_O_BINS];
    cf *tw_n2;       /* T_{n2}                                    */
    cf *tw_n1;       /* T_{n1}                                    */
    int32_t *wq;     /* [n1][2]: rint(2^22 T_{n1}[m]) -- the column stage's fixed-point twiddles (S6) */
    cf *ts;          /* [n1 / 2 + 1][n2]: the twiddles between the stages, T_N[q1 k2] 2^-37 (the scale of wq and of pcm / 32768
                      * folded in: a power of two, exact) FORMED AS THE ROW STAGE FORMS THEM, from every fourth one:
                      * ts[q1][4 m] = T_N[4 q1 m] 2^-37, ts[q1][4 m + e] = ts[q1][4 m] * T_N[q1 e] (S1), e = 1, 2, 3     */
    int32_t *pos_n2; /* digit-reversed position of output k2       */
    cf *g[HPFW_O_BINS]; /* window * chirp / (M * P), length lg[j]  */
    int n_cls;
    bluestein_class bc[8];
    /* S15: forward DFT of a clip whose length has a prime factor above 7, as a chirp-z convolution */
    unsigned conv;   /* HPFW_O_CONV_*: essentia conventions that cannot be checked offline (hpfw_oracle.h) */
    int bluestein;
    int64_t bz_l;    /* L = n1 * n2 >= N + (kmax - kmin) - 1, n2 = 6300 */
    cf *bz_w;        /* [n1][n2]: w[r + n1 t] = e^{-i pi n^2 / N}, 0 from n = N on */
    cf *bz_tl;       /* [n1][n2]: T_L[r k2]                                          */
    cf *bz_bhat;     /* [n1][n2]: DFT_L(conj chirp)[n2 k1 + k2]                      */
    cf *bz_wk;       /* [kmax - kmin]: w[k] / L                                      */
};

/* essentia NSGConstantQ::designWindow as configured at cqt.h:54-61: band centres
 * f_j = fmin 2^(j/24), bandwidth Q f_j (gamma = 0), positions floor(f_j / fftres), window
 * lengths max(round(bw_j / fftres), minimumWindow); rasterize "full": M = max_j Lg_j over the
 * constant-Q bands (DC / Nyquist bands are computed by essentia but discarded at cqt.h:64-69). */
static int make_bands(hpfw_oracle_plan *p)
{
    int64_t n = p->info.n_samples;
    double fftres = SAMPLE_RATE / (double)n;
    double q = pow(2.0, 1.0 / BINS_PER_OCTAVE) - pow(2.0, -1.0 / BINS_PER_OCTAVE);
    int nb = (int)floor(BINS_PER_OCTAVE * log2(MAX_FREQ / MIN_FREQ)) + 1;
    if (nb != HPFW_O_BINS) return -1;
    int64_t kmin = n, kmax = 0, m = 0;
    for (int j = 0; j < HPFW_O_BINS; ++j) {
        double f = MIN_FREQ * pow(2.0, (double)j / BINS_PER_OCTAVE);
        int64_t posit;
        double bw;
        if (p->conv & HPFW_O_CONV_FLOAT_GEOMETRY) { /* essentia's Real is float */
            float fres = (float)SAMPLE_RATE / (float)n;
            float qf = powf(2.0f, 1.0f / (float)BINS_PER_OCTAVE) - powf(2.0f, -1.0f / (float)BINS_PER_OCTAVE);
            float ff = (float)MIN_FREQ * powf(2.0f, (float)j / (float)BINS_PER_OCTAVE);
            posit = (int64_t)floorf(ff / fres);
            bw = (double)(qf * ff / fres);
        } else {
            posit = (int64_t)floor(f / fftres);
            bw = q * f / fftres;
        }
        int64_t lg = (p->conv & HPFW_O_CONV_LG_HALF_EVEN) ? (int64_t)nearbyint(bw) : (int64_t)round(bw);
        if (lg < MIN_WINDOW) lg = MIN_WINDOW;
        int64_t st = posit - lg / 2;
        p->start[j] = (int32_t)st;
        p->lg[j] = (int32_t)lg;
        if (st < kmin) kmin = st;
        if (st + lg > kmax) kmax = st + lg;
        if (lg > m) m = lg;
    }
    if (kmin < 0 || kmax > n / 2) return -1; /* bands must stay inside the positive half */
    p->info.kmin = kmin;
    p->info.kmax = kmax;
    p->info.m = m;
    p->info.c = (m + DOWNSAMPLE - 1) / DOWNSAMPLE; /* valid columns only */
    p->info.n_frames = p->info.c - (HPFW_O_CTX - 1);
    p->info.n_hp = p->info.n_frames - HPFW_O_LAG;
    if (p->info.n_frames < 0) p->info.n_frames = 0;
    if (p->info.n_hp < 0) p->info.n_hp = 0;
    return 0;
}

/* S5: V_P = DFT_P(v) in double: iterative radix-2 decimation in time, twiddles from twiddle_d,
 * butterfly t = w*b (4 mul, 1 sub, 1 add), a' = a + t, b' = a - t.  No fused operations. */
static void fft_r2_double(double *re, double *im, int64_t n)
{
    for (int64_t i = 1, j = 0; i < n; ++i) {
        int64_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) {
            double t = re[i]; re[i] = re[j]; re[j] = t;
            t = im[i]; im[i] = im[j]; im[j] = t;
        }
    }
    for (int64_t len = 2; len <= n; len <<= 1) {
        int64_t half = len >> 1, ts = n / len;
        for (int64_t j = 0; j < half; ++j) {
            double wr, wi;
            twiddle_d(ts * j, n, &wr, &wi);
            for (int64_t base = 0; base < n; base += len) {
                int64_t ia = base + j, ib = ia + half;
                double tr = wr * re[ib] - wi * im[ib];
                double ti = wr * im[ib] + wi * re[ib];
                re[ib] = re[ia] - tr;
                im[ib] = im[ia] - ti;
                re[ia] = re[ia] + tr;
                im[ia] = im[ia] + ti;
            }
        }
    }
}

/* S5 for lengths 3 m: X[k] = (F0[k mod m] + W^k F1[k mod m]) + W^2k F2[k mod m], F_r the radix-2
 * transform of x[3 t + r], W = e^{-2 pi i / (3 m)} from twiddle_d */
static void dft_double(double *re, double *im, int64_t n)
{
    if (n % 3 != 0) {
        fft_r2_double(re, im, n);
        return;
    }
    int64_t m = n / 3;
    double *fr = (double *)malloc(sizeof(double) * (size_t)n), *fi = (double *)malloc(sizeof(double) * (size_t)n);
    for (int r = 0; r < 3; ++r) {
        for (int64_t t = 0; t < m; ++t) {
            fr[r * m + t] = re[3 * t + r];
            fi[r * m + t] = im[3 * t + r];
        }
        fft_r2_double(fr + r * m, fi + r * m, m);
    }
    for (int64_t k = 0; k < n; ++k) {
        int64_t km = k % m;
        double w1r, w1i, w2r, w2i;
        twiddle_d(k, n, &w1r, &w1i);
        twiddle_d((2 * k) % n, n, &w2r, &w2i);
        double ar = fr[km] + (w1r * fr[m + km] - w1i * fi[m + km]);
        double ai = fi[km] + (w1r * fi[m + km] + w1i * fr[m + km]);
        re[k] = ar + (w2r * fr[2 * m + km] - w2i * fi[2 * m + km]);
        im[k] = ai + (w2r * fi[2 * m + km] + w2i * fr[2 * m + km]);
    }
    free(fr);
    free(fi);
}

/* S7: chirp-z length of a band: the smallest of {2^a, 3 * 2^a}, at least 64, that holds `need` points */
static int64_t chirpz_length(int64_t need)
{
    int64_t p2 = 64, p3 = 96;
    while (p2 < need) p2 <<= 1;
    while (p3 < need) p3 <<= 1;
    return p3 < p2 ? p3 : p2;
}

static void bz_unit(int64_t m, int64_t n, double *re, double *im); /* S2b, below */

/* e^{+i pi 3 m^2 / M} = conj(e^{-2 pi i r / 2M}), r = 3 m^2 mod 2M reduced exactly in integers; S2b (own cosine / sine on
 * the reduced octant) so that the product can generate the constant-Q tables on the device with the same bits
 * (csrc/trig_d.h; until round 4: glibc's sincos of pi r / M) */
static void chirp_d(int64_t m, int64_t big_m, double *c, double *s)
{
    int64_t mm = m < 0 ? -m : m;
    int64_t r = (int64_t)(((__int128)3 * mm * mm) % (2 * big_m));
    double re, im;
    bz_unit(r, 2 * big_m, &re, &im);
    *c = re;
    *s = -im;
}

static int make_bluestein(hpfw_oracle_plan *p)
{
    int64_t big_m = p->info.m, c = p->info.c;
    p->n_cls = 0;
    for (int j = 0; j < HPFW_O_BINS; ++j) {
        int64_t need = p->lg[j] + c - 1, ps = chirpz_length(need);
        p->psize[j] = (int32_t)ps;
        int k;
        for (k = 0; k < p->n_cls; ++k)
            if (p->bc[k].p == ps) break;
        if (k == p->n_cls) {
            if (p->n_cls == 8) return -1;
            bluestein_class *b = &p->bc[p->n_cls++];
            b->p = ps;
            b->nr = make_radix_list(ps, b->radix);
            b->tw = make_twiddle_table(ps);
            double *re = (double *)calloc((size_t)ps, sizeof(double));
            double *im = (double *)calloc((size_t)ps, sizeof(double));
            /* v[m mod P] = e^{-i pi 3 m^2 / M}, m in [-(P - C), C - 1] */
            for (int64_t mm = -(ps - c); mm <= c - 1; ++mm) {
                double cc, ss;
                chirp_d(mm, big_m, &cc, &ss);
                int64_t idx = mm < 0 ? mm + ps : mm;
                re[idx] = cc;
                im[idx] = -ss;
            }
            dft_double(re, im, ps);
            b->vrev = (cf *)malloc(sizeof(cf) * (size_t)ps);
            for (int64_t kk = 0; kk < ps; ++kk) {
                int64_t pos = hpfw_oracle_digit_pos(kk, ps, b->radix, b->nr);
                b->vrev[pos].r = (float)re[kk];
                b->vrev[pos].i = (float)im[kk];
            }
            free(re);
            free(im);
        }
        p->cls[j] = k;
        /* G_j[k] = hann_Lg[k] * e^{+i pi 3 k^2 / M} / (M * P); hann: essentia Windowing "hann",
         * 0.5 - 0.5 cos(2 pi i / (size - 1)), un-normalised, no zero-phase */
        int64_t lg = p->lg[j];
        p->g[j] = (cf *)malloc(sizeof(cf) * (size_t)lg);
        double scale = 1.0 / (((p->conv & HPFW_O_CONV_NO_IFFT_SCALE) ? 1.0 : (double)big_m) * (double)ps);
        int64_t hann_den = (p->conv & HPFW_O_CONV_HANN_PERIODIC) ? lg : lg - 1;
        for (int64_t i = 0; i < lg; ++i) {
            double hc, hs; /* cos(2 pi i / den) by S2b (csrc/trig_d.h cq_window_d) */
            bz_unit(i % hann_den, hann_den, &hc, &hs);
            double w = 0.5 - 0.5 * hc;
            double cc, ss;
            chirp_d(i, big_m, &cc, &ss);
            p->g[j][i].r = (float)(w * cc * scale);
            p->g[j][i].i = (float)(w * ss * scale);
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* S15 tables.  X[k] = w[k] sum_n (x[n] w[n]) conj(w[k - n]), w[n] = e^{-i pi n^2 / N}: a convolution with the   */
/* lags m = k - n in [kmin - (N - 1), kmax - 1], evaluated cyclically at length L >= N + (kmax - kmin) - 1.      */
/* The product generates these tables on the device, one pass of double arithmetic per element and one forward   */
/* transform of the lags in float; this is the same arithmetic written out on the CPU.                           */
/* ------------------------------------------------------------------------------------------ */

/* S2b: cos and sin of x in [0, pi/4] in double: Taylor polynomials in x^2 as explicit fma chains, coefficients the
 * correctly rounded (-1)^j / k!.  IEEE multiply and fma only, so a CPU and a GPU agree bit for bit. */
static void bz_cos_sin(double x, double *c, double *s)
{
    static const double sc[8] = {0x1.952c77030ad4ap-49, -0x1.ae7f3e733b81fp-41, 0x1.6124613a86d09p-33, -0x1.ae64567f544e4p-26,
                                 0x1.71de3a556c734p-19, -0x1.a01a01a01a01ap-13, 0x1.1111111111111p-7,   -0x1.5555555555555p-3};
    static const double cc[9] = {-0x1.6827863b97d97p-53, 0x1.ae7f3e733b81fp-45,  -0x1.93974a8c07c9dp-37,
                                 0x1.1eed8eff8d898p-29,  -0x1.27e4fb7789f5cp-22, 0x1.a01a01a01a01ap-16,
                                 -0x1.6c16c16c16c17p-10, 0x1.5555555555555p-5,   -0.5};
    double z = x * x, ps = sc[0], pc = cc[0];
    for (int i = 1; i < 8; ++i) ps = fma(ps, z, sc[i]);
    for (int i = 1; i < 9; ++i) pc = fma(pc, z, cc[i]);
    *s = fma(x * z, ps, x);
    *c = fma(z, pc, 1.0);
}

/* e^{-2 pi i m / n}, 0 <= m < n: the octant reduction of S2, S2b for the octant's angle */
static void bz_unit(int64_t m, int64_t n, double *re, double *im)
{
    int64_t a = 8 * m;
    int oct = (int)(a / n);
    int64_t r = a - (int64_t)oct * n;
    int64_t t = (oct & 1) ? (n - r) : r;
    double alpha = M_PI * (double)t / (double)(4 * n);
    double ca, sa, c, s;
    bz_cos_sin(alpha, &ca, &sa);
    switch (oct) {
    case 0: c = ca; s = sa; break;
    case 1: c = sa; s = ca; break;
    case 2: c = -sa; s = ca; break;
    case 3: c = -ca; s = sa; break;
    case 4: c = -ca; s = -sa; break;
    case 5: c = -sa; s = -ca; break;
    case 6: c = sa; s = -ca; break;
    default: c = ca; s = -sa; break;
    }
    *re = c;
    *im = -s;
}

/* w[m] = e^{-i pi m^2 / N} = e^{-2 pi i (m^2 mod 2N) / 2N}, m >= 0 */
static void bz_chirp_d(int64_t m, int64_t n, double *re, double *im) { bz_unit((m * m) % (2 * n), 2 * n, re, im); }

/* S15, the first transform's column stage: A[k1][k2] = sum_r T_n1[r k1] y[r][k2] for all n1 = 16 a rows k1, in two
 * stages with r = 16 r2 + r1 and k1 = k_a + a k_b:
 *   Z[r1][k_a] = chain over r2 ascending of T_a[r2 k_a] y[16 r2 + r1]          (T_a[j] = T_n1[16 j])
 *   A[k_a + a k_b] = chain over r1 ascending of T_n1[r1 (k_a + a k_b)] Z[r1][k_a]
 * every chain from 0 with the four fmas of S6 per term (re += dr yr; re += -di yi; im += di yr; im += dr yi).
 * Planar float arrays [n1][n2]; k2 innermost so that the loops vectorise. */
static void bz_columns_two_stage(const hpfw_oracle_plan *p, const float *yr, const float *yi, float *ar, float *ai)
{
    const int64_t n1 = p->info.n1, n2 = p->info.n2, a = n1 / 16;
    float *zr = (float *)calloc((size_t)(n1 * n2), sizeof(float)), *zi = (float *)calloc((size_t)(n1 * n2), sizeof(float));
    for (int64_t r1 = 0; r1 < 16; ++r1)
        for (int64_t ka = 0; ka < a; ++ka) {
            float *qr = zr + (r1 * a + ka) * n2, *qi = zi + (r1 * a + ka) * n2;
            int64_t idx = 0; /* (r2 ka) mod a */
            for (int64_t r2 = 0; r2 < a; ++r2) {
                const float dr = p->tw_n1[16 * idx].r, di = p->tw_n1[16 * idx].i, ndi = -di;
                const float *pr = yr + (16 * r2 + r1) * n2, *pi = yi + (16 * r2 + r1) * n2;
                for (int64_t k2 = 0; k2 < n2; ++k2) {
                    qr[k2] = fmaf(dr, pr[k2], qr[k2]);
                    qr[k2] = fmaf(ndi, pi[k2], qr[k2]);
                    qi[k2] = fmaf(di, pr[k2], qi[k2]);
                    qi[k2] = fmaf(dr, pi[k2], qi[k2]);
                }
                idx += ka;
                if (idx >= a) idx -= a;
            }
        }
    for (int64_t k1 = 0; k1 < n1; ++k1) {
        const int64_t ka = k1 % a;
        float *or_ = ar + k1 * n2, *oi = ai + k1 * n2;
        memset(or_, 0, sizeof(float) * (size_t)n2);
        memset(oi, 0, sizeof(float) * (size_t)n2);
        for (int64_t r1 = 0; r1 < 16; ++r1) {
            const cf d = p->tw_n1[(r1 * k1) % n1];
            const float dr = d.r, di = d.i, ndi = -d.i;
            const float *pr = zr + (r1 * a + ka) * n2, *pi = zi + (r1 * a + ka) * n2;
            for (int64_t k2 = 0; k2 < n2; ++k2) {
                or_[k2] = fmaf(dr, pr[k2], or_[k2]);
                or_[k2] = fmaf(ndi, pi[k2], or_[k2]);
                oi[k2] = fmaf(di, pr[k2], oi[k2]);
                oi[k2] = fmaf(dr, pi[k2], oi[k2]);
            }
        }
    }
    free(zr);
    free(zi);
}

/* S15 steps A + B of the first transform on a sequence of length L given by rows (planar y [n1][n2], flat index
 * j = n2 k1' + k2'): the two-stage column chains over k1' -> q1, times T_L[q1 k2'], then per q1 the row transform over
 * k2' -> q2.  Leaves A[q1 + n1 q2] at out[q1 n2 + q2]. */
static void bz_first_transform(const hpfw_oracle_plan *p, const float *yr, const float *yi, cf *out)
{
    const int64_t n1 = p->info.n1, n2 = p->info.n2, big_l = n1 * n2;
    float *gr = (float *)malloc(sizeof(float) * (size_t)big_l), *gi = (float *)malloc(sizeof(float) * (size_t)big_l);
    bz_columns_two_stage(p, yr, yi, gr, gi);
    cf *z = (cf *)malloc(sizeof(cf) * (size_t)n2);
    for (int64_t q1 = 0; q1 < n1; ++q1) {
        for (int64_t k2 = 0; k2 < n2; ++k2) {
            cf g = {gr[q1 * n2 + k2], gi[q1 * n2 + k2]};
            z[k2] = c_mul(g, p->bz_tl[q1 * n2 + k2]);
        }
        fft_dif(z, n2, p->info.radix, p->info.n_radix, p->tw_n2);
        for (int64_t q2 = 0; q2 < n2; ++q2) out[q1 * n2 + q2] = z[p->pos_n2[q2]];
    }
    free(z);
    free(gr);
    free(gi);
}

static void make_forward_bluestein(hpfw_oracle_plan *p)
{
    const int64_t n = p->info.n_samples, n2 = p->info.n2, big_l = p->bz_l;
    const int64_t nk = p->info.kmax - p->info.kmin;
    p->bz_w = (cf *)calloc((size_t)big_l, sizeof(cf));   /* w[j], 0 from j = N on */
    p->bz_tl = (cf *)malloc(sizeof(cf) * (size_t)big_l); /* T_L[q1 k2] at q1 n2 + k2 */
    p->bz_wk = (cf *)malloc(sizeof(cf) * (size_t)nk);
    /* the lags at their cyclic index j = m mod L, planar */
    float *br = (float *)calloc((size_t)big_l, sizeof(float)), *bi = (float *)calloc((size_t)big_l, sizeof(float));
    double c, s;
    for (int64_t k = p->info.kmin; k < p->info.kmax; ++k) {
        bz_chirp_d(k, n, &c, &s);
        p->bz_wk[k - p->info.kmin].r = (float)(c / (double)big_l);
        p->bz_wk[k - p->info.kmin].i = (float)(s / (double)big_l);
    }
    for (int64_t j = 0; j < big_l; ++j) {
        if (j < n) {
            bz_chirp_d(j, n, &c, &s);
            p->bz_w[j].r = (float)c;
            p->bz_w[j].i = (float)s;
            if (j <= p->info.kmax - 1) { /* the lag m = j: conj(w[m]) */
                br[j] = (float)c;
                bi[j] = (float)(-s);
            }
        }
        int64_t neg = big_l - j; /* the lag m = j - L */
        if (neg <= n - 1 - p->info.kmin) {
            bz_chirp_d(neg, n, &c, &s);
            br[j] = (float)c;
            bi[j] = (float)(-s);
        }
        bz_unit(((j / n2) * (j % n2)) % big_l, big_l, &c, &s);
        p->bz_tl[j].r = (float)c;
        p->bz_tl[j].i = (float)s;
    }
    /* Bhat[q1 + n1 q2] at [q1 n2 + q2] = the first transform itself applied to the lags */
    p->bz_bhat = (cf *)malloc(sizeof(cf) * (size_t)big_l);
    bz_first_transform(p, br, bi, p->bz_bhat);
    free(br);
    free(bi);
}

hpfw_oracle_plan *hpfw_oracle_plan_create2(int64_t n, int force_bluestein) { return hpfw_oracle_plan_create3(n, force_bluestein, 0); }

hpfw_oracle_plan *hpfw_oracle_plan_create3(int64_t n, int force_bluestein, unsigned conventions)
{
    if (n < 2 || conventions > 15u) return NULL;
    hpfw_oracle_plan *p = (hpfw_oracle_plan *)calloc(1, sizeof(*p));
    p->info.n_samples = n;
    p->conv = conventions;
    {
        int64_t rest = n;
        while (rest % 2 == 0) rest /= 2;
        while (rest % 3 == 0) rest /= 3;
        while (rest % 5 == 0) rest /= 5;
        while (rest % 7 == 0) rest /= 7;
        p->bluestein = force_bluestein || rest != 1;
    }
    if (p->bluestein) {
        if (make_bands(p) != 0) {
            free(p);
            return NULL;
        }
        /* n2 = 6300; n1 = 16 a, the smallest such that n1 n2 >= N + nk - 1 (the first transform's column stage splits
         * into transforms of length a and 16) */
        int64_t need = n + (p->info.kmax - p->info.kmin) - 1, n1 = (need + 16 * 6300 - 1) / (16 * 6300) * 16;
        if (n1 > 8192) {
            free(p);
            return NULL;
        }
        p->info.n1 = n1;
        p->info.n2 = 6300;
        p->info.h = 6300 / 2 + 1;
        p->bz_l = n1 * 6300;
        p->info.n_radix = make_rows_radix_list(6300, p->info.radix);
        p->tw_n2 = make_twiddle_table(6300);
        p->tw_n1 = make_twiddle_table(n1);
        p->pos_n2 = (int32_t *)malloc(sizeof(int32_t) * 6300);
        for (int64_t k = 0; k < 6300; ++k)
            p->pos_n2[k] = (int32_t)hpfw_oracle_digit_pos(k, 6300, p->info.radix, p->info.n_radix);
        make_forward_bluestein(p);
        if (make_bluestein(p) != 0) {
            hpfw_oracle_plan_destroy(p);
            return NULL;
        }
        return p;
    }
    /* N = n1 * n2 with n2 <= N2_MAX: d0 = the smallest divisor that fits; among the divisors in
     * [d0, 5 d0 / 4] prefer an even n2 (its twiddle table halves exactly), then the fewest pairs of
     * radix passes, then the smallest n1 */
    int64_t n1 = 0, best_odd = 2, best_groups = 1 << 30, d0 = 0;
    for (int64_t d = 1; d <= n; ++d) {
        if (n % d || n / d > N2_MAX) continue;
        if (d0 == 0) d0 = d;
        if (4 * d > 5 * d0) break;
        int32_t tmp_r[HPFW_O_MAXRADIX];
        int nr = make_radix_list(n / d, tmp_r);
        if (nr < 0) break; /* not 7-smooth: rejected below */
        int64_t odd = (n / d) & 1, groups = (nr + 1) / 2;
        if (odd < best_odd || (odd == best_odd && groups < best_groups)) {
            best_odd = odd;
            best_groups = groups;
            n1 = d;
        }
    }
    if (n1 == 0) {
        free(p);
        return NULL;
    }
    int64_t n2 = n / n1;
    p->info.n1 = n1;
    p->info.n2 = n2;
    p->info.h = n2 / 2 + 1;
    p->info.n_radix = make_rows_radix_list(n2, p->info.radix);
    if (p->info.n_radix < 0 || make_bands(p) != 0) {
        free(p);
        return NULL;
    }
    /* n1 itself must also be 7-smooth for N to be: check through the radix helper */
    int32_t tmp[HPFW_O_MAXRADIX];
    if (make_radix_list(n1, tmp) < 0) {
        free(p);
        return NULL;
    }
    p->tw_n2 = make_twiddle_table(n2);
    p->tw_n1 = make_twiddle_table(n1);
    p->wq = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)n1);
    for (int64_t m = 0; m < n1; ++m) {
        double c, s;
        twiddle_d(m, n1, &c, &s);
        p->wq[2 * m] = (int32_t)rint(c * 4194304.0);
        p->wq[2 * m + 1] = (int32_t)rint(s * 4194304.0);
    }
    const int64_t hq = n1 / 2 + 1;
    p->ts = (cf *)malloc(sizeof(cf) * (size_t)(hq * n2));
    for (int64_t q1 = 0; q1 < hq; ++q1) {
        cf step[4];
        for (int e = 1; e < 4; ++e) hpfw_oracle_twiddle((q1 * e) % n, n, &step[e].r, &step[e].i);
        for (int64_t k2 = 0; k2 < n2; ++k2) {
            float re, im;
            hpfw_oracle_twiddle((q1 * (k2 & ~(int64_t)3)) % n, n, &re, &im);
            cf seed = {ldexpf(re, -37), ldexpf(im, -37)};
            p->ts[q1 * n2 + k2] = (k2 & 3) ? c_mul(seed, step[k2 & 3]) : seed;
        }
    }
    p->pos_n2 = (int32_t *)malloc(sizeof(int32_t) * (size_t)n2);
    for (int64_t k = 0; k < n2; ++k)
        p->pos_n2[k] = (int32_t)hpfw_oracle_digit_pos(k, n2, p->info.radix, p->info.n_radix);
    if (make_bluestein(p) != 0) {
        hpfw_oracle_plan_destroy(p);
        return NULL;
    }
    return p;
}

hpfw_oracle_plan *hpfw_oracle_plan_create(int64_t n) { return hpfw_oracle_plan_create2(n, 0); }

void hpfw_oracle_plan_destroy(hpfw_oracle_plan *p)
{
    if (!p) return;
    free(p->tw_n2);
    free(p->tw_n1);
    free(p->wq);
    free(p->ts);
    free(p->pos_n2);
    free(p->bz_w);
    free(p->bz_tl);
    free(p->bz_bhat);
    free(p->bz_wk);
    for (int j = 0; j < HPFW_O_BINS; ++j) free(p->g[j]);
    for (int k = 0; k < p->n_cls; ++k) {
        free(p->bc[k].tw);
        free(p->bc[k].vrev);
    }
    free(p);
}

void hpfw_oracle_plan_get_info(const hpfw_oracle_plan *p, hpfw_oracle_plan_info *out) { *out = p->info; }

void hpfw_oracle_plan_bands(const hpfw_oracle_plan *p, int32_t *start, int32_t *lg, int32_t *psize)
{
    memcpy(start, p->start, sizeof(p->start));
    memcpy(lg, p->lg, sizeof(p->lg));
    memcpy(psize, p->psize, sizeof(p->psize));
}

/* ------------------------------------------------------------------------------------------ */
/* a1 + forward DFT (S6).  x[n] = pcm[n] / 32768 (MonoLoader on PCM16 mono 44.1 kHz, cqt.h:45). */
/* X[k] = sum_n x[n] e^{-2 pi i k n / N}, k in [kmin, kmax), through N = n1 * n2:               */
/*   residues n = a + n1 * n2', pairs of residues packed into one complex length-n2 FFT,         */
/*   Hermitian split, twiddle T_N[a k2], then a length-n1 DFT as an fma chain over a.           */
/* ------------------------------------------------------------------------------------------ */
/* S15: the same bins when N has a prime factor above 7.  a[n] = x[n] w[n]; A = DFT_L(a) by rows (length n2
 * over t for every residue r), T_L[r k2], columns (fma chain over r, as above); C = conj(A Bhat);
 * F = DFT_L(C) the same way (rows over the residues of the flat index, T_L, columns) for the rows k1 that hold
 * consumed bins; X[k] = conj(F[k]) w[k] / L. */
static void spectrum_bluestein(const hpfw_oracle_plan *p, const int16_t *pcm, float *x_ri)
{
    const int64_t n = p->info.n_samples, n1 = p->info.n1, n2 = p->info.n2, big_l = n1 * n2;
    /* a[j] = x[j] w[j] by rows of n2 samples (flat index j = n2 k1' + k2'), zeros from j = N on */
    float *yr = (float *)malloc(sizeof(float) * (size_t)big_l), *yi = (float *)malloc(sizeof(float) * (size_t)big_l);
    for (int64_t j = 0; j < big_l; ++j) {
        float x = j < n ? (float)pcm[j] / 32768.0f : 0.0f;
        yr[j] = x * p->bz_w[j].r;
        yi[j] = x * p->bz_w[j].i;
    }
    /* first transform: columns (k1' -> q1), T_L, rows (k2' -> q2): A[q1 + n1 q2] at [q1][q2] */
    cf *ya = (cf *)malloc(sizeof(cf) * (size_t)big_l);
    bz_first_transform(p, yr, yi, ya);
    free(yr);
    free(yi);
    /* C = conj(A Bhat); second transform F = DFT_L(C), C indexed q1 + n1 q2: per q1 the row transform over q2 -> m2,
     * times T_L[q1 m2]; then the column chain over q1 for the rows m1 that hold consumed bins (m = m2 + n2 m1) */
    cf *z = (cf *)malloc(sizeof(cf) * (size_t)n2);
    for (int64_t q1 = 0; q1 < n1; ++q1) {
        for (int64_t q2 = 0; q2 < n2; ++q2) {
            cf v = c_mul(ya[q1 * n2 + q2], p->bz_bhat[q1 * n2 + q2]);
            z[q2].r = v.r;
            z[q2].i = -v.i;
        }
        fft_dif(z, n2, p->info.radix, p->info.n_radix, p->tw_n2);
        for (int64_t m2 = 0; m2 < n2; ++m2) ya[q1 * n2 + m2] = c_mul(z[p->pos_n2[m2]], p->bz_tl[q1 * n2 + m2]);
    }
    for (int64_t k = p->info.kmin; k < p->info.kmax; ++k) {
        int64_t m1 = k / n2, m2 = k % n2;
        float ar = 0.0f, ai = 0.0f;
        for (int64_t q1 = 0; q1 < n1; ++q1) {
            cf d = p->tw_n1[(q1 * m1) % n1];
            cf y = ya[q1 * n2 + m2];
            ar = fmaf(d.r, y.r, ar);
            ar = fmaf(-d.i, y.i, ar);
            ai = fmaf(d.i, y.r, ai);
            ai = fmaf(d.r, y.i, ai);
        }
        cf fc = {ar, -ai};
        cf v = c_mul(fc, p->bz_wk[k - p->info.kmin]);
        x_ri[2 * (k - p->info.kmin)] = v.r;
        x_ri[2 * (k - p->info.kmin) + 1] = v.i;
    }
    free(z);
    free(ya);
}

void hpfw_oracle_spectrum(const hpfw_oracle_plan *p, const int16_t *pcm, float *x_ri)
{
    if (p->bluestein) {
        spectrum_bluestein(p, pcm, x_ri);
        return;
    }
    /* S6 (7-smooth N = n1 n2): the clip as it lies is an [n1][n2] matrix of samples, x[n2 k1 + k2].
     *   columns: G[q1][k2] = sum_k1 wq[(q1 k1) mod n1] pcm[n2 k1 + k2], q1 <= n1 / 2 -- int16 samples times 23-bit
     *            fixed-point twiddles: exact integers (|G| < 2^50), rounded ONCE to f32;
     *   twiddle: z[k2] = G[q1][k2] ts[q1][k2]                      (S1; ts = T_N[q1 k2] 2^-37 from every fourth one, above)
     *   rows:    Z = FFT_n2(z) (S4), X[q1 + n1 q2] = Z[q2]; bins of the rows q1 > n1 / 2 by X[k] = conj(X[N - k]):
     *            X[(n1 - q1) + n1 q2] = conj(Z_q1[n2 - 1 - q2]). */
    const int64_t n1 = p->info.n1, n2 = p->info.n2, hq = n1 / 2 + 1;
    const int64_t kmin = p->info.kmin, kmax = p->info.kmax;
    const int64_t q2lo = kmin / n1, q2hi = (kmax - 1) / n1;
    int64_t *gr = (int64_t *)malloc(sizeof(int64_t) * (size_t)n2);
    int64_t *gi = (int64_t *)malloc(sizeof(int64_t) * (size_t)n2);
    cf *z = (cf *)malloc(sizeof(cf) * (size_t)n2);
    for (int64_t q1 = 0; q1 < hq; ++q1) {
        memset(gr, 0, sizeof(int64_t) * (size_t)n2);
        memset(gi, 0, sizeof(int64_t) * (size_t)n2);
        for (int64_t k1 = 0; k1 < n1; ++k1) {
            const int64_t wr = p->wq[2 * ((q1 * k1) % n1)], wi = p->wq[2 * ((q1 * k1) % n1) + 1];
            const int16_t *row = pcm + n2 * k1;
            for (int64_t k2 = 0; k2 < n2; ++k2) {
                gr[k2] += wr * (int64_t)row[k2];
                gi[k2] += wi * (int64_t)row[k2];
            }
        }
        for (int64_t k2 = 0; k2 < n2; ++k2) {
            cf g = {(float)gr[k2], (float)gi[k2]}; /* one rounding to nearest-even of the exact integer */
            z[k2] = c_mul(g, p->ts[q1 * n2 + k2]);
        }
        fft_dif(z, n2, p->info.radix, p->info.n_radix, p->tw_n2);
        for (int64_t q2 = q2lo; q2 <= q2hi; ++q2) {
            const int64_t k = q1 + n1 * q2;
            if (k >= kmin && k < kmax) {
                cf v = z[p->pos_n2[q2]];
                x_ri[2 * (k - kmin)] = v.r;
                x_ri[2 * (k - kmin) + 1] = v.i;
            }
            const int64_t km = (n1 - q1) + n1 * q2; /* the mirrored row, where it is not computed itself */
            if (q1 >= 1 && n1 - q1 >= hq && km >= kmin && km < kmax) {
                cf v = z[p->pos_n2[n2 - 1 - q2]];
                x_ri[2 * (km - kmin)] = v.r;
                x_ri[2 * (km - kmin) + 1] = -v.i;
            }
        }
    }
    free(z);
    free(gi);
    free(gr);
}

/* ------------------------------------------------------------------------------------------ */
/* a2 (inverse half) + a3 (S7).  For band j essentia multiplies the slice                        */
/* X[posit_j - floor(Lg/2) + i], i < Lg, by the Hann window, places it in a length-M buffer      */
/* and takes IFFT_M (cqt.h:66-71); hpfw keeps |c_j[3c]| (cqt.h:73-81).  The placement rotation   */
/* and the global phase only change the phase of c_j, so                                         */
/*   |c_j[3c]| = | (1/M) sum_i X[s_j + i] hann[i] e^{2 pi i 3 c i / M} |                          */
/* which is evaluated as a Bluestein chirp-z transform of length P_j >= Lg_j + C - 1.            */
/* ------------------------------------------------------------------------------------------ */
void hpfw_oracle_cqmag(const hpfw_oracle_plan *p, const float *x_ri, float *mag)
{
    const cf *x = (const cf *)x_ri;
    const int64_t c = p->info.c;
    int64_t pmax = 0;
    for (int k = 0; k < p->n_cls; ++k)
        if (p->bc[k].p > pmax) pmax = p->bc[k].p;
    cf *a = (cf *)malloc(sizeof(cf) * (size_t)pmax);
    for (int j = 0; j < HPFW_O_BINS; ++j) {
        const bluestein_class *b = &p->bc[p->cls[j]];
        const int64_t ps = b->p, lg = p->lg[j], off = p->start[j] - p->info.kmin;
        for (int64_t i = 0; i < lg; ++i) a[i] = c_mul(x[off + i], p->g[j][i]);
        memset(a + lg, 0, sizeof(cf) * (size_t)(ps - lg));
        fft_dif(a, ps, b->radix, b->nr, b->tw);
        for (int64_t i = 0; i < ps; ++i) a[i] = c_mul(a[i], b->vrev[i]);
        fft_idit(a, ps, b->radix, b->nr, b->tw);
        for (int64_t i = 0; i < c; ++i) mag[j * c + i] = sqrtf(fmaf(a[i].r, a[i].r, a[i].i * a[i].i));
    }
    free(a);
}

/* ------------------------------------------------------------------------------------------ */
/* a4 (S8).  convert.h:18-25 squares, convert.h:7-16 takes 10 log10 relative to the maximum     */
/* with a 1e-10 floor and clips 80 dB below the (new) maximum, which is 0 by construction.       */
/* log10 is evaluated in double by a fixed sequence of IEEE operations.                          */
/* ------------------------------------------------------------------------------------------ */
double hpfw_oracle_log10(double x)
{
    union { double d; uint64_t u; } v;
    v.d = x;
    int e = (int)((v.u >> 52) & 0x7ff) - 1023;
    v.u = (v.u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL; /* m in [1, 2) */
    double m = v.d;
    if (m > 1.4142135623730951) { m *= 0.5; e += 1; }
    double f = m - 1.0;
    double s = f / (2.0 + f);
    double z = s * s;
    /* atanh series: log(m) = 2 s (1 + z/3 + z^2/5 + ... + z^11/23) */
    double r = 1.0 / 23.0;
    r = fma(r, z, 1.0 / 21.0);
    r = fma(r, z, 1.0 / 19.0);
    r = fma(r, z, 1.0 / 17.0);
    r = fma(r, z, 1.0 / 15.0);
    r = fma(r, z, 1.0 / 13.0);
    r = fma(r, z, 1.0 / 11.0);
    r = fma(r, z, 1.0 / 9.0);
    r = fma(r, z, 1.0 / 7.0);
    r = fma(r, z, 1.0 / 5.0);
    r = fma(r, z, 1.0 / 3.0);
    r = fma(r, z, 1.0);
    double lm = 2.0 * s * r;
    return fma((double)e, 0.30102999566398119521, lm * 0.43429448190325182765);
}

static inline float db_term(float pw)
{
    float x = pw < 1e-10f ? 1e-10f : pw;
    return (float)(10.0 * hpfw_oracle_log10((double)x));
}

void hpfw_oracle_db(const float *mag, int64_t n, float *s_db)
{
    float pmax = 0.0f;
    for (int64_t i = 0; i < n; ++i) {
        float pw = mag[i] * mag[i];
        if (pw > pmax) pmax = pw;
    }
    float ref = db_term(pmax);
    for (int64_t i = 0; i < n; ++i) {
        float l = db_term(mag[i] * mag[i]) - ref;
        s_db[i] = l < -80.0f ? -80.0f : l;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* a5 + a6 (S9).  frames[b*20 + t, n] = S[b, n + t] (hashprint_handle.h:84-90, row-major        */
/* flatten of the 121 x 20 block), P = filters * frames (parallel_collector.h:57) as one fma    */
/* chain per output over k = b*20 + t ascending, starting from 0.                                */
/* ------------------------------------------------------------------------------------------ */
void hpfw_oracle_project(const float *f, const float *s_db, int64_t c, float *proj)
{
    const int64_t nf = c - (HPFW_O_CTX - 1);
    for (int64_t n = 0; n < nf; ++n) {
        float acc[HPFW_O_NFILT];
        for (int r = 0; r < HPFW_O_NFILT; ++r) acc[r] = 0.0f;
        for (int b = 0; b < HPFW_O_BINS; ++b) {
            for (int t = 0; t < HPFW_O_CTX; ++t) {
                const float sv = s_db[b * c + n + t];
                const float *fk = f + (size_t)(b * HPFW_O_CTX + t) * HPFW_O_NFILT;
                for (int r = 0; r < HPFW_O_NFILT; ++r) acc[r] = fmaf(fk[r], sv, acc[r]);
            }
        }
        for (int r = 0; r < HPFW_O_NFILT; ++r) proj[r * nf + n] = acc[r];
    }
}

/* a7 + a8.  bit (63 - r) of hp[i] = (P[r,i] - P[r,i+80] >= 0)  hashprint_handle.h:119-122,137-142 */
void hpfw_oracle_pack(const float *proj, int64_t nf, uint64_t *hp)
{
    for (int64_t i = 0; i + HPFW_O_LAG < nf; ++i) {
        uint64_t v = 0;
        for (int r = 0; r < HPFW_O_NFILT; ++r) {
            float d = proj[r * nf + i] - proj[r * nf + i + HPFW_O_LAG];
            if (d >= 0.0f) v |= 1ULL << (63 - r);
        }
        hp[i] = v;
    }
}

/* HashprintHandle<N, SH, FramesContext, T> for any template arguments (hashprint_handle.h:50-64): calc_frames
 * (:79-93: k = row * context + t), filters * frames as an fma chain over k ascending, calc_fingerprint
 * (:115-125) and bool_col_to_num (:137-142: bit (bits - 1 - r) <-> filter row r).  f column-major
 * [bits][rows * context]: (r, k) at r + bits * k; s row-major [rows][stride] with `cols` valid columns;
 * proj [bits][stride - context + 1]; hp: one 64-bit word per hashprint (the low `bits` bits used). */
void hpfw_oracle_project_cfg(const float *f, const float *s, int rows, int context, int bits, int64_t cols, int64_t stride,
                             float *proj)
{
    const int64_t nf = cols - context + 1, pst = stride - context + 1;
    for (int64_t n = 0; n < nf; ++n)
        for (int r = 0; r < bits; ++r) {
            float acc = 0.0f;
            for (int row = 0; row < rows; ++row)
                for (int t = 0; t < context; ++t)
                    acc = fmaf(f[(size_t)r + (size_t)bits * (size_t)(row * context + t)], s[(int64_t)row * stride + n + t], acc);
            proj[(int64_t)r * pst + n] = acc;
        }
}

void hpfw_oracle_pack_cfg(const float *proj, int bits, int lag, int64_t n_frames, int64_t proj_stride, uint64_t *hp)
{
    for (int64_t i = 0; i + lag < n_frames; ++i) {
        uint64_t v = 0;
        for (int r = 0; r < bits; ++r) {
            float d = proj[(int64_t)r * proj_stride + i] - proj[(int64_t)r * proj_stride + i + lag];
            if (d >= 0.0f) v |= 1ULL << (bits - 1 - r);
        }
        hp[i] = v;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* S9q / S10q: the projection in fixed point, exactly.                                           */
/* The reference multiplies filters and frames in f32 (an Eigen/MKL sgemm, parallel_collector.h:57,127) and keeps    */
/* only the SIGN of P[r,i] - P[r,i+80] (hashprint_handle.h:119-122).  Here both factors are rounded once to fixed     */
/* point and the 2420-term sums are exact integers:                                                                   */
/*   u[b][c]  = rint(clamp(S[b][c], -80, 0) * 98304)        (S in [-80, 0] dB already, convert.h:12-15; 98304 = 3 2^15 */
/*                                                          keeps the difference of two columns inside 24 bits)       */
/*   fq[r][k] = rint(F[r][k] * 2^(21 - ilogb(max_k |F[r][k]|)))   (a positive power of two per filter row: it cannot  */
/*                                                          change a sign)                                            */
/*   Pq[r][n] = sum_{b,t} fq[r][20 b + t] * u[b][n + t]    (int64, |Pq| < 2^57)                                       */
/*   D[r][i]  = Pq[r][i] - Pq[r][i + 80] = sum_{b,t} fq[r][20 b + t] * (u[b][i + t] - u[b][i + 80 + t])              */
/*   bit (63 - r) of hp[i] = (D[r][i] >= 0)                                                                           */
/* The distance to the real-number projection (rounding of S: 5.1e-6 dB, of F: 2^-23 of the row's largest entry) is   */
/* below the rounding error of an f32 sgemm over 2420 terms in any order; integer sums have no order.                 */
/* ------------------------------------------------------------------------------------------ */
#define HPFW_O_QSCALE 98304.0f

static int32_t q_fixed(float s)
{
    s = s < -80.0f ? -80.0f : (s > 0.0f ? 0.0f : s);
    return (int32_t)rintf(s * HPFW_O_QSCALE);
}

void hpfw_oracle_quantise_filters(const float *f, int32_t *fq /* [64][2420], row-major */)
{
    const int kk = HPFW_O_BINS * HPFW_O_CTX;
    for (int r = 0; r < HPFW_O_NFILT; ++r) {
        float m = 0.0f;
        for (int k = 0; k < kk; ++k) m = fmaxf(m, fabsf(f[(size_t)k * HPFW_O_NFILT + r]));
        const int e = m > 0.0f ? 21 - ilogbf(m) : 0;
        for (int k = 0; k < kk; ++k) fq[(size_t)r * kk + k] = (int32_t)rintf(ldexpf(f[(size_t)k * HPFW_O_NFILT + r], e));
    }
}

void hpfw_oracle_quantise_db(const float *s_db, int64_t count, int32_t *u)
{
    for (int64_t i = 0; i < count; ++i) u[i] = q_fixed(s_db[i]);
}

void hpfw_oracle_project_q(const float *f, const float *s_db, int64_t c, int64_t *proj /* [64][c - 19] */)
{
    const int64_t nf = c - (HPFW_O_CTX - 1);
    const int kk = HPFW_O_BINS * HPFW_O_CTX;
    int32_t *fq = (int32_t *)malloc(sizeof(int32_t) * (size_t)HPFW_O_NFILT * kk);
    int32_t *u = (int32_t *)malloc(sizeof(int32_t) * (size_t)(HPFW_O_BINS * c));
    hpfw_oracle_quantise_filters(f, fq);
    hpfw_oracle_quantise_db(s_db, HPFW_O_BINS * c, u);
    for (int r = 0; r < HPFW_O_NFILT; ++r) {
        int64_t *pr = proj + (int64_t)r * nf;
        for (int64_t n = 0; n < nf; ++n) pr[n] = 0;
        for (int b = 0; b < HPFW_O_BINS; ++b)
            for (int t = 0; t < HPFW_O_CTX; ++t) {
                const int64_t w = fq[(size_t)r * kk + b * HPFW_O_CTX + t];
                const int32_t *ub = u + b * c + t;
                for (int64_t n = 0; n < nf; ++n) pr[n] += w * (int64_t)ub[n];
            }
    }
    free(u);
    free(fq);
}

/* the sums whose signs are the hashprint bits, with the difference taken first (what the GPU kernel forms): */
/* D[r][i] = sum_{b,t} fq[r][20 b + t] * (u[b][i + t] - u[b][i + 80 + t]), i < c - 99                          */
void hpfw_oracle_delta_q(const float *f, const float *s_db, int64_t c, int64_t *delta /* [64][c - 99] */)
{
    const int64_t nhp = c - (HPFW_O_CTX - 1) - HPFW_O_LAG;
    const int kk = HPFW_O_BINS * HPFW_O_CTX;
    if (nhp <= 0) return;
    int32_t *fq = (int32_t *)malloc(sizeof(int32_t) * (size_t)HPFW_O_NFILT * kk);
    int32_t *u = (int32_t *)malloc(sizeof(int32_t) * (size_t)(HPFW_O_BINS * c));
    int32_t *du = (int32_t *)malloc(sizeof(int32_t) * (size_t)(HPFW_O_BINS * c));
    hpfw_oracle_quantise_filters(f, fq);
    hpfw_oracle_quantise_db(s_db, HPFW_O_BINS * c, u);
    for (int b = 0; b < HPFW_O_BINS; ++b)
        for (int64_t col = 0; col < c; ++col) du[b * c + col] = col + HPFW_O_LAG < c ? u[b * c + col] - u[b * c + col + HPFW_O_LAG] : 0;
    for (int r = 0; r < HPFW_O_NFILT; ++r) {
        int64_t *dr = delta + (int64_t)r * nhp;
        for (int64_t n = 0; n < nhp; ++n) dr[n] = 0;
        for (int b = 0; b < HPFW_O_BINS; ++b)
            for (int t = 0; t < HPFW_O_CTX; ++t) {
                const int64_t w = fq[(size_t)r * kk + b * HPFW_O_CTX + t];
                const int32_t *db_ = du + b * c + t;
                for (int64_t n = 0; n < nhp; ++n) dr[n] += w * (int64_t)db_[n];
            }
    }
    free(du);
    free(u);
    free(fq);
}

void hpfw_oracle_pack_q(const int64_t *proj, int64_t nf, uint64_t *hp)
{
    for (int64_t i = 0; i + HPFW_O_LAG < nf; ++i) {
        uint64_t v = 0;
        for (int r = 0; r < HPFW_O_NFILT; ++r)
            if (proj[r * nf + i] - proj[r * nf + i + HPFW_O_LAG] >= 0) v |= 1ULL << (63 - r);
        hp[i] = v;
    }
}

/* 1 (default): the fixed-point projection S9q / S10q; 0: the f32 fma chain of S9 / S10.  What extraction uses. */
static int g_projection_mode = 1;
void hpfw_oracle_set_projection(int mode) { g_projection_mode = mode ? 1 : 0; }
int hpfw_oracle_get_projection(void) { return g_projection_mode; }

int64_t hpfw_oracle_extract(const hpfw_oracle_plan *p, const float *f, const int16_t *pcm, uint64_t *hp)
{
    const int64_t c = p->info.c, nk = p->info.kmax - p->info.kmin;
    if (p->info.n_hp <= 0) return 0;
    float *x = (float *)malloc(sizeof(float) * 2 * (size_t)nk);
    float *mag = (float *)malloc(sizeof(float) * (size_t)(HPFW_O_BINS * c));
    float *proj = (float *)malloc(sizeof(float) * (size_t)(HPFW_O_NFILT * p->info.n_frames));
    hpfw_oracle_spectrum(p, pcm, x);
    hpfw_oracle_cqmag(p, x, mag);
    hpfw_oracle_db(mag, HPFW_O_BINS * c, mag);
    if (g_projection_mode) {
        int64_t *pq = (int64_t *)malloc(sizeof(int64_t) * (size_t)(HPFW_O_NFILT * p->info.n_frames));
        hpfw_oracle_project_q(f, mag, c, pq);
        hpfw_oracle_pack_q(pq, p->info.n_frames, hp);
        free(pq);
    } else {
        hpfw_oracle_project(f, mag, c, proj);
        hpfw_oracle_pack(proj, p->info.n_frames, hp);
    }
    free(proj);
    free(mag);
    free(x);
    return p->info.n_hp;
}

typedef struct {
    const hpfw_oracle_plan *p;
    const float *f;
    const int16_t *pcm;
    uint64_t *hp;
    int64_t lo, hi;
} extract_job;

static void *extract_worker(void *arg)
{
    extract_job *j = (extract_job *)arg;
    for (int64_t i = j->lo; i < j->hi; ++i)
        hpfw_oracle_extract(j->p, j->f, j->pcm + i * j->p->info.n_samples, j->hp + i * j->p->info.n_hp);
    return NULL;
}

int64_t hpfw_oracle_extract_batch(const hpfw_oracle_plan *p, const float *f, const int16_t *pcm,
                                  int64_t n_clips, uint64_t *hp, int n_threads)
{
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    pthread_t th[256];
    extract_job jobs[256];
    int64_t chunk = (n_clips + n_threads - 1) / n_threads;
    int used = 0;
    for (int t = 0; t < n_threads; ++t) {
        int64_t lo = t * chunk, hi = lo + chunk > n_clips ? n_clips : lo + chunk;
        if (lo >= hi) break;
        extract_job jb = {p, f, pcm, hp, lo, hi};
        jobs[used] = jb;
        pthread_create(&th[used], NULL, extract_worker, &jobs[used]);
        ++used;
    }
    for (int t = 0; t < used; ++t) pthread_join(th[t], NULL);
    return p->info.n_hp;
}

/* ------------------------------------------------------------------------------------------ */
/* a9.  storage.h:33-54                                                                         */
/* ------------------------------------------------------------------------------------------ */
void hpfw_oracle_match_clip(const uint64_t *q, int64_t k, const uint64_t *r, int64_t n,
                            uint64_t *best_dist, int64_t *best_off)
{
    uint64_t best = UINT64_MAX;
    int64_t boff = 0;
    if (n < k) k = n;
    for (int64_t i = 0; i < n - k + 1; ++i) {
        uint64_t cnt = 0;
        for (int64_t j = 0; j < k; ++j) cnt += (uint64_t)__builtin_popcountll(q[j] ^ r[i + j]);
        if (cnt < best) {
            best = cnt;
            boff = i;
        }
    }
    *best_dist = best;
    *best_off = boff;
}

/* Two steps so that every host thread has work whatever the shape (many queries x few clips, or a handful of
 * queries against 10^5 clips): (1) the per-(query, clip) table of storage.h:33-54 results, work items claimed in
 * blocks from a shared counter; (2) per query, the top-k of its row. */
typedef struct {
    const uint64_t *db;
    const int64_t *db_off;
    int64_t n_clips;
    const uint64_t *q;
    const int64_t *q_off;
    int64_t n_q;
    uint32_t *t_dist; /* [n_q][n_clips]; 0xffffffff = the pair was skipped (empty clip or query) */
    int32_t *t_off;
    int64_t next;     /* shared work counter (blocks of 16 pairs) */
} search_job;

static void *search_worker(void *arg)
{
    search_job *s = (search_job *)arg;
    const int64_t total = s->n_q * s->n_clips, blk = 16;
    for (;;) {
        int64_t lo = __atomic_fetch_add(&s->next, blk, __ATOMIC_RELAXED);
        if (lo >= total) break;
        int64_t hi = lo + blk > total ? total : lo + blk;
        for (int64_t w = lo; w < hi; ++w) {
            int64_t qi = w / s->n_clips, cidx = w % s->n_clips;
            int64_t n = s->db_off[cidx + 1] - s->db_off[cidx];
            int64_t k = s->q_off[qi + 1] - s->q_off[qi];
            s->t_dist[w] = 0xffffffffu;
            s->t_off[w] = 0;
            if (n <= 0 || k <= 0) continue;
            uint64_t d;
            int64_t off;
            hpfw_oracle_match_clip(s->q + s->q_off[qi], k, s->db + s->db_off[cidx], n, &d, &off);
            s->t_dist[w] = (uint32_t)d;
            s->t_off[w] = (int32_t)off;
        }
    }
    return NULL;
}

void hpfw_oracle_search_topk(const uint64_t *db, const int64_t *db_off, int64_t n_clips,
                             const uint64_t *q, const int64_t *q_off, int64_t n_q, int topk,
                             hpfw_oracle_hit *out, int n_threads)
{
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    const int64_t total = n_q * n_clips;
    search_job job = {db, db_off, n_clips, q, q_off, n_q, NULL, NULL, 0};
    job.t_dist = (uint32_t *)malloc((size_t)(total > 0 ? total : 1) * sizeof(uint32_t));
    job.t_off = (int32_t *)malloc((size_t)(total > 0 ? total : 1) * sizeof(int32_t));
    pthread_t th[256];
    int used = 0;
    if (n_threads > 1 && total > 16) {
        for (int t = 0; t < n_threads && (int64_t)t * 16 < total; ++t)
            if (pthread_create(&th[used], NULL, search_worker, &job) == 0) ++used;
    }
    search_worker(&job);
    for (int t = 0; t < used; ++t) pthread_join(th[t], NULL);
    for (int64_t qi = 0; qi < n_q; ++qi) {
        hpfw_oracle_hit *top = out + qi * topk;
        int have = 0;
        for (int t = 0; t < topk; ++t) {
            top[t].dist = 0xffffffffu;
            top[t].clip = 0xffffffffu;
            top[t].offset = 0;
            top[t].pad = 0;
        }
        for (int64_t cidx = 0; cidx < n_clips; ++cidx) {
            int64_t n = db_off[cidx + 1] - db_off[cidx];
            int64_t k = q_off[qi + 1] - q_off[qi];
            if (n <= 0 || k <= 0) continue;
            const uint32_t d = job.t_dist[qi * n_clips + cidx];
            /* insertion by ascending (dist, clip): clips arrive in ascending id, so a strict
             * comparison on dist keeps the earlier clip first (storage.h:56) */
            int pos = have;
            while (pos > 0 && top[pos - 1].dist > d) --pos;
            if (pos >= topk) continue;
            int last = have < topk ? have : topk - 1;
            for (int t = last; t > pos; --t) top[t] = top[t - 1];
            top[pos].dist = d;
            top[pos].clip = (uint32_t)cidx;
            top[pos].offset = job.t_off[qi * n_clips + cidx];
            top[pos].pad = 0;
            if (have < topk) ++have;
        }
    }
    free(job.t_dist);
    free(job.t_off);
}

/* ------------------------------------------------------------------------------------------ */
/* f3: the Mel front-end (mel.h:34-104), DESIGN.md appendix B and S11-S15                        */
/* ------------------------------------------------------------------------------------------ */
struct hpfw_oracle_mel {
    float window[HPFW_O_MEL_FRAME];
    float coeff[HPFW_O_MEL_BANDS * HPFW_O_MEL_BINS];
    int32_t radix[HPFW_O_MAXRADIX];
    int nr;
    cf *tw;
    int32_t *pos; /* digit-reversed position of output k */
};

static double hz2mel_htk(double f) { return 2595.0 * log10(1.0 + f / 700.0); }
static double mel2hz_htk(double m) { return 700.0 * (pow(10.0, m / 2595.0) - 1.0); }

hpfw_oracle_mel *hpfw_oracle_mel_create(void)
{
    hpfw_oracle_mel *m = (hpfw_oracle_mel *)calloc(1, sizeof(hpfw_oracle_mel));
    const int n = HPFW_O_MEL_FRAME;
    /* essentia Windowing: hann 0.5 - 0.5 cos(2 pi i / (size - 1)), normalized: area 1, times 2 */
    double sum = 0.0;
    for (int i = 0; i < n; ++i) sum += 0.5 - 0.5 * cos(2.0 * M_PI * (double)i / (double)(n - 1));
    for (int i = 0; i < n; ++i)
        m->window[i] = (float)((0.5 - 0.5 * cos(2.0 * M_PI * (double)i / (double)(n - 1))) * (2.0 / sum));
    m->nr = make_rows_radix_list(n, m->radix);
    m->tw = make_twiddle_table(n);
    m->pos = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    for (int k = 0; k < n; ++k) m->pos[k] = (int32_t)hpfw_oracle_digit_pos(k, n, m->radix, m->nr);
    /* essentia MelBands(inputSize 2206, numberBands 33; defaults: 0 .. 22050 Hz, htkMel, weighting
     * "warping", normalize "unit_sum", type "power") -> TriangularBands on 35 mel-spaced frequencies */
    double fb[HPFW_O_MEL_BANDS + 2];
    const double lo = hz2mel_htk(0.0), hi = hz2mel_htk(22050.0), inc = (hi - lo) / (HPFW_O_MEL_BANDS + 1);
    for (int i = 0; i < HPFW_O_MEL_BANDS + 2; ++i) fb[i] = mel2hz_htk(lo + inc * i);
    const double fscale = (44100.0 / 2.0) / (double)(HPFW_O_MEL_BINS - 1);
    for (int i = 0; i < HPFW_O_MEL_BANDS; ++i) {
        const double w0 = hz2mel_htk(fb[i]), w1 = hz2mel_htk(fb[i + 1]), w2 = hz2mel_htk(fb[i + 2]);
        const int jb = (int)(fb[i] / fscale + 0.5), je = (int)(fb[i + 2] / fscale + 0.5);
        double c[HPFW_O_MEL_BINS];
        double weight = 0.0;
        for (int j = 0; j < HPFW_O_MEL_BINS; ++j) c[j] = 0.0;
        for (int j = jb; j <= je && j < HPFW_O_MEL_BINS; ++j) {
            const double bf = j * fscale;
            if (bf >= fb[i] && bf < fb[i + 1]) c[j] = (hz2mel_htk(bf) - w0) / (w1 - w0);
            else if (bf >= fb[i + 1] && bf < fb[i + 2]) c[j] = (w2 - hz2mel_htk(bf)) / (w2 - w1);
            weight += c[j];
        }
        for (int j = 0; j < HPFW_O_MEL_BINS; ++j)
            m->coeff[i * HPFW_O_MEL_BINS + j] = (float)(weight > 0.0 ? c[j] / weight : 0.0);
    }
    return m;
}

void hpfw_oracle_mel_destroy(hpfw_oracle_mel *m)
{
    if (!m) return;
    free(m->tw);
    free(m->pos);
    free(m);
}

void hpfw_oracle_mel_tables(const hpfw_oracle_mel *m, float *window, float *coeff)
{
    memcpy(window, m->window, sizeof(m->window));
    memcpy(coeff, m->coeff, sizeof(m->coeff));
}

/* essentia FrameCutter(frameSize 4410, hopSize 441, startFromZero false): frame f starts at sample
 * 441 f - 2205 (zeros outside the signal); frames are cut while the start lies inside the signal */
int64_t hpfw_oracle_mel_frames(int64_t n)
{
    return n <= 0 ? 0 : (n + HPFW_O_MEL_FRAME / 2 + HPFW_O_MEL_HOP - 1) / HPFW_O_MEL_HOP;
}

void hpfw_oracle_mel_power(const hpfw_oracle_mel *m, const int16_t *pcm, int64_t n, float *power, uint8_t *keep)
{
    const int fs = HPFW_O_MEL_FRAME, half = fs / 2, nb = HPFW_O_MEL_BINS;
    const int64_t nfr = hpfw_oracle_mel_frames(n);
    cf *z = (cf *)malloc(sizeof(cf) * (size_t)fs);
    float *pw = (float *)malloc(sizeof(float) * (size_t)(2 * nb));
    for (int64_t f0 = 0; f0 < nfr; f0 += 2) { /* frames in pairs through one complex transform (S6) */
        for (int w = 0; w < 2; ++w) {
            const int64_t f = f0 + w, start = f * HPFW_O_MEL_HOP - half;
            int64_t e = 0; /* essentia isSilent: instantPower < 1e-10  <=>  sum pcm^2 <= 473 (exact in integers) */
            if (f < nfr)
                for (int i = 0; i < fs; ++i) {
                    const int64_t sidx = start + i;
                    if (sidx >= 0 && sidx < n) e += (int64_t)pcm[sidx] * pcm[sidx];
                }
            if (f < nfr) keep[f] = e > 473;
            for (int t = 0; t < fs; ++t) { /* zero-phase: transform input t is windowed sample (t + 2205) mod 4410 */
                const int i = t < fs - half ? t + half : t - (fs - half);
                const int64_t sidx = start + i;
                float v = 0.0f;
                if (f < nfr && sidx >= 0 && sidx < n) v = ((float)pcm[sidx] / 32768.0f) * m->window[i];
                if (w == 0) z[t].r = v; else z[t].i = v;
            }
        }
        fft_dif(z, fs, m->radix, m->nr, m->tw);
        for (int k = 0; k < nb; ++k) {
            const cf zk = z[m->pos[k]], zm = z[m->pos[k == 0 ? 0 : fs - k]];
            const cf va = {0.5f * (zk.r + zm.r), 0.5f * (zk.i - zm.i)};
            const cf vb = {0.5f * (zk.i + zm.i), 0.5f * (zm.r - zk.r)};
            const float ma = sqrtf(fmaf(va.r, va.r, va.i * va.i)), mb = sqrtf(fmaf(vb.r, vb.r, vb.i * vb.i));
            pw[k] = ma * ma;          /* essentia Spectrum gives the magnitude, MelBands (type power) squares it */
            pw[nb + k] = mb * mb;
        }
        for (int w = 0; w < 2 && f0 + w < nfr; ++w)
            for (int i = 0; i < HPFW_O_MEL_BANDS; ++i) {
                float acc = 0.0f; /* fma chain over the bins, ascending (the MFMA order) */
                for (int j = 0; j < nb; ++j) acc = fmaf(m->coeff[i * nb + j], pw[w * nb + j], acc);
                power[(int64_t)i * nfr + f0 + w] = acc;
            }
    }
    free(z);
    free(pw);
}

int64_t hpfw_oracle_mel_spectrogram(const hpfw_oracle_mel *m, const int16_t *pcm, int64_t n, float *out)
{
    const int64_t nfr = hpfw_oracle_mel_frames(n);
    float *power = (float *)malloc(sizeof(float) * (size_t)(HPFW_O_MEL_BANDS * (nfr > 0 ? nfr : 1)));
    uint8_t *keep = (uint8_t *)malloc((size_t)(nfr > 0 ? nfr : 1));
    hpfw_oracle_mel_power(m, pcm, n, power, keep);
    int64_t c = 0;
    for (int64_t f = 0; f < nfr; ++f)
        if (keep[f]) {
            for (int i = 0; i < HPFW_O_MEL_BANDS; ++i) out[(int64_t)i * nfr + c] = power[(int64_t)i * nfr + f];
            ++c;
        }
    /* power_to_db (convert.h:7-16) over the kept columns */
    float pmax = 0.0f;
    for (int i = 0; i < HPFW_O_MEL_BANDS; ++i)
        for (int64_t k = 0; k < c; ++k)
            if (out[(int64_t)i * nfr + k] > pmax) pmax = out[(int64_t)i * nfr + k];
    const float ref = db_term(pmax);
    for (int i = 0; i < HPFW_O_MEL_BANDS; ++i)
        for (int64_t k = 0; k < c; ++k) {
            const float l = db_term(out[(int64_t)i * nfr + k]) - ref;
            out[(int64_t)i * nfr + k] = l < -80.0f ? -80.0f : l;
        }
    free(power);
    free(keep);
    return c;
}

/* annoy_storage.h:41-63 with exact neighbours (see hpfw_oracle.h) */
void hpfw_oracle_knn_windows(const uint64_t *db, const int64_t *db_off, int64_t n_clips, const uint64_t *q,
                             int64_t k, int win, int nn, uint64_t *keys)
{
    const int64_t n_win = k - win + 1;
    for (int64_t i = 0; i < n_win; ++i) {
        uint64_t *best = keys + i * nn;
        for (int r = 0; r < nn; ++r) best[r] = ~0ull;
        for (int64_t c = 0; c < n_clips; ++c) {
            const int64_t r0 = db_off[c], n = db_off[c + 1] - r0;
            for (int64_t p = 0; p + win <= n; ++p) {
                uint64_t d = 0;
                for (int w = 0; w < win; ++w) d += (uint64_t)__builtin_popcountll(q[i + w] ^ db[r0 + p + w]);
                uint64_t key = (d << 40) | (uint64_t)(r0 + p);
                if (key >= best[nn - 1]) continue;
                int r = nn - 1;
                while (r > 0 && best[r - 1] > key) {
                    best[r] = best[r - 1];
                    --r;
                }
                best[r] = key;
            }
        }
    }
}

void hpfw_oracle_vote_windows(const uint64_t *keys, int64_t n_win, int nn, const int64_t *db_off, int64_t n_clips,
                              hpfw_oracle_vote *out)
{
    /* buckets (clip, offset) -> float count; at most n_win * nn of them */
    const int64_t cap = n_win > 0 ? n_win * nn : 1;
    int64_t *b_clip = (int64_t *)malloc((size_t)cap * sizeof(int64_t));
    int64_t *b_off = (int64_t *)malloc((size_t)cap * sizeof(int64_t));
    float *b_cnt = (float *)malloc((size_t)cap * sizeof(float));
    int64_t nb = 0;
    out->clip = -1;
    out->offset = 0;
    out->cnt = 0.0f;
    out->pad = 0.0f;
    for (int64_t i = 0; i < n_win; ++i)
        for (int r = 0; r < nn; ++r) {
            const uint64_t key = keys[i * nn + r];
            if (key == ~0ull) continue;
            const uint64_t d = key >> 40;
            const int64_t pos = (int64_t)(key & ((1ull << 40) - 1));
            int64_t c = 0;
            while (c + 1 < n_clips && db_off[c + 1] <= pos) ++c; /* clip holding this position */
            const int64_t off = i - (pos - db_off[c]);
            int64_t s = 0;
            while (s < nb && !(b_clip[s] == c && b_off[s] == off)) ++s;
            if (s == nb) {
                b_clip[s] = c;
                b_off[s] = off;
                b_cnt[s] = 0.0f;
                ++nb;
            }
            b_cnt[s] = (float)((double)b_cnt[s] + 1.0 / (double)(float)(d + 1)); /* annoy_storage.h:53 */
            if (b_cnt[s] > out->cnt) {
                out->clip = c;
                out->offset = off;
                out->cnt = b_cnt[s];
            }
        }
    free(b_clip);
    free(b_off);
    free(b_cnt);
}

/* FNV-1a checksums of the plan tables, so a test can compare them with the product's tables
 * without either side exposing the tables themselves. */
static uint64_t fnv1a(const void *data, size_t bytes, uint64_t h)
{
    const unsigned char *p = (const unsigned char *)data;
    for (size_t i = 0; i < bytes; ++i) {
        h ^= p[i];
        h *= 1099511628211ULL;
    }
    return h;
}

/* the chirp-z tables (0 w, 1 T_L, 2 Bhat: [n1][n2]; 3 w[k] / L: [kmax - kmin]) as (re, im) floats; returns the
 * number of floats, 0 for a plan without them; out may be NULL */
/* which = 4 (every plan): the constant-Q stage's windows G_j, bands concatenated (S5: sum of lg values) */
int64_t hpfw_oracle_chirpz_table(const hpfw_oracle_plan *p, int which, float *out)
{
    if (which == 4) {
        int64_t count = 0;
        for (int j = 0; j < HPFW_O_BINS; ++j) {
            if (out) memcpy(out + count, p->g[j], sizeof(cf) * (size_t)p->lg[j]);
            count += 2 * (int64_t)p->lg[j];
        }
        return count;
    }
    if (!p->bluestein || which < 0 || which > 3) return 0;
    const cf *tab[4] = {p->bz_w, p->bz_tl, p->bz_bhat, p->bz_wk};
    int64_t count = 2 * (which == 3 ? p->info.kmax - p->info.kmin : p->bz_l);
    if (out) memcpy(out, tab[which], sizeof(float) * (size_t)count);
    return count;
}

void hpfw_oracle_plan_checksum(const hpfw_oracle_plan *p, uint64_t *out8)
{
    const uint64_t seed = 1469598103934665603ULL;
    out8[0] = fnv1a(p->tw_n2, (size_t)p->info.n2 * 8, seed);
    out8[1] = fnv1a(p->tw_n1, (size_t)p->info.n1 * 8, seed);
    out8[2] = p->ts ? fnv1a(p->wq, (size_t)p->info.n1 * 8, fnv1a(p->ts, (size_t)((p->info.n1 / 2 + 1) * p->info.n2) * 8, seed)) : seed;
    out8[3] = fnv1a(p->pos_n2, (size_t)p->info.n2 * 4, seed);
    uint64_t h = fnv1a(p->start, sizeof(p->start), seed);
    h = fnv1a(p->lg, sizeof(p->lg), h);
    out8[4] = fnv1a(p->psize, sizeof(p->psize), h);
    h = seed;
    for (int j = 0; j < HPFW_O_BINS; ++j) h = fnv1a(p->g[j], (size_t)p->lg[j] * 8, h);
    out8[5] = h;
    uint64_t ht = seed, hv = seed;
    for (int k = 0; k < p->n_cls; ++k) {
        ht = fnv1a(p->bc[k].tw, (size_t)p->bc[k].p * 8, ht);
        hv = fnv1a(p->bc[k].vrev, (size_t)p->bc[k].p * 8, hv);
    }
    out8[6] = ht;
    out8[7] = hv;
}
