/*
 * hpfw_oracle.h -- CPU restatement (plain C) of the hpfw index()/search() hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as
 * the checker / the reported CPU baseline.  The product path (hpfw_amd/csrc, include/) never
 * includes, links or calls anything from here.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures for this path
 *  and it cannot be compiled in this image (essentia, cereal, spdlog,
 * boost and the Eigen/Core umbrella header are absent; DESIGN.md "Oracle").  The restatement is
 * pinned instead to (1) the reference sources read as text, cited per function below, and
 * (2) an independent float64 numpy statement of the same mathematics (oracle/nsgt_f64.py).
 *
 * Every function states the reference file:line it follows.  Where the reference leaves the
 * floating-point evaluation order to a third party (essentia/FFTW, Eigen/MKL, -ffast-math), the
 * order is fixed by DESIGN.md "Arithmetic specification"; the HIP kernels implement the same
 * specification so that their results can be compared bit for bit.
 */
#ifndef HPFW_ORACLE_H
#define HPFW_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HPFW_O_BINS 121      /* CQT<..., NumberBins = 121, ...>  cqt.h:21                  */
#define HPFW_O_CTX 20        /* HashprintHandle<uint64_t, CQT<>, 20, 80>  live_song_id.h:16 */
#define HPFW_O_LAG 80
#define HPFW_O_NFILT 64      /* sizeof(uint64_t) * 8  hashprint_handle.h:64                 */
#define HPFW_O_FRAME (HPFW_O_BINS * HPFW_O_CTX) /* 2420  hashprint_handle.h:60              */
#define HPFW_O_MAXRADIX 24

typedef struct hpfw_oracle_plan hpfw_oracle_plan;

typedef struct {
    int64_t n_samples; /* N                                                        */
    int64_t n1;        /* number of residue classes (short DFT length)             */
    int64_t n2;        /* long in-core FFT length, N = n1 * n2                      */
    int64_t h;         /* n2 / 2 + 1 : stored half spectrum of each residue FFT     */
    int64_t kmin;      /* first forward-DFT bin consumed by any CQ band            */
    int64_t kmax;      /* one past the last                                        */
    int64_t m;         /* M = max_j Lg_j : length of every band's inverse DFT      */
    int64_t c;         /* C = ceil(M / 3) spectrogram columns                      */
    int64_t n_frames;  /* C - 19                                                   */
    int64_t n_hp;      /* C - 99 (0 when the clip is too short)                    */
    int32_t n_radix;   /* passes of the length-n2 FFT                              */
    int32_t radix[HPFW_O_MAXRADIX];
} hpfw_oracle_plan_info;

/* ---- geometry: essentia NSGConstantQ as configured at cqt.h:54-61 ---- */
hpfw_oracle_plan *hpfw_oracle_plan_create(int64_t n_samples); /* NULL when the clip is too short or too long */
/* force_bluestein: take the chirp-z forward transform (the path of lengths with a prime factor
 * above 7) even when N is 7-smooth */
hpfw_oracle_plan *hpfw_oracle_plan_create2(int64_t n_samples, int force_bluestein);
/* essentia conventions that cannot be checked offline (essentia is not vendored), switchable so that one real
 * essentia output can pin them: DESIGN.md appendix A.  0 = the restatement's defaults. */
#define HPFW_O_CONV_HANN_PERIODIC 1u  /* window 0.5 - 0.5 cos(2 pi i / L) instead of 2 pi i / (L - 1)     */
#define HPFW_O_CONV_LG_HALF_EVEN 2u   /* Lg rounded half-to-even instead of half-away-from-zero            */
#define HPFW_O_CONV_FLOAT_GEOMETRY 4u /* fftres, f_j, posit_j, Lg_j in float (essentia's Real), not double */
#define HPFW_O_CONV_NO_IFFT_SCALE 8u  /* band transforms without the inverse FFT's 1/M                     */
hpfw_oracle_plan *hpfw_oracle_plan_create3(int64_t n_samples, int force_bluestein, unsigned conventions);
/* S15: the chirp-z tables (0 w, 1 T_L, 2 Bhat: [n1][n2]; 3 w[k] / L: [kmax - kmin]) as (re, im) floats; returns the
 * number of floats (0 for a plan without them); out may be NULL */
int64_t hpfw_oracle_chirpz_table(const hpfw_oracle_plan *p, int which, float *out);
void hpfw_oracle_plan_destroy(hpfw_oracle_plan *p);
void hpfw_oracle_plan_get_info(const hpfw_oracle_plan *p, hpfw_oracle_plan_info *out);
/* per band j = 0..120: slice start in the forward DFT, window length Lg_j, Bluestein size P_j */
void hpfw_oracle_plan_bands(const hpfw_oracle_plan *p, int32_t *start, int32_t *lg, int32_t *psize);

/* ---- a1 + a2 (forward half): PCM16 -> forward DFT bins [kmin, kmax)   cqt.h:45-52, 66-71 ---- */
void hpfw_oracle_spectrum(const hpfw_oracle_plan *p, const int16_t *pcm, float *x_ri /* [kmax-kmin][2] */);
/* ---- a2 (band inverse DFT) + a3: |c_j[3c]|, bin-major [121][C]         cqt.h:66-81 ---- */
void hpfw_oracle_cqmag(const hpfw_oracle_plan *p, const float *x_ri, float *mag);
/* ---- a4: amplitude_to_db -> power_to_db                                convert.h:7-25 ---- */
void hpfw_oracle_db(const float *mag, int64_t n, float *s_db);
/* ---- a5 + a6: implicit calc_frames and filters * frames
 *      hashprint_handle.h:79-93, parallel_collector.h:57,127.
 *      f_colmajor: Filters = Matrix<float,64,Dynamic> column-major (hashprint_handle.h:68):
 *      element (r,k) at r + 64*k.  s_db bin-major [121][c].  out P row-major [64][c-19]. ---- */
void hpfw_oracle_project(const float *f_colmajor, const float *s_db, int64_t c, float *proj);
/* ---- a5..a8 in fixed point (S9q / S10q, see hpfw_oracle.c): exact integer sums of once-rounded factors ---- */
void hpfw_oracle_quantise_filters(const float *f_colmajor, int32_t *fq /* [64][2420] row-major */);
void hpfw_oracle_project_q(const float *f_colmajor, const float *s_db, int64_t c, int64_t *proj /* [64][c-19] */);
void hpfw_oracle_pack_q(const int64_t *proj, int64_t n_frames, uint64_t *hp /* [n_frames-80] */);
/* the same sums with the lag-80 difference taken first: delta[r][i] = proj[r][i] - proj[r][i + 80], [64][c - 99] */
void hpfw_oracle_delta_q(const float *f_colmajor, const float *s_db, int64_t c, int64_t *delta);
void hpfw_oracle_quantise_db(const float *s_db, int64_t count, int32_t *u);
/* which projection hpfw_oracle_extract* use: 0 = the f32 fma chain (S9), 1 = fixed point (S9q) */
void hpfw_oracle_set_projection(int mode);
int hpfw_oracle_get_projection(void);
/* ---- a7 + a8: calc_fingerprint + fingerprint_to_hashprint   hashprint_handle.h:115-142 ---- */
void hpfw_oracle_pack(const float *proj, int64_t n_frames, uint64_t *hp /* [n_frames-80] */);
/* ---- the same for any HashprintHandle<N, SH, FramesContext, T> (hashprint_handle.h:50-64), e.g. the
 *      combiner's <uint16_t, MelSpectrogram<>, 32, 50> (combiner.h:12): see hpfw_oracle.c ---- */
void hpfw_oracle_project_cfg(const float *f_colmajor, const float *s, int rows, int context, int bits, int64_t cols,
                             int64_t stride, float *proj);
void hpfw_oracle_pack_cfg(const float *proj, int bits, int lag, int64_t n_frames, int64_t proj_stride, uint64_t *hp);
/* ---- a1..a8 for one clip (calc_hashprint, parallel_collector.h:54-59); returns n_hp ---- */
int64_t hpfw_oracle_extract(const hpfw_oracle_plan *p, const float *f_colmajor, const int16_t *pcm,
                            uint64_t *hp);
/* ---- a10: static chunks of ceil(n/T) clips per thread (flow_builder.hpp:321-325) ---- */
int64_t hpfw_oracle_extract_batch(const hpfw_oracle_plan *p, const float *f_colmajor,
                                  const int16_t *pcm, int64_t n_clips, uint64_t *hp, int n_threads);

/* ---- a9: MemoryStorage::find inner loops, storage.h:33-54: per reference clip the first
 *      strict minimum over offsets of sum_j popcount(q[j] ^ r[off + j]), k = min(k, n). ---- */
void hpfw_oracle_match_clip(const uint64_t *q, int64_t k, const uint64_t *r, int64_t n,
                            uint64_t *best_dist, int64_t *best_off);

typedef struct {
    uint32_t dist;
    uint32_t clip;
    int32_t offset;
    uint32_t pad;
} hpfw_oracle_hit;

/* f3: MelSpectrogram<44100, 33, 4410, 441>::spectrogram (include/hpfw/spectrum/mel.h:34-104): essentia
 * FrameCutter(4410, 441) -> Windowing(hann) -> Spectrum -> MelBands(33) per frame, silent frames
 * dropped (mel.h:94-96), power_to_db over the kept columns (mel.h:103).  essentia is not vendored: the
 * algorithms are restated in DESIGN.md appendix B and, like the constant-Q, are PARITY UNPINNED.
 * Tables: window [4410] (hann, normalised to area 2, before the zero-phase rotation), coeff [33][2206]. */
#define HPFW_O_MEL_BANDS 33
#define HPFW_O_MEL_FRAME 4410
#define HPFW_O_MEL_HOP 441
#define HPFW_O_MEL_BINS 2206
typedef struct hpfw_oracle_mel hpfw_oracle_mel;
hpfw_oracle_mel *hpfw_oracle_mel_create(void);
void hpfw_oracle_mel_destroy(hpfw_oracle_mel *m);
void hpfw_oracle_mel_tables(const hpfw_oracle_mel *m, float *window, float *coeff);
int64_t hpfw_oracle_mel_frames(int64_t n_samples); /* frames cut from n samples, silent ones included */
/* band powers before the dB conversion: power [33][n_frames] (row stride n_frames), all frames;
 * keep [n_frames] = 1 for frames that are not silent */
void hpfw_oracle_mel_power(const hpfw_oracle_mel *m, const int16_t *pcm, int64_t n, float *power, uint8_t *keep);
/* the spectrogram: out [33][n_frames] (row stride n_frames) with the kept columns compacted to the front;
 * returns their number */
int64_t hpfw_oracle_mel_spectrogram(const hpfw_oracle_mel *m, const int16_t *pcm, int64_t n, float *out);

/* AnnStorage::find (annoy_storage.h:41-63) with the approximate Annoy forest replaced by the exact
 * nearest neighbours.  Items are windows of `win` (64: the reference indexes 64 consecutive uint64
 * words per item, AnnoyIndex<..., Hamming, ...>(64), annoy_storage.h:23,32) consecutive hashprints;
 * item id = global hashprint position db_off[clip] + p, p = 0 .. n_clip - win (the reference's items
 * whose window runs past the end of the hashprint -- an out-of-bounds read -- are not created).
 * knn: for every query position i = 0 .. k - win the `nn` items of smallest Hamming distance over the
 * win * 64 bits, ascending (distance, item id): keys[i][r] = dist << 40 | item id, ~0 when fewer exist. */
void hpfw_oracle_knn_windows(const uint64_t *db, const int64_t *db_off, int64_t n_clips, const uint64_t *q,
                             int64_t k, int win, int nn, uint64_t *keys);

typedef struct {
    int64_t clip;   /* -1: no match (annoy_storage.h:43 initial best_match) */
    int64_t offset; /* i - p of the winning (clip, offset) bucket            */
    float cnt;      /* its accumulated 1 / (d + 1) votes                     */
    float pad;
} hpfw_oracle_vote;

/* the voting of annoy_storage.h:45-61 over those neighbours, in the reference's order (i ascending,
 * then rank): cnt[clip][i - p] += 1.0 / (float)(d + 1) into a float; the first bucket to exceed the
 * running maximum wins. */
void hpfw_oracle_vote_windows(const uint64_t *keys, int64_t n_win, int nn, const int64_t *db_off, int64_t n_clips,
                              hpfw_oracle_vote *out);

/* top-k over the whole database, ascending (dist, clip id): storage.h:56-60 keeps the first
 * strict minimum in database order; the notebook keeps the 10 smallest (liveid.ipynb cell 9).
 * db: concatenated hashprints, db_off[n_clips+1].  Unused slots: dist = clip = 0xffffffff. */
void hpfw_oracle_search_topk(const uint64_t *db, const int64_t *db_off, int64_t n_clips,
                             const uint64_t *q, const int64_t *q_off, int64_t n_q, int topk,
                             hpfw_oracle_hit *out /* [n_q][topk] */, int n_threads);

/* FNV-1a checksums of the eight table groups (compared with the product's own tables) */
void hpfw_oracle_plan_checksum(const hpfw_oracle_plan *p, uint64_t *out8);

/* ---- pieces exposed so tests can pin them one by one ---- */
double hpfw_oracle_log10(double x);                                       /* DESIGN.md "dB" */
void hpfw_oracle_twiddle(int64_t m, int64_t n, float *re, float *im);     /* e^{-2 pi i m/n} */
/* in-place forward DIF FFT (digit-reversed output) and its inverse DIT twin */
void hpfw_oracle_fft_dif(float *a_ri, int64_t n, const int32_t *radix, int n_radix);
void hpfw_oracle_fft_idit(float *a_ri, int64_t n, const int32_t *radix, int n_radix);
int64_t hpfw_oracle_digit_pos(int64_t k, int64_t n, const int32_t *radix, int n_radix);

#ifdef __cplusplus
}
#endif
#endif
