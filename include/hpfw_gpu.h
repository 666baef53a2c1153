/*
 * hpfw_gpu.h -- C-ABI of the MI355X (gfx950) hashprint hot path: the drop-in boundary.
 *
 * What it replaces (paths relative to the hpfw reference tree):
 *   extraction  ParallelCollector::calc_hashprint / collect_fingerprints
 *               include/hpfw/core/parallel_collector.h:54-59, 115-137, i.e.
 *               CQT<>::spectrogram            include/hpfw/spectrum/cqt.h:36-84
 *               amplitude_to_db/power_to_db   include/hpfw/spectrum/convert.h:7-25
 *               calc_frames, filters*frames, calc_fingerprint, fingerprint_to_hashprint
 *                                             include/hpfw/core/hashprint_handle.h:79-142
 *   search      MemoryStorage::build / find   include/hpfw/audioproblems/live-song-id/storage.h:21-64
 *               and the notebook's top-10 rule examples/python/liveid.ipynb cell 9
 *   legacy FFI  the eight extern "C" symbols of modules/python/parallel_collector_wrapper.hpp:21-38
 *               (declared at the end of this file with the same shapes)
 *
 * Conventions
 *   - plain C, no C++ or torch types; every function returns 0 on success and a negative
 *     hpfw_status otherwise and never throws; hpfw_gpu_last_error() returns a thread-local
 *     message for the last failure on the calling thread.
 *   - pointers named d_* are DEVICE pointers (HBM) on the handle's device; all others are host
 *     pointers.  `stream` is a hipStream_t passed as void* (NULL = the default stream); device
 *     entry points only enqueue work on it and return without synchronising.
 *   - a handle may be used by one host thread at a time.
 *   - layouts: PCM is int16 mono 44.1 kHz, clips of equal length back to back;
 *     spectrograms are bin-major [121][C] (the reference's Eigen matrix is column-major 121 x C:
 *     element (b, c) at b + 121 c; ours is at b * C + c); filters are the reference's
 *     Matrix<float,64,Dynamic> column-major: element (r, k) at r + 64 k, k = bin * 20 + t;
 *     hashprints are uint64, bit (63 - r) <-> filter row r (hashprint_handle.h:137-142).
 */
#ifndef HPFW_GPU_H
#define HPFW_GPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HPFW_BINS 121     /* cqt.h:21 NumberBins                      */
#define HPFW_CONTEXT 20   /* live_song_id.h:16 FramesContext          */
#define HPFW_LAG 80       /* live_song_id.h:16 T                      */
#define HPFW_FILTERS 64   /* hashprint_handle.h:64 sizeof(uint64_t)*8 */
#define HPFW_FRAME_SIZE (HPFW_BINS * HPFW_CONTEXT)

typedef enum {
    HPFW_OK = 0,
    HPFW_E_INVALID = -1,     /* bad argument                                       */
    HPFW_E_UNSUPPORTED = -2, /* clip too short (below ~1.3 s) or too long (~18 min) */
    HPFW_E_NOFILTERS = -3,   /* extraction before hpfw_gpu_set_filters             */
    HPFW_E_HIP = -4,         /* a HIP runtime call failed (message has the detail) */
    HPFW_E_NOMEM = -5,
    HPFW_E_IO = -6           /* legacy file entry points: unreadable / unsupported WAV */
} hpfw_status;

typedef struct hpfw_gpu hpfw_gpu; /* opaque */

/* geometry of clips of n_samples samples: essentia NSGConstantQ as configured at cqt.h:54-61 */
typedef struct {
    int64_t n_samples;
    int64_t n1, n2;   /* forward transform split N = n1 * n2                       */
    int64_t kmin, kmax; /* forward DFT bins [kmin, kmax) consumed by the 121 bands  */
    int64_t m;        /* M: inverse transform length of every band                 */
    int64_t c;        /* spectrogram columns ceil(M / 3)                           */
    int64_t n_frames; /* c - 19    hashprint_handle.h:84                           */
    int64_t n_hp;     /* c - 99    hashprint_handle.h:118                          */
} hpfw_geometry;

/* one search result; SearchResult{filename, cnt, offset} of storage.h:11-15 with the filename
 * replaced by the clip's index in hpfw_gpu_index_add order */
typedef struct {
    uint32_t dist;   /* sum of popcounts over the query                  */
    uint32_t clip;   /* clip_base + index in add order; 0xffffffff = none */
    int32_t offset;  /* first offset reaching dist (storage.h:50-53)      */
    uint32_t pad;
} hpfw_hit;

const char *hpfw_gpu_last_error(void);
const char *hpfw_gpu_version(void);

int hpfw_gpu_create(int device, hpfw_gpu **out);
void hpfw_gpu_destroy(hpfw_gpu *h);
int hpfw_gpu_device(const hpfw_gpu *h); /* the device ordinal the handle was created on */

/* filters = ParallelCollector::filters (parallel_collector.h:77), host pointer, 64 x 2420 floats */
int hpfw_gpu_set_filters(hpfw_gpu *h, const float *filters_colmajor);
/* the sizes of a clip length (columns, frames, hashprints: cqt.h:66-73, hashprint_handle.h:79-93).  Host arithmetic only:
 * no table of the length is built or uploaded for the question; a length the extraction would refuse (a factor n2 beyond
 * the LDS) is refused here with the same status.  Like every entry point but hpfw_gpu_prepare_length it belongs to the one
 * host thread that drives the handle. */
int hpfw_gpu_geometry(hpfw_gpu *h, int64_t n_samples, hpfw_geometry *out);

/* essentia's NSGConstantQ is not vendored with hpfw and its version is not pinned (CMakeLists.txt:36), so four of
 * its conventions are restated from the published algorithm and cannot be checked offline (DESIGN.md appendix
 * A).  They are switchable per handle: a maintainer holding one real essentia output can pin them without
 * touching a kernel (the tables of every clip length are rebuilt).  0 = the defaults. */
#define HPFW_CONV_HANN_PERIODIC 1u  /* window 0.5 - 0.5 cos(2 pi i / L) instead of 2 pi i / (L - 1)                 */
#define HPFW_CONV_LG_HALF_EVEN 2u   /* Lg = round-half-to-even(bw / fftres) instead of round-half-away-from-zero   */
#define HPFW_CONV_FLOAT_GEOMETRY 4u /* fftres, f_j, posit_j, Lg_j evaluated in float (essentia's Real), not double */
#define HPFW_CONV_NO_IFFT_SCALE 8u  /* band transforms without the inverse FFT's 1/M (seen only by the 1e-10 floor) */
int hpfw_gpu_set_conventions(hpfw_gpu *h, unsigned flags);

/* ---- extraction: calc_hashprint for n_clips clips of n_samples samples each ------------- */
/* d_hp receives [n_clips][n_hp] */
int hpfw_gpu_extract_pcm16(hpfw_gpu *h, const int16_t *d_pcm, int64_t n_samples, int64_t n_clips,
                           uint64_t *d_hp, void *stream);
/* host buffers; copies in, runs, copies out, synchronises */
int hpfw_gpu_extract_pcm16_host(hpfw_gpu *h, const int16_t *pcm, int64_t n_samples, int64_t n_clips,
                                uint64_t *hp);
/* clips processed per internal pass at most (workspace = ~9.5 MB per clip at 30 s; a call splits into passes of equal size); 0 = default (256) */
int hpfw_gpu_set_batch(hpfw_gpu *h, int clips_per_pass);

/* The smallest supported clip length >= n_samples, or -1 beyond the longest supported clip.  Host-only.
 * Any length between the shortest clip that yields a hashprint (54 254 samples, 1.23 s) and the longest the
 * tables allow is supported as it is -- the reference hands the file's exact sample count to NSGConstantQ
 * (cqt.h:54-55): 7-smooth lengths (every multiple of 1/7 s at 44.1 kHz among them) take the mixed-radix forward
 * transform, all others the chirp-z (Bluestein) one, about three times slower.  Nothing is padded. */
int64_t hpfw_gpu_supported_length(int64_t n_samples);

/* ---- per-stage entry points (parity checkpoints; same kernels the full chain runs) ------- */
/* PCM -> forward DFT bins [kmin,kmax): d_x [n_clips][kmax-kmin][2]           cqt.h:45-52,66 */
int hpfw_gpu_stage_spectrum(hpfw_gpu *h, const int16_t *d_pcm, int64_t n_samples, int64_t n_clips,
                            float *d_x, void *stream);
/* bins -> |c_j[3c]|: d_mag [n_clips][121][C]                                  cqt.h:66-81 */
int hpfw_gpu_stage_cqmag(hpfw_gpu *h, const float *d_x, int64_t n_samples, int64_t n_clips,
                         float *d_mag, void *stream);
/* amplitude_to_db, per clip: d_mag, d_db [n_clips][121][C] (may alias)        convert.h:7-25 */
int hpfw_gpu_stage_db(hpfw_gpu *h, const float *d_mag, int64_t n_clips, int64_t c, float *d_db,
                      void *stream);
/* filters * calc_frames(S): d_proj [n_clips][64][C-19]       hashprint_handle.h:79-93 + :57 */
int hpfw_gpu_stage_project(hpfw_gpu *h, const float *d_db, int64_t n_clips, int64_t c,
                           float *d_proj, void *stream);
/* calc_fingerprint + fingerprint_to_hashprint: d_hp [n_clips][n_frames-80]   :115-142 */
int hpfw_gpu_stage_pack(hpfw_gpu *h, const float *d_proj, int64_t n_clips, int64_t n_frames,
                        uint64_t *d_hp, void *stream);

/* PCM -> dB spectrogram through the front end exactly as extraction runs it (the chirp-z kernel
 * writes dB terms, the reference level is applied afterwards): d_db [n_clips][121][C]
 *                                                              cqt.h:45-81 + convert.h:7-25 */
int hpfw_gpu_stage_spectrogram(hpfw_gpu *h, const int16_t *d_pcm, int64_t n_samples,
                               int64_t n_clips, float *d_db, void *stream);

/* ---- Mel front-end: spectrum::MelSpectrogram<44100, 33, 4410, 441>::spectrogram (mel.h:34-104) ----
 * essentia FrameCutter(4410, 441) -> Windowing(hann) -> Spectrum -> MelBands(33) per frame, silent frames
 * dropped (mel.h:94-96), power_to_db over the kept columns (mel.h:103).  Any clip length.
 * hpfw_gpu_mel_frames: frames cut from n_samples (silent ones included) = the row stride of the output.
 * d_out [n_clips][33][frames]: the kept columns at the front of every row; d_cols [n_clips] their number. */
int64_t hpfw_gpu_mel_frames(int64_t n_samples);
int hpfw_gpu_mel_spectrogram_pcm16(hpfw_gpu *h, const int16_t *d_pcm, int64_t n_samples,
                                   int64_t n_clips, float *d_out, int32_t *d_cols, void *stream);
int hpfw_gpu_mel_spectrogram_pcm16_host(hpfw_gpu *h, const int16_t *pcm, int64_t n_samples,
                                        int64_t n_clips, float *out, int32_t *cols);

/* ---- HashprintHandle<N, SpectrogramHandler, FramesContext, T> with other template arguments -------------
 * (hashprint_handle.h:50-64).  Everything above is the live-id default <uint64_t, CQT<>, 20, 80>
 * (live_song_id.h:16); this is the same calc_frames / filters * frames / calc_fingerprint /
 * fingerprint_to_hashprint (hashprint_handle.h:79-142) for any spectrogram height, context, lag and word
 * size -- the combiner's HashPrint<uint16_t, MelSpectrogram<>, 32, 50> (combiner.h:12) first of all. */
typedef struct {
    int32_t rows;    /* Spectrogram::RowsAtCompileTime: 33 for MelSpectrogram<> (mel.h:17-21), 121 for CQT<> */
    int32_t context; /* FramesContext                                                                          */
    int32_t lag;     /* T                                                                                      */
    int32_t bits;    /* 8 * sizeof(N) = NumOfFilters (hashprint_handle.h:64): 16, 32 or 64                     */
} hpfw_handle_config;
#define HPFW_CONFIG_COMBINER {33, 32, 50, 16} /* combiner.h:12 */
/* filters: Matrix<float, NumOfFilters, Dynamic> column-major, element (r, k) at r + bits * k, k = row * context + t */
int hpfw_gpu_cfg_set_filters(hpfw_gpu *h, const hpfw_handle_config *cfg, const float *filters_colmajor);
/* d_s [n_clips][rows][stride] (row-major: element (row, col) of clip i at (i * rows + row) * stride + col);
 * d_cols [n_clips]: valid columns of each clip, or NULL = stride (the Mel front end drops silent frames, so its
 * clips differ).  d_hp: uintN [n_clips][hp_stride]; clip i receives max(cols_i - context + 1 - lag, 0) words.
 * d_proj (optional, NULL to skip): the projection filters * frames [n_clips][bits][stride - context + 1]. */
int hpfw_gpu_cfg_hashprints(hpfw_gpu *h, const hpfw_handle_config *cfg, const float *d_s, const int32_t *d_cols,
                            int64_t n_clips, int64_t stride, void *d_hp, int64_t hp_stride, float *d_proj, void *stream);
/* filter learning for such a configuration: calc_cov of every clip's frames (hashprint_handle.h:96-102: centred on the
 * clip's own frame means, / (n_frames - 1); clips with fewer than two frames add nothing) accumulated on the GPU as
 * ParallelCollector::preprocess does (parallel_collector.h:93-97), then calc_filters (hashprint_handle.h:105-112) on the
 * host: the `bits` leading eigenvectors become the configuration's filters.  d_s / d_cols / stride as above.
 * cov: host, [rows * context][rows * context].  context >= 9. */
int hpfw_gpu_cfg_cov_reset(hpfw_gpu *h, const hpfw_handle_config *cfg);
int hpfw_gpu_cfg_cov_accumulate(hpfw_gpu *h, const hpfw_handle_config *cfg, const float *d_s, const int32_t *d_cols,
                                int64_t n_clips, int64_t stride, void *stream);
int hpfw_gpu_cfg_cov_get(hpfw_gpu *h, const hpfw_handle_config *cfg, float *cov, int64_t *n_clips);
int hpfw_gpu_cfg_learn_filters(hpfw_gpu *h, const hpfw_handle_config *cfg, float *filters_colmajor_out);
/* the combiner's Algo end to end on host buffers: MelSpectrogram<44100, 33, 4410, 441>::spectrogram (mel.h:34-104)
 * + HashprintHandle<uint16_t, Mel, 32, 50>: hp [n_clips][hp_stride] (hp_stride >= hpfw_gpu_mel_frames(n) - 81),
 * n_hp [n_clips] the number of hashprints of each clip (0 when too few frames are left after the silent ones) */
int hpfw_gpu_mel_hashprints_pcm16_host(hpfw_gpu *h, const int16_t *pcm, int64_t n_samples, int64_t n_clips,
                                       uint16_t *hp, int64_t hp_stride, int32_t *n_hp);

/* ---- filter learning: ParallelCollector::preprocess + calc_filters ------------------------
 * (parallel_collector.h:82-112, hashprint_handle.h:96-112).  The handle owns accum_cov
 * (2420 x 2420, parallel_collector.h:76): per clip, the covariance of its context frames (centred
 * on the clip's own mean, / (n_frames - 1)) is added on the GPU (f32 MFMA); learn_filters takes the
 * eigenvectors of the 64 largest eigenvalues on the host and installs them as the filters.
 * Eigenvector signs are arbitrary in the reference; here the largest component is positive. */
int hpfw_gpu_cov_reset(hpfw_gpu *h);
int hpfw_gpu_cov_accumulate_pcm16(hpfw_gpu *h, const int16_t *d_pcm, int64_t n_samples, int64_t n_clips,
                                  void *stream);
int hpfw_gpu_cov_accumulate_pcm16_host(hpfw_gpu *h, const int16_t *pcm, int64_t n_samples, int64_t n_clips);
/* stage entry point: from dB spectrograms d_db [n_clips][121][C] */
int hpfw_gpu_cov_accumulate_db(hpfw_gpu *h, const float *d_db, int64_t n_clips, int64_t c, void *stream);
/* host copies of accum_cov (full symmetric 2420 x 2420 floats; synchronises) and the file count */
int hpfw_gpu_cov_get(hpfw_gpu *h, float *cov, int64_t *n_files);
int hpfw_gpu_cov_set(hpfw_gpu *h, const float *cov, int64_t n_files);
/* filters_colmajor_out may be NULL; layout as hpfw_gpu_set_filters */
int hpfw_gpu_learn_filters(hpfw_gpu *h, float *filters_colmajor_out);
/* accum_cov where it lives: DEVICE pointer to 2420 x 2420 floats (allocated and zeroed on first use; the tiles
 * of 128 x 128 on or above the diagonal are maintained, the rest stays zero) and the number of files added --
 * what a multi-GPU host sums with one ncclAllReduce (include/hpfw_gpu_multi.h) */
int hpfw_gpu_cov_device(hpfw_gpu *h, float **d_cov);
int64_t hpfw_gpu_cov_files(hpfw_gpu *h);
int hpfw_gpu_cov_set_files(hpfw_gpu *h, int64_t n_files);
/* host-only: unit eigenvectors of the m largest eigenvalues of a symmetric n x n float matrix */
int hpfw_gpu_host_top_eigenvectors(const float *cov, int n, int m, float *out, double *evals);

/* Hashprints of one cached dB spectrogram, as collect_fingerprints computes them from
 * cache/spectros/<stem> (parallel_collector.h:114-137).  s_colmajor is the matrix as the cereal file
 * holds it (utils.h:77-106: int32 rows = 121, int32 cols, column-major floats).  *n_hp = cols - 99
 * (0 when the spectrogram is too short); hp receives them when hp_cap >= *n_hp. */
int hpfw_gpu_extract_db_host(hpfw_gpu *h, const float *s_colmajor, int32_t rows, int32_t cols,
                             uint64_t *hp, int64_t hp_cap, int64_t *n_hp);

/* ---- index + search: MemoryStorage::build / find ----------------------------------------- */
int hpfw_gpu_index_clear(hpfw_gpu *h);
/* appends n_clips hashprints; clip i is hp[offsets[i] .. offsets[i+1]); host or device source */
int hpfw_gpu_index_add(hpfw_gpu *h, const uint64_t *hp, const int64_t *offsets, int64_t n_clips);
int hpfw_gpu_index_add_device(hpfw_gpu *h, const uint64_t *d_hp, const int64_t *offsets,
                              int64_t n_clips, void *stream);
int64_t hpfw_gpu_index_size(hpfw_gpu *h); /* number of clips */
/* The index back on the host -- what MemoryStorage::save dumps (storage.h:67-75).  offsets
 * [n_clips + 1] is always written; hp [offsets[n_clips]] when hp != NULL (hp_cap = its capacity). */
int hpfw_gpu_index_get(hpfw_gpu *h, int64_t *offsets, uint64_t *hp, int64_t hp_cap);
/* added to every reported clip id (rank's first global clip id when the index is sharded) */
int hpfw_gpu_index_set_clip_base(hpfw_gpu *h, uint32_t clip_base);

/* top-k clips per query, ascending (dist, clip): query q is q_hp[q_off[q] .. q_off[q+1]).
 * d_q_hp device; q_off host; d_out device [n_q][k].  k <= 64. */
int hpfw_gpu_search_topk_device(hpfw_gpu *h, const uint64_t *d_q_hp, const int64_t *q_off,
                                int64_t n_q, int k, hpfw_hit *d_out, void *stream);
/* host convenience: copies queries in and hits out, synchronises */
int hpfw_gpu_search_topk(hpfw_gpu *h, const uint64_t *q_hp, const int64_t *q_off, int64_t n_q,
                         int k, hpfw_hit *out);
/* AnnStorage::find (annoy_storage.h:41-63) with the approximate Annoy forest replaced by exact nearest
 * neighbours.  Items are windows of 64 consecutive hashprints (the reference indexes 64 uint64 words per
 * item, annoy_storage.h:23,32; its items whose window runs past the end of a hashprint -- an
 * out-of-bounds read -- are not created).  For every position i of a query, the 5 windows of the index
 * nearest in Hamming distance over the 4096 bits, by (distance, position in the database), vote
 * cnt[clip][i - p] += 1 / (d + 1) (float accumulator, :53); the first bucket to exceed the running
 * maximum wins (:55-59).  Host arrays; out[n_q]; clip = 0xffffffff when the query has no window or the
 * index no item. */
typedef struct {
    uint32_t clip;
    uint32_t pad;
    int64_t offset; /* i - p of the winning bucket */
    float cnt;      /* its votes */
    float pad2;
} hpfw_vote;
int hpfw_gpu_search_votes(hpfw_gpu *h, const uint64_t *q_hp, const int64_t *q_off, int64_t n_q,
                          hpfw_vote *out);
/* the neighbours themselves (for tests): keys [n_windows][5], dist << 40 | global hashprint position,
 * ascending, ~0 where fewer exist; windows of all queries in order, n_windows = sum max(k - 63, 0) */
int hpfw_gpu_knn_windows(hpfw_gpu *h, const uint64_t *q_hp, const int64_t *q_off, int64_t n_q,
                         uint64_t *keys, int64_t keys_cap);

/* deterministic merge of per-shard top-k lists (e.g. after an all-gather): in [n_shards][n_q][k]
 * -> out [n_q][k], ascending (dist, clip).  Host arrays. */
int hpfw_gpu_merge_topk(const hpfw_hit *in, int n_shards, int64_t n_q, int k, hpfw_hit *out);

/* ---- events: time a region on `stream` with HIP events (bench.py uses these so that the
 * timing is taken on the stream the kernels run on) ---------------------------------------- */
int hpfw_gpu_timer_start(hpfw_gpu *h, void *stream);
int hpfw_gpu_timer_stop(hpfw_gpu *h, void *stream, float *ms); /* synchronises on the stop event */
/* per-kernel device time, measured with one HIP event pair per launch on the launch stream.
 * mask bit i enables kernel kind i in the order reported by hpfw_gpu_get_kernel_timing
 * (fwd_rows, fwd_cols, cq_chirpz, db, project_mfma, delta_pack, hamming_scan, topk, pcm_pairs); -1 = all,
 * 0 = off.  Setting the mask resets the accumulated times. */
int hpfw_gpu_set_kernel_timing(hpfw_gpu *h, int mask);
/* names[i] (static strings) and ms[i], launches[i] for i < *n; pass capacity in *n */
int hpfw_gpu_get_kernel_timing(hpfw_gpu *h, const char **names, float *ms, int *launches, int *n);

/* ---- host-only diagnostic: FNV-1a checksums of the eight groups of constant tables built for
 * clips of n_samples samples (twiddles, digit reversal, bands, window*chirp, chirp spectra).
 * No device is touched.  (For a chirp-z length slot 2, T_N, is the hash of nothing: see hpfw_gpu_chirpz_table.) */
int hpfw_gpu_plan_checksum(int64_t n_samples, uint64_t *out8);
/* the same with the chirp-z forward transform forced and under given conventions (HPFW_CONV_*) */
int hpfw_gpu_plan_checksum_ex(int64_t n_samples, int force_bluestein, unsigned conventions, uint64_t *out8);

/* ---- the projection's arithmetic.  The reference multiplies filters and frames in f32 (an Eigen/MKL sgemm,
 * parallel_collector.h:57,127) and keeps only the sign of P[r,i] - P[r,i+80] (hashprint_handle.h:119-122).  mode 1
 * (default): both factors rounded once to fixed point (S to 1/98304 dB, F to 2^-22 of its row's largest entry), the
 * lag-80 difference taken on the quantised spectrogram and its 2420-term sums with the filters exact integers on the
 * int8 matrix pipe (closer to the real-number product than an f32 sgemm in any order; no projection is
 * ever stored).  mode 0: the f32 fma chain in ascending k on the f32 matrix pipe.  The two differ in a
 * hashprint bit only where the difference of the two projections is within rounding of zero. */
int hpfw_gpu_set_projection(hpfw_gpu *h, int mode);
int hpfw_gpu_get_projection(hpfw_gpu *h);
/* dB spectrograms [n_clips][121][c] (device, as hpfw_gpu_stage_spectrogram writes them) -> hashprints
 * [n_clips][c - 99] with the handle's filters and projection mode */
int hpfw_gpu_hashprints_from_db(hpfw_gpu *h, const float *d_db, int64_t n_clips, int64_t c, uint64_t *d_hp, void *stream);
/* parity checkpoint of mode 1: the exact integer sums D[r][i] = sum_k fq[r][k] (u[k][i] - u[k][i + 80]) whose signs are
 * the hashprint bits, d_delta [n_clips][64][c - 99] int64 (device); d_hp may be NULL.  Same kernel as extraction. */
int hpfw_gpu_stage_delta_q(hpfw_gpu *h, const float *d_db, int64_t n_clips, int64_t c, int64_t *d_delta, uint64_t *d_hp,
                           void *stream);

/* ---- table preparation ahead of time.  A corpus of real recordings brings a new clip length with almost every file,
 * and the host half of a length's tables (constant-Q windows and chirp spectra, twiddles) costs more than the
 * extraction of the file: 3 ms for 30 s, 15 ms for 3 minutes.  hpfw_gpu_prepare_length builds that half on the
 * CALLING thread and keeps it for the next call that meets the length; it may be called from any number of threads
 * concurrently with any other call on the handle (the collectors' file-reader threads do so for every file they
 * decode).  Returns HPFW_E_UNSUPPORTED for a length outside the supported range. */
int hpfw_gpu_prepare_length(hpfw_gpu *h, int64_t n_samples);

/* ---- diagnostic: the tables of the chirp-z forward transform (clip lengths with a prime factor above 7), which are
 * generated on the device.  which: 0 = chirp w [n1][n2], 1 = T_L [n1][n2], 2 = Bhat [n1][n2],
 * 3 = w[k] / L [kmax - kmin]; complex as (re, im) float pairs.  *count = floats in the table; out may be NULL to
 * ask for the count only.  HPFW_E_INVALID for a 7-smooth length (unless HPFW_FORCE_BLUESTEIN is set).
 * which = 4, for EVERY length: the constant-Q stage's window table (121 bands concatenated, sum of Lg values), which is
 * generated on the device too (reference cqt.h:54-61: essentia builds these windows per file). */
int hpfw_gpu_chirpz_table(hpfw_gpu *h, int64_t n_samples, int which, float *out, int64_t capacity, int64_t *count);

/* ---- diagnostic: the handle's extraction workspaces as the last call left them (device pointers; valid until the next
 * call that grows them).  which: 0 = z, the column stage's output (chunked forward transform: one region per stream),
 * 1 = forward bins in the rows layout, 2 = dB terms / spectrograms, 4 = wave maxima.  Used by tools/ and tests to name
 * the first stage that differs; not part of the reference's interface. */
int hpfw_gpu_debug_workspace(hpfw_gpu *h, int which, void **d_ptr, size_t *bytes);

/* ---- legacy FFI: modules/python/parallel_collector_wrapper.hpp:12-38, same shapes --------- */
typedef struct {
    char *filename;
    uint64_t *hashprint;
    int hp_size;
} FilenameHashprintPair; /* wrapper.hpp:12-16 */

typedef struct hpfw_legacy_collector hpfw_legacy_collector; /* stands in for LiveIdCollector */

hpfw_legacy_collector *par_collector_new(void);                         /* wrapper.hpp:21 */
void par_collector_del(hpfw_legacy_collector *collector);               /* wrapper.hpp:23 */
FilenameHashprintPair *par_collector_prepare(hpfw_legacy_collector *collector,
                                             const char **filenames, int n, int *got); /* :25-28 */
uint64_t *par_collector_calc_hashprint(hpfw_legacy_collector *collector, const char *filename,
                                       int *size);                      /* wrapper.hpp:30 */
void par_collector_save(hpfw_legacy_collector *collector, const char *cache); /* wrapper.hpp:32 */
void par_collector_load(hpfw_legacy_collector *collector, const char *cache); /* wrapper.hpp:34 */
/* not in the reference's FFI: calc_hashprint for a list of files in one call (batched like prepare,
 * nothing learned) -- what LiveSongIdentification::search (live_song_id.h:37-41) does query by query.
 * n entries in input order, release with prepare_result_free(res, n); a file that failed has
 * hashprint == NULL and hp_size == 0; NULL when no filters are loaded */
FilenameHashprintPair *par_collector_calc_hashprints(hpfw_legacy_collector *collector,
                                                     const char **filenames, int n);
void prepare_result_free(FilenameHashprintPair *res, int got);          /* wrapper.hpp:36 */
void calc_hashprint_result_free(uint64_t *hp);                          /* wrapper.hpp:38 */

#ifdef __cplusplus
}
#endif
#endif
