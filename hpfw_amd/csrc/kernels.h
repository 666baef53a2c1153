// kernels.h -- host-callable launchers of the gfx950 kernels (one per hot loop of the reference,
// SURVEY.md section 2.1).  Each launcher only enqueues on `stream`.
#pragma once
#include <cstdint>
#include <vector>
#include <atomic>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_math.h"
#include "fft_lds.h"
#include "fft_rows.h"

namespace hpfw {

// "once per device" for hipFuncSetAttribute calls: a process may hold handles on several devices (one
// handle = one device), and a launcher may be entered from several host threads
struct PerDeviceOnce {
    std::atomic<uint64_t> done{0};
    static uint64_t bit()
    {
        int dev = 0;
        (void)hipGetDevice(&dev);
        return 1ull << (dev & 63);
    }
    bool need() const { return !(done.load(std::memory_order_acquire) & bit()); }
    void mark() { done.fetch_or(bit(), std::memory_order_release); } // after the calls: a second thread may repeat them, harmlessly
};


constexpr int kBins = 121;
constexpr int kCtx = 20;
constexpr int kLag = 80;
constexpr int kFilters = 64;
constexpr int kFrame = kBins * kCtx;

// S6, the column stage of the forward transform on the int8 matrix cores (k_forward.hip): the clip as it lies is an
// [n1][n2] sample matrix; rows q1 <= n1 / 2 of its length-n1 DFT down the columns, exact integers, rounded once to f32
// (the row stage multiplies them by the twiddles between the stages on its way in).
constexpr int kZBlock = 128;   // columns per block of the column stage's output (= columns per workgroup of both column kernels)
inline long long z_floats_per_clip(int hq, int n2) { return (long long)((n2 + kZBlock - 1) / kZBlock) * 2 * hq * kZBlock; }

struct ColsQArgs {
    int n1, n2, hq;      // hq = n1 / 2 + 1 rows
    int mt, ks;          // 32-row tiles of the (Re, Im) interleaved rows; 32-sample steps of k1
    const void *image;   // digits of the fixed-point twiddles as the A operand [mt][ks][3][64][16 bytes]
    const double *corr;  // [2 hq]: what the samples' +128 digit offset adds to every element of a row
    long long *stamps;   // diagnostic builds (-DHPFW_COLS_STAMPS) only: per workgroup 8 cycle sums
    int n_clips;         // set by the launch of the register-resident kernel: its grid is one-dimensional
    int variant;         // HPFW_COLS_VARIANT at handle creation (tests, diagnosis): 1 = the LDS-staged kernel for every n1
    long long zclip;     // floats of z per clip: z is [clip][block of kZBlock columns][row 2 q1 + (Re: 0, Im: 1)][kZBlock] (round 4: a
                         // workgroup's stores of a tile are then ONE contiguous 16 KB instead of 32 pieces 25 KB apart)
};

// Where the forward bins of one clip lie: in natural order from bin q0 on (n1 == 1: the chirp-z forward transform, the
// stage entry points), or as the row stage of S6 leaves them, x[k mod n1][k / n1 - q0] in rows of w elements.
#ifndef HPFW_XS_BATCH
#define HPFW_XS_BATCH 4
#endif
constexpr int kXsBatch = HPFW_XS_BATCH; // elements per thread whose loads are in flight together in the band loaders below
struct XsView {
    const cf *base;
    int n1, w, q0;
    unsigned long long magic; // ceil(2^40 / n1): k / n1 = (k magic) >> 40 for every k < 2^21, n1 <= 2^13
    HPFW_DEVICE_MEMBER cf operator()(int k) const
    {
        if (n1 == 1) return base[k - q0];
        const int q2 = (int)(((unsigned long long)(unsigned)k * magic) >> 40);
        return base[(int64_t)(k - q2 * n1) * w + (q2 - q0)];
    }
};
// the slice of one band: element i is bin start + i
struct XsBand {
    XsView v;
    int start;
    HPFW_DEVICE_MEMBER cf operator()(int i) const { return v(start + i); }
    // every element of the slice times its window value, in natural order: f(i, x[i] g[i])
    // (kXsBatch elements per thread at a time, all their loads issued before the first product: one element per trip left
    // one memory round trip per element in a row -- tools/cq_stamps.py: a fifth to a third of a band's time)
    template <class F>
    HPFW_DEVICE_MEMBER void for_each(int tid, int nthreads, int lg, const cf *__restrict__ g, F f) const
    {
        for (int i0 = tid; i0 < lg; i0 += kXsBatch * nthreads) {
            cf xv[kXsBatch], gv[kXsBatch];
#pragma unroll
            for (int u = 0; u < kXsBatch; ++u) {
                const int i = i0 + u * nthreads;
                const int ic = i < lg ? i : lg - 1; // (a valid element again: loaded, not used)
                xv[u] = v(start + ic);
                gv[u] = g[ic];
            }
#pragma unroll
            for (int u = 0; u < kXsBatch; ++u) {
                const int i = i0 + u * nthreads;
                if (i < lg) f(i, c_mul(xv[u], gv[u]));
            }
        }
    }
};
// the same slice walked in the order the rows layout stores it (row q1 = k mod n1, then the q2 of the slice, which are
// consecutive in memory), with the window permuted likewise on the host (CqPlanDev::g2): lanes read runs of a row
// instead of 64 different rows.  t = q1 nq2 + tq is bin k = q1 + n1 (q2a + tq), element i = k - start (skipped outside
// [0, lg): there the permuted window holds zeros).
struct XsBandRows {
    const cf *base;  // clip's first element
    int n1, w, q0;   // as XsView
    int start, q2a, nq2;
    unsigned magic;  // ceil(2^32 / nq2) (nq2 >= 2): t / nq2 = umulhi(t, magic) for every t < 2^19
    template <class F>
    HPFW_DEVICE_MEMBER void for_each(int tid, int nthreads, int lg, const cf *__restrict__ g2, F f) const
    {
        // every t < total names an element of the clip's rows (inside or outside the band) and an entry of g2: the loads
        // are unconditional and issued kXsBatch at a time, only the use is guarded (written as `if (inside) f(.., load ..)`
        // every element cost a memory round trip of its own: tools/cq_stamps.py)
        const int total = n1 * nq2;
        for (int t0 = tid; t0 < total; t0 += kXsBatch * nthreads) {
            cf xv[kXsBatch], gv[kXsBatch];
            int ii[kXsBatch];
#pragma unroll
            for (int u = 0; u < kXsBatch; ++u) {
                const int t = t0 + u * nthreads;
                const int tc = t < total ? t : total - 1;
                const int q1 = nq2 == 1 ? tc : (int)(((unsigned long long)(unsigned)tc * magic) >> 32);
                const int tq = tc - q1 * nq2;
                const int i = q1 + n1 * (q2a + tq) - start;
                ii[u] = (t < total && i >= 0 && i < lg) ? i : -1;
                xv[u] = base[(int64_t)q1 * w + (q2a - q0 + tq)];
                gv[u] = g2[tc];
            }
#pragma unroll
            for (int u = 0; u < kXsBatch; ++u)
                if (ii[u] >= 0) f(ii[u], c_mul(xv[u], gv[u]));
        }
    }
};

// Forward transform of a clip length with a prime factor above 7 (k_bluestein.hip): a chirp-z
// convolution of length n1 * n2, n2 = 6300.  Tables are [n1][n2] complex.
struct BzArgs {
    int n1, n2, n2pad;  // n2pad: row stride (floats) of the planar buffers, a multiple of 32
    int kmin, kmax;     // forward bins consumed
    int a;              // n1 = 16 a: the first transform's column stage splits into length-a and length-16 transforms
    int n_tiles1;       // row tiles of 32 (16 outputs k_a each) of the length-a stage
    int k1lo, k1n;      // rows k1lo .. k1lo + k1n - 1 of the second transform hold the consumed bins
    int n_tiles2;       // row tiles of their coefficient image apack2 (a multiple of nt2)
    int nt2;            // row tiles per wave of that stage: what the consumed rows need, at most 3
    const float *wp;    // chirp w[j], planar [2][L] (Re plane, Im plane), 0 from sample N on
    const cf *tl;       // T_L[q1 k2] at [q1 n2 + k2]
    const cf *bhat;     // DFT_L(conj chirp)[q1 + n1 q2] at [q1 n2 + q2]
    const cf *wk;       // w[k] / L  [kmax - kmin]
    const float *apack1; // coefficient image of the length-a DFT          [a][n_tiles1][64]
    const float *apack3; // ... of the length-16 stage with its twiddles, per k_a  [a][16][64]
    const float *apack2; // ... of the rows k1lo .. k1lo + k1n - 1   [n1][n_tiles2][64]
};

// One Bluestein size class (all bands whose chirp-z length is p).
struct CqClassDev {
    int p;              // transform length (power of two)
    int n_bands;        // bands in this class
    RadixList radix;
    const cf *tw;       // T_p   [p]
    CqTwiddles gtw;     // per-butterfly twiddle tables of the fused groups
    const cf *vrev;     // DFT_p(chirp) at digit-reversed positions [p]
    const int *band;    // band index j                 [n_bands]
    int len0, outer;    // p > 16384: `outer` radix-4 passes through global memory around blocks of len0 (k_cq_big.hip)
};

struct CqPlanDev {
    int kmin, nk;       // first forward bin, number of bins per clip
    int xn1, xw, xq0;   // layout of the forward bins (XsView); set per call
    int64_t xclip;      // elements per clip
    unsigned long long xmagic;
    HPFW_DEVICE_MEMBER XsView view(const cf *x, int clip) const { return XsView{x + (int64_t)clip * xclip, xn1, xw, xq0, xmagic}; }
    int c;              // spectrogram columns
    const int *start;   // slice start (absolute bin)   [121]
    const int *lg;      // window length                [121]
    const int64_t *g_off; // offset of band j in g     [121]
    const cf *g;        // window * chirp / (M P)       [sum lg]
    // the same windows in the order of the rows layout (XsBandRows): band j at g2_off[j], n1 * nq2[j] entries
    const cf *g2;
    const int64_t *g2_off; // [121]
    const int *q2a, *nq2;  // [121]
    const unsigned *nq2_magic; // [121]
    int rows_min;          // bands whose slice holds fewer than this many bins per row are walked element by element
};

// per-band constants of the window table (S5; plan.h cq_window_scale / cq_hann_den), by value into cq_window_kernel
struct CqWindowBands {
    double scale[121];
    int hann_den[121];
};
// the window table g [sum lg] and its rows-layout twin g2 (7-smooth lengths) generated on the device; c.lg, c.g_off (and
// c.start, c.q2a, c.nq2, c.g2_off, c.g for the twin) must be in place on the stream
void launch_cq_windows(const CqPlanDev &c, const CqWindowBands &b, int64_t big_m, int lg_max, cf *d_g, hipStream_t s);
void launch_cq_windows_rows(const CqPlanDev &c, int n1, int max_entries, cf *d_g2, hipStream_t s);

// S6 column stage: pcm [n_clips][n1][n2] (clips `clip_samples` apart) -> z [n_clips][hq][Re row, Im row][n2]
void launch_fwd_cols_q(const ColsQArgs &a, const int16_t *d_pcm, int64_t clip_samples, int n_clips, float *d_z, hipStream_t s);
// S6 row stage: z -> x [n_clips][n1][q2w] (XsView layout)
size_t fwd_rows_lds_bytes(const RowsArgs &a);
void launch_fwd_rows2(const RowsArgs &a, const Rows2Out &o, const float *d_z, int n_clips, cf *d_x, hipStream_t s);
// the bins [kmin, kmax) in natural order out of either layout (stage entry point)
void launch_gather_bins(const CqPlanDev &cp, const cf *d_x, int n_clips, cf *d_out, hipStream_t s);
// chirp-z forward transform (k_bluestein.hip): pcm -> G' -> H' -> x; the planar buffers hold bz_plane_bytes(bz, n_clips) each
size_t bz_plane_bytes(const BzArgs &bz, int n_clips);
// coefficient images of the first transform's two column stages (apack1, apack3), from T_n1 on the device
void launch_bz_pack_stages(const BzArgs &bz, const cf *d_tw_n1, float *d_apack1, float *d_apack3, hipStream_t s);
// coefficient image [n1][n_tiles][64] of rows k1_first .. k1_first + k1_count - 1 of the length-n1 DFT, from T_n1 on the device
void launch_bz_pack_coefficients(int n1, int k1_first, int k1_count, const cf *d_tw_n1, int n_tiles, float *d_apack, hipStream_t s);
// the tables bz.wp, bz.tl, bz.wk, bz.bhat of a clip length, generated on the device (d_b: 2 L floats, d_y: one planar buffer)
void launch_bz_make_tables(const RowsArgs &rows, const BzArgs &bz, int64_t n, float *d_b, float *d_y, hipStream_t s);
void launch_bz_cols_first(const BzArgs &bz, const int16_t *d_pcm, int64_t n, int n_clips, float *d_out, hipStream_t s);
void launch_bz_rows_both(const RowsArgs &rows, const BzArgs &bz, const float *d_in, int n_clips, float *d_out, hipStream_t s);
void launch_bz_cols_last(const BzArgs &bz, const float *d_in, int n_clips, cf *d_x, hipStream_t s);
// band chirp-z transforms: x -> mag [n_clips][121][c]; also the maxima each wave saw,
// d_wavemax [n_clips][121][kCqMaxWaves] (slots of absent waves are written as 0)
constexpr int kCqMaxWaves = 16;
// db_term_out: store t(m^2) = (float)(10 log10(max(m^2, 1e-10))) instead of the magnitude m
void launch_cq_class(const CqPlanDev &cp, const CqClassDev &cc, const cf *d_x, int n_clips,
                     float *d_mag, float *d_wavemax, bool db_term_out, hipStream_t s);
// the same for a class whose length exceeds the LDS (cc.outer > 0); d_work: cq_big_work_bytes(cc, n_clips)
size_t cq_big_work_bytes(const CqClassDev &cc, int n_clips);
void launch_cq_big_class(const CqPlanDev &cp, const CqClassDev &cc, const cf *d_x, int n_clips, cf *d_work, float *d_mag,
                         float *d_wavemax, bool db_term_out, hipStream_t s);
// d_clipmax [n_clips] = the largest of each clip's wave maxima
void launch_clipmax(const float *d_wavemax, float *d_clipmax, int n_clips, hipStream_t s);
// dB terms -> dB spectrogram in place: S = max(t - t_max, -80)
void launch_db_finish(float *d_t, const float *d_clipmax, int n_clips, int64_t per_clip, hipStream_t s);
// the same maxima when the chirp-z stage did not run (stage entry point)
void launch_magmax(const float *d_mag, int n_clips, int c, float *d_wavemax, hipStream_t s);
// amplitude_to_db: d_clipmax [n_clips] receives the per-clip maximum of d_wavemax first
void launch_db(const float *d_mag, const float *d_wavemax, float *d_clipmax, int n_clips, int64_t per_clip,
               float *d_db, hipStream_t s);
// filters * frames on f32 MFMA: s_db [n_clips][121][c] -> proj [n_clips][64][c-19]
// d_fpack: filters repacked by pack_filters_for_mfma().  d_tmax != NULL: s_db holds dB terms and
// S = max(t - d_tmax[clip], -80) is applied while the slab is staged.
void launch_project(const float *d_fpack, const float *d_db, const float *d_tmax, int n_clips, int c,
                    float *d_proj, hipStream_t s);
// delta over 80 frames, sign, bit pack: proj [n_clips][64][nf] -> hp [n_clips][nf-80]
void launch_pack(const float *d_proj, int n_clips, int nf, uint64_t *d_hp, hipStream_t s);
// a4's reference level + a5..a8 in fixed point (k_project_q.hip): filter digits image; dB terms (d_tmax != NULL)
// or dB spectrograms -> hashprints in one kernel, the lag-80 difference taken on the quantised spectrogram (exact integers).
// d_dbg: NULL, or (tests) the integer sums D [n_clips][64][c - 99]
size_t project_q_image_bytes();
void pack_filters_q(const float *f_colmajor, std::vector<int8_t> &image);
void launch_hashprints_q(const void *d_fq_image, const float *d_db, const float *d_tmax, int n_clips, int c, uint64_t *d_hp,
                         long long *d_dbg, hipStream_t s);

// ---- HashprintHandle with other template arguments (k_hashprint_cfg.hip) ----------------------------
struct CfgArgs {
    int rows, context, lag, filters; // spectrogram rows, FramesContext, T, 8 sizeof(N)
    const float *fpack;              // filters as the MFMA operand image (pack_cfg_filters)
};
size_t cfg_fpack_floats(int rows, int context, int filters);
void pack_cfg_filters(int rows, int context, int filters, const float *f_colmajor, float *fpack);
size_t project_cfg_lds_bytes(const CfgArgs &a);
// d_s [n_clips][rows][stride], d_cols [n_clips] valid columns (NULL: stride) -> d_proj [n_clips][filters][proj_stride]
void launch_project_cfg(const CfgArgs &a, const float *d_s, const int *d_cols, int n_clips, int64_t stride, float *d_proj,
                        int64_t proj_stride, hipStream_t s);
// -> d_hp [n_clips][hp_stride] of uint16 / uint32 / uint64 (by a.filters)
void launch_pack_cfg(const CfgArgs &a, const float *d_proj, const int *d_cols, int n_clips, int64_t stride, int64_t proj_stride,
                     void *d_hp, int64_t hp_stride, hipStream_t s);

// calc_cov + accumulate for such a configuration (k_cov_cfg.hip): accum [kt][kt], kt = rows * context, both triangles
bool cov_cfg_supported(const CfgArgs &a);
int cov_cfg_tile_count(int kt);
void cov_cfg_tile_list(int kt, int *xy /* 2 * cov_cfg_tile_count(kt) */);
size_t cov_cfg_workspace_bytes(const CfgArgs &a, int n_clips);
void launch_cov_cfg(const CfgArgs &a, const float *d_s, const int *d_cols, int n_clips, int64_t stride, const int *d_tiles,
                    float *d_ws, float *d_accum, hipStream_t s);

// ---- filter learning (index() only) -------------------------------------------------------
// accum [2420][2420] (tiles on or above the diagonal) += sum over clips of centred^T centred / (nf - 1),
// by lag correlations (k_cov.hip); d_ws: cov_workspace_bytes(n_clips, c) of scratch
int cov_tile_count();
void cov_tile_list(int *xy /* 2 * cov_tile_count() */);
size_t cov_workspace_bytes(int n_clips, int c);
void launch_cov(const float *d_db, int n_clips, int c, const int *d_tiles, float *d_ws, float *d_accum, hipStream_t s);
// host: unit eigenvectors of the m largest eigenvalues of a symmetric n x n matrix (eigen_host.cpp)
int top_eigenvectors(const float *cov, int n, int m, float *out, double *evals);

// host helper: [kp][lane][2] operand image of the filters for v_mfma_f32_32x32x2_f32
void pack_filters_for_mfma(const float *filters_colmajor, float *fpack /* 64*2420 */);

// ---- Mel front-end (f3, k_mel.hip) ----------------------------------------------------------
constexpr int kMelFrame = 4410, kMelHop = 441, kMelBins = 2206, kMelBands = 33;
int mel_frames(int64_t n_samples);
size_t mel_work_bytes(int64_t n_samples, int n_clips);
// d_out [n_clips][33][frames] (kept columns at the front of every row), d_count [n_clips] their number
void launch_mel(const RowsArgs &rows, const float *d_win, const float *d_cpack, const int16_t *d_pcm, int64_t n, int n_clips,
                int64_t *d_blk, int *d_pos, int *d_count, float *d_pmax, float *d_work, float *d_out, hipStream_t s);

// ---- search ------------------------------------------------------------------------------
struct SearchArgs {
    const uint64_t *db;      // concatenated hashprints
    const int64_t *db_off;   // [n_clips + 1] (device)
    int n_clips;
    const uint64_t *q;       // concatenated queries (device)
    const int64_t *q_off;    // [n_q + 1] (device)
    int n_q;
    int k_max;               // longest query
    uint64_t *best;          // [n_q][n_clips]: (dist << 32) | offset
};
void launch_hamming_scan(const SearchArgs &a, hipStream_t s);
// the same scan on the matrix cores (k_search_mfma.hip): queries expanded to fp4 +-1 first
int hamming_mfma_kt_pad(int k_max);            // rows of the expanded-query image per group of 32
size_t hamming_mfma_lds_bytes(int k_max);      // dynamic LDS of the scan; > 160 KB: use launch_hamming_scan
void launch_expand_queries(const uint64_t *d_q, const int64_t *d_q_off, int n_q, int kt_pad, void *d_qa, hipStream_t s);
// one query against the whole index with the tile rows = 32 shifts of the query (few queries: no padded rows)
size_t hamming_shift_lds_bytes(int k);
size_t hamming_shift_image_bytes(int k);       // scratch for the query's expanded image (d_qexp)
void launch_hamming_shift(const uint64_t *d_db, const int64_t *d_db_off, int n_clips, int n_off_max, const uint64_t *d_q,
                          int k, void *d_qexp, uint64_t *d_best, hipStream_t s);
// nearest windows (AnnStorage semantics with exact neighbours): rows = windows of `win` hashprints
void launch_expand_windows(const uint64_t *d_q, const int64_t *d_w_start, int n_win, int win, int kt_pad, void *d_qa,
                           hipStream_t s);
void launch_knn_windows(const uint64_t *d_db, const int64_t *d_db_off, int n_clips, int n_off_max, const void *d_qa,
                        int kt_pad, int n_win, int win, int nn, void *d_slots /* [n_win][8] u64, preset to ~0 */,
                        hipStream_t s);
// d_gk: per group of 32 queries {longest, shortest non-empty}; n_off_max: most offsets any (query, clip) has
void launch_hamming_mfma(const SearchArgs &a, const void *d_qa, int kt_pad, const int *d_gk, int n_off_max, hipStream_t s);
void launch_topk(const uint64_t *d_best, int n_q, int n_clips, int k, uint32_t clip_base, void *d_out,
                 hipStream_t s);
// the same result in two steps (64 slices of the clips per query, then a merge): large indexes, few queries
size_t topk_scratch_bytes(int n_q, int k);
void launch_topk_two_step(const uint64_t *d_best, int n_q, int n_clips, int k, uint32_t clip_base, void *d_scratch,
                          void *d_out, hipStream_t s);

// fail-loud launch check
const char *last_launch_error();

} // namespace hpfw
