// k_cov.hip -- a11 (index() only): the covariance of the context frames, accumulated over clips.
//
// Replaces HashprintHandle::calc_cov + the accumulation under a mutex (reference
// include/hpfw/core/hashprint_handle.h:96-102, include/hpfw/core/parallel_collector.h:93-97):
// per file, frames^T [n_frames x 2420] is centred column-wise (its own mean over frames) and
// cov = centred^T centred / (n_frames - 1) is added to accum_cov.  28 GFLOP per 30 s clip -- the
// largest FLOP item of the whole product -- so it runs on v_mfma_f32_32x32x2_f32 with the same
// implicit im2col as the projection: frames[b*20 + t, n] = S[b, n + t] is read from an LDS slab of
// S, never materialised.  Only tiles on or above the diagonal are computed (190 of 361); the host
// mirrors the result.  Accuracy bar: tolerance against a float64 numpy evaluation (the reference's
// own result depends on MKL's summation order).
#include "kernels.h"

#include <algorithm>

namespace hpfw {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// mu[clip][b*20 + t] = mean over n < nf of S[b][n + t]
__global__ __launch_bounds__(256) void frame_mean_kernel(const float *__restrict__ sdb, int c, int nf,
                                                         float *__restrict__ mu)
{
    __shared__ double part[4];
    const int b = blockIdx.x, clip = blockIdx.y, tid = threadIdx.x;
    const float *row = sdb + ((int64_t)clip * kBins + b) * c;
    double s = 0.0;
    for (int i = tid; i < c; i += 256) s += (double)row[i];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    if ((tid & 63) == 0) part[tid >> 6] = s;
    __syncthreads();
    if (tid < kCtx) {
        double tot = part[0] + part[1] + part[2] + part[3];
        for (int i = 0; i < tid; ++i) tot -= (double)row[i];          // columns before the window
        for (int i = tid + nf; i < c; ++i) tot -= (double)row[i];     // columns after it
        mu[(int64_t)clip * kFrame + b * kCtx + tid] = (float)(tot / (double)nf);
    }
}

constexpr int kCvTile = 128;                // rows (and columns) of the covariance per workgroup
constexpr int kCvFrames = 256;              // frames per staged chunk
constexpr int kCvRow = kCvFrames + kCtx;    // LDS slab row stride (276 floats)
constexpr int kCvBins = 8;                  // bins a 128-wide range of k = b*20 + t can touch
constexpr int kCvSplits = 16;               // work splits per tile pair (190 x 16 workgroups)

// One workgroup = one 128 x 128 tile pair x one split of the work items (item = clip x 256-frame
// chunk): part[split][tile][128][128] = sum over the split's items of
// (X[ka][n] - mu[ka]) (X[kb][n] - mu[kb]); wave (wr, wc) of the 2 x 2 wave grid owns 64 x 64.
// Splitting is what fills the chip: 190 tile pairs alone would leave a quarter of the CUs idle and
// the rest with one wave per SIMD.  The partials are summed in split order by cov_reduce_kernel,
// so the result does not depend on scheduling.
__global__ __launch_bounds__(256) void cov_kernel(const float *__restrict__ sdb, const float *__restrict__ mu,
                                                  int c, int nf, int n_chunks, int n_items, int items_per_split,
                                                  const int2 *__restrict__ tiles, float *__restrict__ part)
{
    __shared__ float slab[2][kCvBins * kCvRow];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1, hb = lane >> 5, li = lane & 31;
    const int2 tile = tiles[blockIdx.x];
    const int k0[2] = {tile.x * kCvTile, tile.y * kCvTile};
    const int bin0[2] = {k0[0] / kCtx, k0[1] / kCtx};
    const bool same = tile.x == tile.y;
    int krow[2][2], off[2][2]; // [operand A/B][32-row tile]
    bool valid[2][2];
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int k = k0[o] + (o == 0 ? wr : wc) * 64 + t * 32 + li;
            valid[o][t] = k < kFrame;
            const int kk = valid[o][t] ? k : kFrame - 1;
            krow[o][t] = kk;
            off[o][t] = (kk / kCtx - bin0[o]) * kCvRow + kk % kCtx;
        }
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f32x16{0};

    const int item0 = blockIdx.y * items_per_split, item1 = min(n_items, item0 + items_per_split);
    int cur_clip = -1;
    float m[2][2] = {{0.0f, 0.0f}, {0.0f, 0.0f}};
    for (int item = item0; item < item1; ++item) {
        const int clip = item / n_chunks, n0 = (item - clip * n_chunks) * kCvFrames;
        const float *S = sdb + (int64_t)clip * kBins * c;
        if (clip != cur_clip) {
            cur_clip = clip;
#pragma unroll
            for (int o = 0; o < 2; ++o)
#pragma unroll
                for (int t = 0; t < 2; ++t) m[o][t] = mu[(int64_t)clip * kFrame + krow[o][t]];
        }
        const int nfr = min(kCvFrames, nf - n0);
        __syncthreads();
        for (int o = 0; o < (same ? 1 : 2); ++o)
            for (int i = tid; i < kCvBins * (kCvFrames + kCtx - 1); i += 256) {
                const int bl = i / (kCvFrames + kCtx - 1), col = i - bl * (kCvFrames + kCtx - 1);
                const int b = bin0[o] + bl, gc = n0 + col;
                slab[o][bl * kCvRow + col] = (b < kBins && gc < c) ? S[(int64_t)b * c + gc] : 0.0f;
            }
        __syncthreads();
        const float *sa = slab[0], *sb = slab[same ? 0 : 1];
        for (int st = 0; st < (nfr + 1) / 2; ++st) {
            const int n = 2 * st + hb;
            const bool in = n < nfr;
            float av[2], bv[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                av[t] = (in && valid[0][t]) ? sa[off[0][t] + n] - m[0][t] : 0.0f;
                bv[t] = (in && valid[1][t]) ? sb[off[1][t] + n] - m[1][t] : 0.0f;
            }
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0], bv[0], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0], bv[1], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1], bv[0], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1], bv[1], acc[1][1], 0, 0, 0);
        }
    }
    // D layout: column = lane & 31 (operand B row), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    float *out = part + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (kCvTile * kCvTile);
#pragma unroll
    for (int ta = 0; ta < 2; ++ta)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int ra = wr * 64 + ta * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * hb;
                out[ra * kCvTile + wc * 64 + tb * 32 + li] = acc[ta][tb][reg];
            }
}

// accum[ka][kb] += scale * (part[0] + part[1] + ... in split order)
__global__ __launch_bounds__(256) void cov_reduce_kernel(const float *__restrict__ part, int n_splits, float scale,
                                                         const int2 *__restrict__ tiles, float *__restrict__ accum)
{
    const int2 tile = tiles[blockIdx.x];
    const int e = blockIdx.y * 256 + threadIdx.x;
    const int ka = tile.x * kCvTile + e / kCvTile, kb = tile.y * kCvTile + e % kCvTile;
    float s = 0.0f;
    for (int g = 0; g < n_splits; ++g)
        s += part[((int64_t)g * gridDim.x + blockIdx.x) * (kCvTile * kCvTile) + e];
    if (ka < kFrame && kb < kFrame) accum[(int64_t)ka * kFrame + kb] += scale * s;
}

void launch_frame_mean(const float *d_db, int n_clips, int c, float *d_mu, hipStream_t s)
{
    hipLaunchKernelGGL(frame_mean_kernel, dim3(kBins, n_clips), dim3(256), 0, s, d_db, c, c - (kCtx - 1), d_mu);
}

int cov_tile_count()
{
    const int nt = (kFrame + kCvTile - 1) / kCvTile;
    return nt * (nt + 1) / 2;
}

void cov_tile_list(int *xy)
{
    const int nt = (kFrame + kCvTile - 1) / kCvTile;
    int w = 0;
    for (int i = 0; i < nt; ++i)
        for (int j = i; j < nt; ++j) {
            xy[2 * w] = i;
            xy[2 * w + 1] = j;
            ++w;
        }
}

int cov_splits(int n_clips, int c)
{
    const int nf = c - (kCtx - 1), n_chunks = (nf + kCvFrames - 1) / kCvFrames;
    return (int)std::min<int64_t>(kCvSplits, (int64_t)n_clips * n_chunks);
}

size_t cov_part_bytes(int n_clips, int c)
{
    return (size_t)cov_splits(n_clips, c) * cov_tile_count() * kCvTile * kCvTile * sizeof(float);
}

void launch_cov(const float *d_db, const float *d_mu, int n_clips, int c, const int *d_tiles, float *d_part,
                float *d_accum, hipStream_t s)
{
    const int nf = c - (kCtx - 1), n_chunks = (nf + kCvFrames - 1) / kCvFrames;
    const int n_items = n_clips * n_chunks, n_splits = cov_splits(n_clips, c);
    const int per = (n_items + n_splits - 1) / n_splits;
    const float scale = 1.0f / (float)(nf - 1); // hashprint_handle.h:101: / (rows - 1)
    const int2 *tiles = reinterpret_cast<const int2 *>(d_tiles);
    hipLaunchKernelGGL(cov_kernel, dim3(cov_tile_count(), n_splits), dim3(256), 0, s, d_db, d_mu, c, nf, n_chunks,
                       n_items, per, tiles, d_part);
    hipLaunchKernelGGL(cov_reduce_kernel, dim3(cov_tile_count(), kCvTile * kCvTile / 256), dim3(256), 0, s, d_part,
                       n_splits, scale, tiles, d_accum);
}

} // namespace hpfw
