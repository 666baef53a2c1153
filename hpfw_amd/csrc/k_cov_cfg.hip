// k_cov_cfg.hip -- HashprintHandle::calc_cov (reference include/hpfw/core/hashprint_handle.h:96-102) and
// ParallelCollector::preprocess' `accum_cov += cov` (include/hpfw/core/parallel_collector.h:93-97) for
// HashprintHandle template arguments other than the live-id default, whose covariance k_cov.hip computes by lag
// correlations specialised for 121 x 20: here, for any (rows, context) -- the combiner's 33 x 32 = 1056 first of all --
// the plain product
//     cov = centred^T centred / (n_frames - 1),  centred[n][k] = X[k][n] - mean_n X[k][n],  X[k][n] = S[row][n + t]
// with k = row * context + t, as G = X X^T on f32 MFMA with BOTH operands taken by implicit im2col from an LDS slab of
// the spectrogram (the frames are never materialised), and  cov = (G - s s^T / n_frames) / (n_frames - 1),
// s[k] = sum_n X[k][n].  The spectrogram is centred on its row means while it is staged, so that G and s s^T / n are
// small numbers of the size of the variances and their difference loses nothing in float.
//   cov_cfg_rowmean_kernel   mean of every spectrogram row over its valid columns
//   cov_cfg_sums_kernel      s[clip][k]
//   cov_cfg_tiles_kernel     workgroup = one 128 x 128 tile on or above the diagonal x one group of clips: per clip the
//                            tile of G on v_mfma_f32_32x32x2_f32 (wave = 64 x 64), corrected and scaled, added to the
//                            group's running tile; partial tiles per group
//   cov_cfg_reduce_kernel    accum += the groups' tiles in group order (deterministic), mirrored below the diagonal
#include "kernels.h"

namespace hpfw {

extern __shared__ __align__(16) unsigned char smem_raw[];

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kCcTile = 128;   // tile of the covariance per workgroup
constexpr int kCcChunk = 128;  // frames staged per slab
constexpr int kCcMaxRows = 16; // spectrogram rows a 128-wide block of k can span (context >= 9)

__global__ __launch_bounds__(256) void cov_cfg_rowmean_kernel(CfgArgs a, const float *__restrict__ s, const int *__restrict__ cols,
                                                              int64_t stride, float *__restrict__ mean)
{
    __shared__ float part[4];
    const int row = blockIdx.x, clip = blockIdx.y;
    const int c = cols ? cols[clip] : (int)stride;
    const float *p = s + ((int64_t)clip * a.rows + row) * stride;
    float acc = 0.0f;
    for (int i = threadIdx.x; i < c; i += 256) acc += p[i];
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) acc += __shfl_xor(acc, sft);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) mean[(int64_t)clip * a.rows + row] = c > 0 ? ((part[0] + part[1]) + (part[2] + part[3])) / (float)c : 0.0f;
}

// s[clip][k] = sum over the frames n of (S[row][n + t] - mean[row]); one thread per k
__global__ __launch_bounds__(256) void cov_cfg_sums_kernel(CfgArgs a, const float *__restrict__ s, const int *__restrict__ cols,
                                                           int64_t stride, const float *__restrict__ mean, float *__restrict__ sums)
{
    const int k = blockIdx.x * 256 + threadIdx.x, clip = blockIdx.y;
    const int kt = a.rows * a.context;
    if (k >= kt) return;
    const int c = cols ? cols[clip] : (int)stride;
    const int nf = c - a.context + 1;
    const int row = k / a.context, t = k - row * a.context;
    const float *p = s + ((int64_t)clip * a.rows + row) * stride + t;
    const float m = mean[(int64_t)clip * a.rows + row];
    float acc = 0.0f;
    for (int n = 0; n < nf; ++n) acc += p[n] - m;
    sums[(int64_t)clip * kt + k] = acc;
}

// tile (bi, bj), bi <= bj, of one group of clips [c0, c1): partial[group][tile][128][128]
__global__ __launch_bounds__(256) void cov_cfg_tiles_kernel(CfgArgs a, const float *__restrict__ s, const int *__restrict__ cols,
                                                            int64_t stride, const float *__restrict__ mean,
                                                            const float *__restrict__ sums, const int *__restrict__ tiles,
                                                            int n_clips, int clips_per_group, float *__restrict__ partial)
{
    float *slab = reinterpret_cast<float *>(smem_raw); // [2][kCcMaxRows][slab_w]: the rows of the A block, then of the B block
    const int slab_w = kCcChunk + a.context - 1;
    const int tile = blockIdx.x, group = blockIdx.y;
    const int bi = tiles[2 * tile], bj = tiles[2 * tile + 1];
    const int kt = a.rows * a.context;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, part = lane >> 5;
    const int wi = wave >> 1, wj = wave & 1; // this wave's 64 x 64 quarter of the tile
    // first spectrogram row of each block
    const int ra0 = (bi * kCcTile) / a.context, rb0 = (bj * kCcTile) / a.context;
    // per lane: where its two A rows (k = bi*128 + wi*64 + {0, 32} + j) and two B columns sit in the slab
    int a_off[2], b_off[2];
    bool a_ok[2], b_ok[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int ka = bi * kCcTile + wi * 64 + h * 32 + j, kb = bj * kCcTile + wj * 64 + h * 32 + j;
        a_ok[h] = ka < kt;
        b_ok[h] = kb < kt;
        const int rowa = ka / a.context, rowb = kb / a.context;
        a_off[h] = (rowa - ra0) * slab_w + (ka - rowa * a.context);
        b_off[h] = (kCcMaxRows + rowb - rb0) * slab_w + (kb - rowb * a.context);
    }
    const int rows_a = min(a.rows - ra0, kCcMaxRows), rows_b = min(a.rows - rb0, kCcMaxRows);
    f32x16 total[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) total[x][y] = f32x16{0};
    const int c0 = group * clips_per_group, c1 = min(n_clips, c0 + clips_per_group);
    for (int clip = c0; clip < c1; ++clip) {
        const int c = cols ? cols[clip] : (int)stride;
        const int nf = c - a.context + 1;
        if (nf < 2) continue; // calc_cov divides by rows() - 1: no covariance from fewer than two frames
        const float *S = s + (int64_t)clip * a.rows * stride;
        const float *mu = mean + (int64_t)clip * a.rows;
        f32x16 acc[2][2];
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y) acc[x][y] = f32x16{0};
        for (int n0 = 0; n0 < nf; n0 += kCcChunk) {
            __syncthreads();
            for (int i = threadIdx.x; i < (rows_a + rows_b) * slab_w; i += 256) {
                const int rr = i / slab_w, col = i - rr * slab_w;
                const bool isb = rr >= rows_a;
                const int row = isb ? rb0 + rr - rows_a : ra0 + rr;
                const int dst = (isb ? kCcMaxRows + rr - rows_a : rr) * slab_w + col;
                slab[dst] = (n0 + col < c) ? S[(int64_t)row * stride + n0 + col] - mu[row] : 0.0f;
            }
            __syncthreads();
            const int steps = min(kCcChunk, nf - n0);
            for (int q = 0; q < steps; q += 2) {
                const int nn = q + part;                 // this lane's frame within the chunk (k index of the MFMA)
                const bool live = n0 + nn < nf;          // frames beyond the last one contribute nothing
                float av[2], bv[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    av[h] = (live && a_ok[h]) ? slab[a_off[h] + nn] : 0.0f;
                    bv[h] = b_ok[h] ? slab[b_off[h] + nn] : 0.0f;
                }
#pragma unroll
                for (int x = 0; x < 2; ++x)
#pragma unroll
                    for (int y = 0; y < 2; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[x], bv[y], acc[x][y], 0, 0, 0);
            }
        }
        // cov of this clip = (G - s_i s_j / nf) / (nf - 1), added to the group's running tile
        const float *sk = sums + (int64_t)clip * kt;
        const float inv_nf = 1.0f / (float)nf, inv_n1 = 1.0f / (float)(nf - 1);
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y) {
                const int kj = bj * kCcTile + wj * 64 + y * 32 + j;
                const float sj = kj < kt ? sk[kj] : 0.0f;
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int ki = bi * kCcTile + wi * 64 + x * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * part;
                    const float si = ki < kt ? sk[ki] : 0.0f;
                    total[x][y][reg] += (acc[x][y][reg] - si * sj * inv_nf) * inv_n1;
                }
            }
    }
    float *out = partial + ((int64_t)group * gridDim.x + tile) * kCcTile * kCcTile;
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int li = wi * 64 + x * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * part, lj = wj * 64 + y * 32 + j;
                out[li * kCcTile + lj] = total[x][y][reg];
            }
}

// accum[k][k'] += sum over the groups (in order) of the tile's partials; both triangles written
__global__ __launch_bounds__(256) void cov_cfg_reduce_kernel(int kt, const int *__restrict__ tiles, int n_tiles, int n_groups,
                                                             const float *__restrict__ partial, float *__restrict__ accum)
{
    const int tile = blockIdx.x;
    const int bi = tiles[2 * tile], bj = tiles[2 * tile + 1];
    for (int e = threadIdx.x; e < kCcTile * kCcTile; e += 256) {
        const int li = e / kCcTile, lj = e - li * kCcTile;
        const int ki = bi * kCcTile + li, kj = bj * kCcTile + lj;
        if (ki >= kt || kj >= kt || ki > kj) continue; // the upper triangle of a diagonal tile only
        float v = 0.0f;
        for (int g = 0; g < n_groups; ++g) v += partial[((int64_t)g * n_tiles + tile) * kCcTile * kCcTile + e];
        const float nv = accum[(int64_t)ki * kt + kj] + v;
        accum[(int64_t)ki * kt + kj] = nv;
        accum[(int64_t)kj * kt + ki] = nv;
    }
}

int cov_cfg_tile_count(int kt)
{
    const int nb = (kt + kCcTile - 1) / kCcTile;
    return nb * (nb + 1) / 2;
}

void cov_cfg_tile_list(int kt, int *xy)
{
    const int nb = (kt + kCcTile - 1) / kCcTile;
    int w = 0;
    for (int i = 0; i < nb; ++i)
        for (int j = i; j < nb; ++j) {
            xy[2 * w] = i;
            xy[2 * w + 1] = j;
            ++w;
        }
}

bool cov_cfg_supported(const CfgArgs &a) { return (kCcTile + a.context - 2) / a.context + 1 <= kCcMaxRows; }

int cov_cfg_groups(int n_clips) { return n_clips < 16 ? (n_clips < 1 ? 1 : n_clips) : 16; }

size_t cov_cfg_workspace_bytes(const CfgArgs &a, int n_clips)
{
    const int kt = a.rows * a.context;
    return ((size_t)n_clips * a.rows + (size_t)n_clips * kt + (size_t)cov_cfg_groups(n_clips) * cov_cfg_tile_count(kt) * kCcTile * kCcTile) * sizeof(float);
}

// d_ws: cov_cfg_workspace_bytes; d_tiles: cov_cfg_tile_list on the device; d_accum [kt][kt]
void launch_cov_cfg(const CfgArgs &a, const float *d_s, const int *d_cols, int n_clips, int64_t stride, const int *d_tiles,
                    float *d_ws, float *d_accum, hipStream_t s)
{
    if (n_clips <= 0) return;
    const int kt = a.rows * a.context, n_tiles = cov_cfg_tile_count(kt), groups = cov_cfg_groups(n_clips);
    const int per_group = (n_clips + groups - 1) / groups;
    float *mean = d_ws, *sums = mean + (size_t)n_clips * a.rows, *partial = sums + (size_t)n_clips * kt;
    hipLaunchKernelGGL(cov_cfg_rowmean_kernel, dim3(a.rows, n_clips), dim3(256), 0, s, a, d_s, d_cols, stride, mean);
    hipLaunchKernelGGL(cov_cfg_sums_kernel, dim3((kt + 255) / 256, n_clips), dim3(256), 0, s, a, d_s, d_cols, stride, mean, sums);
    const size_t lds = (size_t)2 * kCcMaxRows * (kCcChunk + a.context - 1) * sizeof(float);
    hipLaunchKernelGGL(cov_cfg_tiles_kernel, dim3(n_tiles, groups), dim3(256), lds, s, a, d_s, d_cols, stride, mean, sums, d_tiles,
                       n_clips, per_group, partial);
    hipLaunchKernelGGL(cov_cfg_reduce_kernel, dim3(n_tiles), dim3(256), 0, s, kt, d_tiles, n_tiles, groups, partial, d_accum);
}

} // namespace hpfw
