// k_cov.hip -- a11 (index() only): the covariance of the context frames, accumulated over clips.
//
// Replaces HashprintHandle::calc_cov + the accumulation under a mutex (reference
// include/hpfw/core/hashprint_handle.h:96-102, include/hpfw/core/parallel_collector.h:93-97):
// per file, frames^T [n_frames x 2420] is centred column-wise (its own mean over frames) and
// cov = centred^T centred / (n_frames - 1) is added to accum_cov: 28 GFLOP per 30 s clip as a dense
// product, the largest FLOP item of the whole reference.
//
// The frames are windows of one spectrogram, frames[b*20 + t, n] = S[b, n + t], so the product is a
// set of lag correlations and does not need 2420 x 2420 x n_frames multiply-adds.  With Z = S minus
// each bin's mean over the clip, d = t - t' >= 0 and U the number of columns:
//   sum_n frames[(b,t), n] frames[(b',t'), n]
//     = G[(b,d), b']                      G[(b,d), b'] = sum_u Z[b, u + d] Z[b', u]   (all u in range)
//     - sum_{j=1..min(t,t')} Z[b, t - j] Z[b', t' - j]                                 (columns before the window)
//     - sum_{j=0..18-max(t,t')} Z[b, nf + t + j] Z[b', nf + t' + j]                    (columns after it)
// and the centring on the window means adds - nf delta[(b,t)] delta[(b',t')].  The three corrections
// are 39 rank-one terms: V V^T with V [2420 x 39] per clip.  So per clip
//   lag_corr_kernel   G = [2420 x U] . [U x 121]     1.4 GFLOP on v_mfma_f32_32x32x2_f32 (implicit im2col)
//   vvt_kernel        V V^T, tiles on or above the diagonal   0.23 GFLOP on the same instruction
// instead of 14.9 GFLOP, and one expansion G -> [2420 x 2420] per call instead of per clip.  The
// result differs from the dense product only by rounding (tolerance-tested against float64; the
// reference's own bits depend on MKL's summation order).  Partial sums are kept per split of the
// work and reduced in a fixed order, so the result does not depend on scheduling.
#include "kernels.h"

#include <algorithm>

namespace hpfw {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kCvTile = 128;                 // rows (and columns) of an output tile
constexpr int kCvVec = 40;                   // correction vectors per clip (39 used)
constexpr int kLgRowsPad = 19 * kCvTile;     // 2432 >= 2420 rows (b, d) of G
constexpr int kLgCols = 128;                 // >= 121 bins
constexpr int kLgChunk = 64;                 // spectrogram columns per staged chunk
constexpr int kLgRowA = kLgChunk + kCtx;     // 84: slab row stride of the shifted operand
constexpr int kLgRowB = kLgChunk + 1;        // 65: slab row stride of the plain operand (odd: no bank conflicts)
constexpr int kLgBinsA = 8;                  // bins a 128-wide range of rows (b, d) can touch
constexpr int kLgSplits = 53;                // 19 x 53 = 1007 workgroups: four per CU
constexpr int kVvSplits = 8;

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// One workgroup per (bin, clip): Z = S - mean over the clip's columns (in a workspace), and this bin's 20
// rows of the clip's correction vectors V [40][2420]:
//   v 0..18  (j = 1..19):  Z[b, t - j]        for t >= j,        else 0
//   v 19..37 (j = 0..18):  Z[b, nf + t + j]   for t + j <= 18,   else 0
//   v 38:                  sqrt(nf) * (mean of Z over the window of t)
__global__ __launch_bounds__(256) void lag_prep_kernel(const float *__restrict__ sdb, int c, int nf,
                                                       float *__restrict__ z, float *__restrict__ v)
{
    __shared__ double part[4];
    __shared__ float edge[2 * kCtx]; // Z[0..19] and Z[nf - 1 .. nf + 18]
    const int b = blockIdx.x, clip = blockIdx.y, tid = threadIdx.x;
    const float *row = sdb + ((int64_t)clip * kBins + b) * c;
    float *zrow = z + ((int64_t)clip * kBins + b) * c;
    double s = 0.0;
    for (int i = tid; i < c; i += 256) s += (double)row[i];
    s = wave_sum(s);
    if ((tid & 63) == 0) part[tid >> 6] = s;
    __syncthreads();
    const float m = (float)((part[0] + part[1] + part[2] + part[3]) / (double)c);
    __syncthreads();
    double zs = 0.0;
    for (int i = tid; i < c; i += 256) {
        const float zz = row[i] - m;
        zrow[i] = zz;
        zs += (double)zz;
        if (i < kCtx) edge[i] = zz;
        if (i >= nf - 1) edge[kCtx + i - (nf - 1)] = zz;
    }
    zs = wave_sum(zs);
    if ((tid & 63) == 0) part[tid >> 6] = zs;
    __syncthreads();
    const double total = part[0] + part[1] + part[2] + part[3];
    float *vc = v + (int64_t)clip * kCvVec * kFrame + b * kCtx;
    if (tid < kCtx) {
        const int t = tid;
        double win = total;
        for (int i = 0; i < t; ++i) win -= (double)edge[i];                         // columns before the window
        for (int i = t + nf; i < c; ++i) win -= (double)edge[kCtx + i - (nf - 1)];  // columns after it
        vc[(int64_t)38 * kFrame + t] = (float)(sqrt((double)nf) * (win / (double)nf));
        vc[(int64_t)39 * kFrame + t] = 0.0f;
        for (int j = 1; j <= 19; ++j) vc[(int64_t)(j - 1) * kFrame + t] = t >= j ? edge[t - j] : 0.0f;
        for (int j = 0; j <= 18; ++j) vc[(int64_t)(19 + j) * kFrame + t] = t + j <= 18 ? edge[kCtx + 1 + t + j] : 0.0f;
    }
}

// part[split][row (b,d)][bin b'] = sum over the split's (clip, 64-column chunk) items of Z[b, u + d] Z[b', u].
// Workgroup = 128 rows x 128 bins; wave (wr, wc) of the 2 x 2 wave grid owns 64 x 64.
__global__ __launch_bounds__(256) void lag_corr_kernel(const float *__restrict__ z, int c, int n_chunks, int n_items,
                                                       int items_per_split, float *__restrict__ part)
{
    __shared__ float slab_a[kLgBinsA * kLgRowA];
    __shared__ float slab_b[kLgCols * kLgRowB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1, hb = lane >> 5, li = lane & 31;
    const int r0 = blockIdx.x * kCvTile, bin0 = r0 / kCtx;
    int off_a[2], off_b[2];
    bool ok_a[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int r = r0 + wr * 64 + t * 32 + li;
        ok_a[t] = r < kFrame;
        const int rr = ok_a[t] ? r : kFrame - 1;
        off_a[t] = (rr / kCtx - bin0) * kLgRowA + rr % kCtx;
        off_b[t] = (wc * 64 + t * 32 + li) * kLgRowB;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f32x16{0};
    const int item0 = blockIdx.y * items_per_split, item1 = min(n_items, item0 + items_per_split);
    // the two slabs of an item go global -> registers while the previous item multiplies, registers ->
    // LDS between two barriers: the loads' latency hides behind the MFMAs
    constexpr int kNa = (kLgBinsA * (kLgChunk + kCtx - 1) + 255) / 256; // 3 words of slab A per thread
    constexpr int kNb = kLgCols * kLgChunk / 256;                       // 32 words of slab B per thread
    float ra[kNa], rb[kNb];
    auto stage_load = [&](int item) {
        const int clip = item / n_chunks, u0 = (item - clip * n_chunks) * kLgChunk;
        const float *Z = z + (int64_t)clip * kBins * c;
#pragma unroll
        for (int k = 0; k < kNa; ++k) {
            const int i = tid + k * 256;
            const int bl = i / (kLgChunk + kCtx - 1), col = i - bl * (kLgChunk + kCtx - 1);
            const int b = bin0 + bl, gc = u0 + col;
            ra[k] = (bl < kLgBinsA && b < kBins && gc < c) ? Z[(int64_t)b * c + gc] : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < kNb; ++k) {
            const int i = tid + k * 256;
            const int b = i / kLgChunk, col = i - b * kLgChunk;
            const int gc = u0 + col;
            rb[k] = (b < kBins && gc < c) ? Z[(int64_t)b * c + gc] : 0.0f;
        }
    };
    if (item0 < item1) stage_load(item0);
    for (int item = item0; item < item1; ++item) {
        __syncthreads(); // every wave is done with the previous slabs
#pragma unroll
        for (int k = 0; k < kNa; ++k) {
            const int i = tid + k * 256;
            const int bl = i / (kLgChunk + kCtx - 1), col = i - bl * (kLgChunk + kCtx - 1);
            if (bl < kLgBinsA) slab_a[bl * kLgRowA + col] = ra[k];
        }
#pragma unroll
        for (int k = 0; k < kNb; ++k) {
            const int i = tid + k * 256;
            const int b = i / kLgChunk, col = i - b * kLgChunk;
            slab_b[b * kLgRowB + col] = rb[k];
        }
        __syncthreads();
        if (item + 1 < item1) stage_load(item + 1);
#pragma unroll 4
        for (int st = 0; st < kLgChunk / 2; ++st) {
            const int n = 2 * st + hb;
            float av[2], bv[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                av[t] = ok_a[t] ? slab_a[off_a[t] + n] : 0.0f;
                bv[t] = slab_b[off_b[t] + n];
            }
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0], bv[0], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0], bv[1], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1], bv[0], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1], bv[1], acc[1][1], 0, 0, 0);
        }
    }
    // D layout: column = lane & 31 (operand B: bin), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    float *out = part + (int64_t)blockIdx.y * kLgRowsPad * kLgCols;
#pragma unroll
    for (int ta = 0; ta < 2; ++ta)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int r = r0 + wr * 64 + ta * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * hb;
                out[(int64_t)r * kLgCols + wc * 64 + tb * 32 + li] = acc[ta][tb][reg];
            }
}

// gsum = part[0] + part[1] + ... in split order
__global__ __launch_bounds__(256) void lag_reduce_kernel(const float *__restrict__ part, int n_splits,
                                                         float *__restrict__ gsum)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    float s = 0.0f;
    for (int g = 0; g < n_splits; ++g) s += part[(int64_t)g * kLgRowsPad * kLgCols + i];
    gsum[i] = s;
}

// part[split][tile][128][128] = sum over the split's clips of V V^T on the tile pair tiles[blockIdx.x];
// operands straight from global memory (a lane's row of a correction vector: 128 contiguous bytes per
// half-wave), two correction vectors per MFMA.
__global__ __launch_bounds__(256) void vvt_kernel(const float *__restrict__ v, int n_clips, int clips_per_split,
                                                  const int2 *__restrict__ tiles, float *__restrict__ part)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1, hb = lane >> 5, li = lane & 31;
    const int2 tile = tiles[blockIdx.x];
    int ka[2], kb[2];
    bool ok_a[2], ok_b[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        ka[t] = tile.x * kCvTile + wr * 64 + t * 32 + li;
        kb[t] = tile.y * kCvTile + wc * 64 + t * 32 + li;
        ok_a[t] = ka[t] < kFrame;
        ok_b[t] = kb[t] < kFrame;
        ka[t] = ok_a[t] ? ka[t] : 0;
        kb[t] = ok_b[t] ? kb[t] : 0;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f32x16{0};
    const int c0 = blockIdx.y * clips_per_split, c1 = min(n_clips, c0 + clips_per_split);
    for (int clip = c0; clip < c1; ++clip) {
        const float *vc = v + ((int64_t)clip * kCvVec + hb) * kFrame;
#pragma unroll 4
        for (int st = 0; st < kCvVec / 2; ++st) {
            const float *vr = vc + (int64_t)2 * st * kFrame;
            float av[2], bv[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                av[t] = ok_a[t] ? vr[ka[t]] : 0.0f;
                bv[t] = ok_b[t] ? vr[kb[t]] : 0.0f;
            }
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0], bv[0], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0], bv[1], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1], bv[0], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1], bv[1], acc[1][1], 0, 0, 0);
        }
    }
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

// accum[k][k'] += scale * (G[(b, d), b'] or G[(b', -d), b]  -  sum over splits of the V V^T partial tiles)
// for k = (b, t), k' = (b', t'), d = t - t', on the tile pair tiles[blockIdx.x]
__global__ __launch_bounds__(256) void cov_expand_kernel(const float *__restrict__ gsum, const float *__restrict__ vpart,
                                                         int n_vsplits, float scale, const int2 *__restrict__ tiles,
                                                         float *__restrict__ accum)
{
    const int2 tile = tiles[blockIdx.x];
    const int e = blockIdx.y * 256 + threadIdx.x;
    const int ka = tile.x * kCvTile + e / kCvTile, kb = tile.y * kCvTile + e % kCvTile;
    float vv = 0.0f;
    for (int g = 0; g < n_vsplits; ++g) vv += vpart[((int64_t)g * gridDim.x + blockIdx.x) * (kCvTile * kCvTile) + e];
    if (ka < kFrame && kb < kFrame) {
        const int ba = ka / kCtx, ta = ka - ba * kCtx, bb = kb / kCtx, tb = kb - bb * kCtx;
        const float g = ta >= tb ? gsum[(int64_t)(ba * kCtx + ta - tb) * kLgCols + bb]
                                 : gsum[(int64_t)(bb * kCtx + tb - ta) * kLgCols + ba];
        accum[(int64_t)ka * kFrame + kb] += scale * (g - vv);
    }
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

// workspace of one launch_cov over n_clips clips of c columns: Z, V, partial G, G, partial V V^T
static size_t cov_ws_floats(int n_clips, int c, size_t off[5])
{
    size_t o = 0;
    off[0] = o;
    o += (size_t)n_clips * kBins * c;
    off[1] = o;
    o += (size_t)n_clips * kCvVec * kFrame;
    off[2] = o;
    o += (size_t)kLgSplits * kLgRowsPad * kLgCols;
    off[3] = o;
    o += (size_t)kLgRowsPad * kLgCols;
    off[4] = o;
    o += (size_t)kVvSplits * cov_tile_count() * kCvTile * kCvTile;
    return o;
}

size_t cov_workspace_bytes(int n_clips, int c)
{
    size_t off[5];
    return cov_ws_floats(n_clips, c, off) * sizeof(float);
}

void launch_cov(const float *d_db, int n_clips, int c, const int *d_tiles, float *d_ws, float *d_accum, hipStream_t s)
{
    size_t off[5];
    cov_ws_floats(n_clips, c, off);
    float *z = d_ws + off[0], *v = d_ws + off[1], *gpart = d_ws + off[2], *gsum = d_ws + off[3], *vpart = d_ws + off[4];
    const int nf = c - (kCtx - 1);
    const float scale = 1.0f / (float)(nf - 1); // hashprint_handle.h:101: / (rows - 1)
    const int2 *tiles = reinterpret_cast<const int2 *>(d_tiles);
    hipLaunchKernelGGL(lag_prep_kernel, dim3(kBins, n_clips), dim3(256), 0, s, d_db, c, nf, z, v);
    const int n_chunks = (c + kLgChunk - 1) / kLgChunk, n_items = n_clips * n_chunks;
    const int g_splits = std::min(kLgSplits, n_items), per = (n_items + g_splits - 1) / g_splits;
    hipLaunchKernelGGL(lag_corr_kernel, dim3(kLgRowsPad / kCvTile, g_splits), dim3(256), 0, s, z, c, n_chunks, n_items,
                       per, gpart);
    hipLaunchKernelGGL(lag_reduce_kernel, dim3(kLgRowsPad * kLgCols / 256), dim3(256), 0, s, gpart, g_splits, gsum);
    const int v_splits = std::min(kVvSplits, n_clips), cper = (n_clips + v_splits - 1) / v_splits;
    hipLaunchKernelGGL(vvt_kernel, dim3(cov_tile_count(), v_splits), dim3(256), 0, s, v, n_clips, cper, tiles, vpart);
    hipLaunchKernelGGL(cov_expand_kernel, dim3(cov_tile_count(), kCvTile * kCvTile / 256), dim3(256), 0, s, gsum, vpart,
                       v_splits, scale, tiles, d_accum);
}

} // namespace hpfw
