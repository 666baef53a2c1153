// k_project_q.hip -- a5..a8 in fixed point: dB spectrogram -> exact integer projection -> delta -> 64-bit hashprints
// (DESIGN.md S9q / S10q).
//
// The reference multiplies filters and frames in f32 (an Eigen/MKL sgemm [64 x 2420] . [2420 x n_frames],
// include/hpfw/core/parallel_collector.h:57,127) and keeps only the SIGN of P[r,i] - P[r,i+80]
// (hashprint_handle.h:119-122).  On the f32 matrix pipe that product is the largest kernel of the extraction.  Here
// both factors are rounded once to 24-bit fixed point,
//   u[b][c]  = rint(S[b][c] * 2^17) + 40 * 2^17        (S in [-80, 0] dB; the offset cancels in the difference)
//   fq[r][k] = rint(F[r][k] * 2^(21 - ilogb(max_k |F[r][k]|)))   (a positive power of two per row: no sign changes)
// split into three balanced base-256 digits each (d in [-128, 127]: x = d0 + 256 d1 + 65536 d2), and the 2420-term
// sums of the nine digit products run on v_mfma_i32_32x32x32_i8 -- 16 times the multiply-adds per clock of the f32
// form (measured: the instruction takes the f32 form's 16 passes at 16 times its K), exact, and free of any summation
// order: digit products of equal weight share an int32 accumulator (|sum| <= 3 * 2560 * 128 * 128 < 2^27),
// Pq = sum_c acc_c 2^(8c) in int64.  2.8 ms per 1000 clips of 30 s (94 % of that rate) against 5.8 for the f32 kernel.
//
// K is taken as k' = 128 t + bin (bins padded to 128 with zero digits) so that the sixteen bytes a lane hands the
// matrix instruction are sixteen consecutive bins of one spectrogram column: the workgroup keeps the digits of its
// [121 bins][256 + 19 columns] slab in LDS as 16-byte units [bin chunk][column][digit], the filter digits of one
// context step t (24 KB) stream through a double buffer.
#include <cmath>
#include <cstdint>
#include <vector>

#include "kernels.h"

namespace hpfw {

extern __shared__ __align__(16) unsigned char smem_raw[];

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int kQThreads = 512;                 // 8 waves; each owns 64 filters x 32 frames
constexpr int kQTileN = 256;                   // frames per workgroup
constexpr int kQCols = kQTileN + kCtx - 1;     // 275 spectrogram columns in the slab
constexpr int kQChunks = 8;                    // bin chunks of 16 (121 bins padded to 128)
constexpr int kQUnit = 48;                     // bytes per (chunk, column): three digits x 16 bins
constexpr int kQSlabBytes = kQChunks * kQCols * kQUnit;       // 105 600
constexpr int kQStepBytes = 2 * 3 * 4 * 64 * 16;              // filter digits of one t: [tile][digit][c][lane][16] = 24 576
constexpr int kQLdsBytes = kQSlabBytes + 2 * kQStepBytes;     // 154 752 (the epilogue's [64][256] int64 tile, 131 072, lies over it)

__device__ __forceinline__ void q_digits(int u, int &d0, int &d1, int &d2)
{
    d0 = ((u + 128) & 255) - 128;
    const int u1 = (u - d0) >> 8;
    d1 = ((u1 + 128) & 255) - 128;
    d2 = (u1 - d1) >> 8;
}

// FROM_T: sdb holds the dB terms t written by the chirp-z kernel; S = max(t - tmax[clip], -80) (convert.h:12-15)
// rides on the staging loads, as in project_kernel.
template <bool FROM_T>
__global__ __launch_bounds__(kQThreads, 2) void project_q_kernel(const v4i *__restrict__ fq_image, const float *__restrict__ sdb,
                                                                 const float *__restrict__ tmax, int c, int nf,
                                                                 long long *__restrict__ proj, uint64_t *__restrict__ hp)
{
    unsigned char *slab = smem_raw;                                   // [chunk][column][digit][16]
    v4i *abuf = reinterpret_cast<v4i *>(smem_raw + kQSlabBytes);      // [2][tile][digit][c][lane]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, nl = lane & 31;
    const int clip = blockIdx.y;
    const int n0 = blockIdx.x * kQTileN;
    const float *S = sdb + (int64_t)clip * kBins * c;
    const float ref = FROM_T ? tmax[clip] : 0.0f;
    // the filter digits of t = 0 are on their way while the slab is quantised
    constexpr int kAPer = kQStepBytes / 16 / kQThreads; // 3 pieces of 16 bytes per thread and step
    v4i areg[kAPer];
#pragma unroll
    for (int e = 0; e < kAPer; ++e) areg[e] = fq_image[tid + e * kQThreads];
    // slab: one (chunk, column) unit = 16 bins of one column, three digit planes.  Every load of the thread's units
    // (up to 5 x 16) is issued before the first value is quantised: one latency instead of five
    constexpr int kUnits = (kQChunks * kQCols + kQThreads - 1) / kQThreads; // 5
    float v[kUnits][16];
#pragma unroll
    for (int r = 0; r < kUnits; ++r) {
        const int unit = tid + r * kQThreads;
        const int q = unit / kQCols, col = unit - q * kQCols;
        const int gc = n0 + col;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int bin = 16 * q + e;
            v[r][e] = (unit < kQChunks * kQCols && bin < kBins && gc < c) ? S[(int64_t)bin * c + gc] : 0.0f;
        }
    }
#pragma unroll
    for (int r = 0; r < kUnits; ++r) {
        const int unit = tid + r * kQThreads;
        if (unit >= kQChunks * kQCols) break;
        const int q = unit / kQCols, col = unit - q * kQCols;
        const int gc = n0 + col;
        unsigned w0[4] = {0, 0, 0, 0}, w1[4] = {0, 0, 0, 0}, w2[4] = {0, 0, 0, 0};
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int bin = 16 * q + e;
            float x = v[r][e];
            if (FROM_T) {
                const float l = x - ref;
                x = l < -80.0f ? -80.0f : l;
            }
            int d0 = 0, d1 = 0, d2 = 0;
            if (bin < kBins && gc < c) q_digits((int)__builtin_rintf(x * 131072.0f) + 40 * 131072, d0, d1, d2);
            w0[e >> 2] |= (unsigned)(d0 & 255) << (8 * (e & 3));
            w1[e >> 2] |= (unsigned)(d1 & 255) << (8 * (e & 3));
            w2[e >> 2] |= (unsigned)(d2 & 255) << (8 * (e & 3));
        }
        v4i *dst = reinterpret_cast<v4i *>(slab + (size_t)unit * kQUnit);
        dst[0] = v4i{(int)w0[0], (int)w0[1], (int)w0[2], (int)w0[3]};
        dst[1] = v4i{(int)w1[0], (int)w1[1], (int)w1[2], (int)w1[3]};
        dst[2] = v4i{(int)w2[0], (int)w2[1], (int)w2[2], (int)w2[3]};
    }
#pragma unroll
    for (int e = 0; e < kAPer; ++e) abuf[tid + e * kQThreads] = areg[e];
    __syncthreads();
    v16i acc[2][5];
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
        for (int cl = 0; cl < 5; ++cl) acc[t2][cl] = v16i{0};
    const bool active = n0 + wave * 32 < nf; // a wave whose frames lie past the clip only helps staging
    const int colw = wave * 32 + nl;         // this lane's frame inside the tile
    v4i a[2][2][3], b[2][3]; // two sets in turn: the operands of K-step s + 1 are read while those of step s multiply
    auto fetch = [&](int set, int t, int cc) {
        const v4i *ab = abuf + (t & 1) * (kQStepBytes / 16);
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int i = 0; i < 3; ++i) a[set][t2][i] = ab[((t2 * 3 + i) * 4 + cc) * 64 + lane];
        const v4i *bu = reinterpret_cast<const v4i *>(slab + ((size_t)(2 * cc + h) * kQCols + colw + t) * kQUnit);
#pragma unroll
        for (int j = 0; j < 3; ++j) b[set][j] = bu[j];
    };
    auto mult = [&](int set) {
        // the two filter tiles in turn: instructions on the same accumulator (digit products of equal weight) stay four apart
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2)
                    acc[t2][i + j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[set][t2][i], b[set][j], acc[t2][i + j], 0, 0, 0);
    };
    for (int t = 0; t < kCtx; ++t) {
        if (t + 1 < kCtx) { // next step's filter digits: loads now, LDS writes after this step's matrix instructions
#pragma unroll
            for (int e = 0; e < kAPer; ++e) areg[e] = fq_image[(size_t)(t + 1) * (kQStepBytes / 16) + tid + e * kQThreads];
        }
        if (active) {
            fetch(0, t, 0);
            fetch(1, t, 1);
            mult(0);
            fetch(0, t, 2);
            mult(1);
            fetch(1, t, 3);
            mult(0);
            mult(1);
        }
        if (t + 1 < kCtx) {
            v4i *an = abuf + ((t + 1) & 1) * (kQStepBytes / 16);
#pragma unroll
            for (int e = 0; e < kAPer; ++e) an[tid + e * kQThreads] = areg[e];
        }
        __syncthreads();
    }
    // S10q here for the frames whose partner 80 frames on lies in this tile (176 of 256): the tile's Pq goes through
    // the LDS (the slab is no longer read) instead of through HBM; only the columns a neighbouring tile needs -- the
    // first 80 -- and those that need a neighbour -- the last 80 -- are written out for pack_q_edge_kernel.
    long long *pl = reinterpret_cast<long long *>(smem_raw); // [64][256]
    long long *P = proj + (int64_t)clip * kFilters * nf;
    const int n = n0 + colw;
    if (active) {
        // D layout of the 32x32 tile: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
        const bool edge = (colw < kLag || colw >= kQTileN - kLag) && n < nf;
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = 32 * t2 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                long long v = acc[t2][4][reg];
#pragma unroll
                for (int cl = 3; cl >= 0; --cl) v = v * 256 + acc[t2][cl][reg];
                pl[row * kQTileN + colw] = v;
                if (edge) P[(int64_t)row * nf + n] = v;
            }
    }
    __syncthreads();
    const int nhp = nf - kLag;
    if (tid < kQTileN - kLag && n0 + tid < nhp) { // (columns of waves past the clip are never read: their frames are >= nhp)
        uint64_t v = 0;
#pragma unroll 16
        for (int r = 0; r < kFilters; ++r) {
            const long long d = pl[r * kQTileN + tid] - pl[r * kQTileN + tid + kLag];
            v |= (uint64_t)(d >= 0) << (63 - r);
        }
        hp[(int64_t)clip * nhp + n0 + tid] = v;
    }
}

// the frames whose partner lies in the next tile: the last 80 of every tile of 256, from the edge columns in HBM
__global__ __launch_bounds__(128) void pack_q_edge_kernel(const long long *__restrict__ proj, int nf, int nhp, uint64_t *__restrict__ hp)
{
    const int i = blockIdx.x * kQTileN + (kQTileN - kLag) + threadIdx.x;
    const int clip = blockIdx.y;
    if (threadIdx.x >= kLag || i >= nhp) return;
    const long long *P = proj + (int64_t)clip * kFilters * nf + i;
    uint64_t v = 0;
#pragma unroll 16
    for (int r = 0; r < kFilters; ++r) {
        const long long d = P[(int64_t)r * nf] - P[(int64_t)r * nf + kLag];
        v |= (uint64_t)(d >= 0) << (63 - r);
    }
    hp[(int64_t)clip * nhp + i] = v;
}

// host: the filters' digits as the A operand of v_mfma_i32_32x32x32_i8, [t][tile][digit][c][lane][16 bytes]: byte e of
// lane l = digit of fq[row = 32 tile + (l & 31)][k = 20 bin + t], bin = 32 c + 16 (l >> 5) + e (zero for bin >= 121)
void pack_filters_q(const float *f, std::vector<int8_t> &image)
{
    std::vector<int32_t> fq((size_t)kFilters * kFrame);
    for (int r = 0; r < kFilters; ++r) {
        float m = 0.0f;
        for (int k = 0; k < kFrame; ++k) m = std::fmax(m, std::fabs(f[(size_t)k * kFilters + r]));
        const int e = m > 0.0f ? 21 - std::ilogb(m) : 0;
        for (int k = 0; k < kFrame; ++k) fq[(size_t)r * kFrame + k] = (int32_t)std::rint(std::ldexp(f[(size_t)k * kFilters + r], e));
    }
    image.assign((size_t)kCtx * kQStepBytes, 0);
    for (int t = 0; t < kCtx; ++t)
        for (int t2 = 0; t2 < 2; ++t2)
            for (int cc = 0; cc < 4; ++cc)
                for (int l = 0; l < 64; ++l)
                    for (int e = 0; e < 16; ++e) {
                        const int bin = 32 * cc + 16 * (l >> 5) + e;
                        if (bin >= kBins) continue;
                        const int u = fq[(size_t)(32 * t2 + (l & 31)) * kFrame + bin * kCtx + t];
                        const int d0 = ((u + 128) & 255) - 128, u1 = (u - d0) >> 8, d1 = ((u1 + 128) & 255) - 128, d2 = (u1 - d1) >> 8;
                        const int d[3] = {d0, d1, d2};
                        for (int i = 0; i < 3; ++i)
                            image[(size_t)t * kQStepBytes + ((((size_t)t2 * 3 + i) * 4 + cc) * 64 + l) * 16 + e] = (int8_t)d[i];
                    }
}

size_t project_q_image_bytes() { return (size_t)kCtx * kQStepBytes; }

// dB spectrograms -> hashprints [n_clips][c - 99] except the last 80 frames of every tile of 256 (launch_pack_q_edge);
// d_proj: scratch of n_clips * 64 * (c - 19) int64 (edge columns only are written)
void launch_hashprints_q(const void *d_fq_image, const float *d_db, const float *d_tmax, int n_clips, int c, long long *d_proj,
                         uint64_t *d_hp, hipStream_t s)
{
    static PerDeviceOnce attr_set;
    if (attr_set.need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(project_q_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(project_q_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024);
        attr_set.mark();
    }
    const int nf = c - (kCtx - 1), nhp = nf - kLag;
    if (nhp <= 0) return;
    dim3 grid((nf + kQTileN - 1) / kQTileN, n_clips);
    if (d_tmax)
        hipLaunchKernelGGL(project_q_kernel<true>, grid, dim3(kQThreads), kQLdsBytes, s, static_cast<const v4i *>(d_fq_image), d_db, d_tmax, c,
                           nf, d_proj, d_hp);
    else
        hipLaunchKernelGGL(project_q_kernel<false>, grid, dim3(kQThreads), kQLdsBytes, s, static_cast<const v4i *>(d_fq_image), d_db, d_tmax, c,
                           nf, d_proj, d_hp);
}

// the hashprints of the last 80 frames of every tile (their partners lie in the next tile), after launch_hashprints_q
void launch_pack_q_edge(const long long *d_proj, int n_clips, int c, uint64_t *d_hp, hipStream_t s)
{
    const int nf = c - (kCtx - 1), nhp = nf - kLag;
    if (nhp <= kQTileN - kLag) return; // a single tile whose every frame has its partner inside
    hipLaunchKernelGGL(pack_q_edge_kernel, dim3((nf + kQTileN - 1) / kQTileN, n_clips), dim3(128), 0, s, d_proj, nf, nhp, d_hp);
}

} // namespace hpfw
