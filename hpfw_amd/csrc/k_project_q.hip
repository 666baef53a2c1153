// k_project_q.hip -- a4 (reference level and clip), a5..a8 in fixed point: dB terms -> 64-bit hashprints in ONE kernel
// (DESIGN.md S9q / S10q).
//
// The reference multiplies filters and frames in f32 (an Eigen/MKL sgemm [64 x 2420] . [2420 x n_frames],
// include/hpfw/core/parallel_collector.h:57,127) and keeps only the SIGN of P[r,i] - P[r,i+80]
// (hashprint_handle.h:119-122).  Here both factors are rounded once to fixed point,
//   u[b][c]  = rint(S[b][c] * 98304)                         (S in [-80, 0] dB; 98304 = 3 * 2^15)
//   fq[r][k] = rint(F[r][k] * 2^(21 - ilogb(max_k |F[r][k]|)))   (a positive power of two per row: no sign changes)
// and the difference is taken BEFORE the product: with Du[b][c] = u[b][c] - u[b][c + 80] (|Du| <= 80 * 98304 < 2^23),
//   D[r][i] = Pq[r][i] - Pq[r][i + 80] = sum_{b,t} fq[r][20 b + t] * Du[b][i + t]
// exactly -- integers have no rounding and no summation order -- and bit (63 - r) of hp[i] = (D[r][i] >= 0).  No
// projection ever exists in memory: the kernel reads the dB terms and writes hashprints.
// Both factors are split into three balanced base-256 digits (d in [-128, 127]: x = d0 + 256 d1 + 65536 d2) and the
// 2420-term sums of the nine digit products run on v_mfma_i32_32x32x32_i8 (32 cycles per instruction and SIMD: 16 times
// the multiply-adds per clock of the f32 form); digit products of equal weight share an int32 accumulator
// (|sum| <= 3 * 2420 * 128 * 128 < 2^27), D = sum_c acc_c 2^(8c) in int64.
//
// K is taken as k' = 128 t + bin (bins padded to 128 with zero digits) so that the sixteen bytes a lane hands the
// matrix instruction are sixteen consecutive bins of one column.  A workgroup = 4 waves = 64 filters x 128 hashprints;
// the digits of its [121 bins][128 + 19 columns] slab of Du live in LDS as 16-byte units [bin chunk][column][digit],
// the filter digits stream through a double buffer in half steps (t, pair of 32-bin chunks: 12 KB).  81 KB of LDS and
// 256 registers: TWO workgroups share a CU, so one's staging (loads, quantisation: vector ALU) and epilogue run under
// the other's matrix instructions.
#include <cmath>
#include <cstdint>
#include <vector>

#include "kernels.h"

namespace hpfw {

extern __shared__ __align__(16) unsigned char smem_raw[];

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int kQThreads = 256;                 // 4 waves; each owns 64 filters x 32 hashprints
constexpr int kQTileN = 128;                   // hashprints per workgroup
constexpr int kQCols = kQTileN + kCtx - 1;     // 147 columns of Du in the slab
constexpr int kQChunks = 8;                    // bin chunks of 16 (121 bins padded to 128)
constexpr int kQUnit = 48;                     // bytes per (chunk, column): three digits x 16 bins
constexpr int kQSlabBytes = kQChunks * kQCols * kQUnit;       // 56 448
constexpr int kQHalfBytes = 2 * 3 * 2 * 64 * 16;              // filter digits of one half step: [tile][digit][chunk pair member][lane][16] = 12 288
constexpr int kQHalfSteps = 2 * kCtx;                         // 40
constexpr int kQLdsBytes = kQSlabBytes + 2 * kQHalfBytes;     // 81 024: two workgroups per CU
constexpr float kQScale = 98304.0f;

// the three balanced base-256 digits of u as the three low bytes of one word (|u| < 2^23)
__device__ __forceinline__ unsigned q_digit_bytes(int u) { return ((unsigned)u + 0x808080u) ^ 0x808080u; }

__device__ __forceinline__ int q_fixed(float s)
{
    return (int)__builtin_rintf(__builtin_fminf(__builtin_fmaxf(s, -80.0f), 0.0f) * kQScale);
}

// FROM_T: sdb holds the dB terms t written by the chirp-z kernel; S = max(t - tmax[clip], -80) (convert.h:12-15)
// rides on the staging loads.  dbg (tests only, NULL in extraction): D as int64 [clip][64][nhp].
template <bool FROM_T>
__global__ __launch_bounds__(kQThreads, 2) void hashprint_q_kernel(const v4i *__restrict__ fq_image, const float *__restrict__ sdb,
                                                                   const float *__restrict__ tmax, int c, int nhp,
                                                                   uint64_t *__restrict__ hp, long long *__restrict__ dbg)
{
    unsigned char *slab = smem_raw;                                   // [chunk][column][digit][16]
    v4i *abuf = reinterpret_cast<v4i *>(smem_raw + kQSlabBytes);      // [2][tile][digit][pair member][lane]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, nl = lane & 31;
    const int clip = blockIdx.y;
    const int n0 = blockIdx.x * kQTileN;
    const float *S = sdb + (int64_t)clip * kBins * c;
    const float ref = FROM_T ? tmax[clip] : 0.0f;
    // the filter digits of half step 0 are on their way while the slab is quantised
    constexpr int kAPer = kQHalfBytes / 16 / kQThreads; // 3 pieces of 16 bytes per thread and half step
    v4i areg[kAPer];
#pragma unroll
    for (int e = 0; e < kAPer; ++e) areg[e] = fq_image[tid + e * kQThreads];
    // slab: one (chunk, column) unit = 16 bins of one column of Du, three digit planes.  The loads of a round (16 bins x
    // the column and its partner 80 on) are all issued before the first value is quantised
    constexpr int kUnits = (kQChunks * kQCols + kQThreads - 1) / kQThreads; // 5
#pragma unroll 1
    for (int r0 = 0; r0 < kUnits; r0 += 2) {
        float va[2][16], vb[2][16];
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int unit = tid + (r0 + rr) * kQThreads;
            const int q = unit / kQCols, col = unit - q * kQCols;
            const int gc = n0 + col;
            const bool in = r0 + rr < kUnits && unit < kQChunks * kQCols && gc + kLag < c;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int bin = 16 * q + e;
                const bool ok = in && bin < kBins;
                va[rr][e] = ok ? S[(int64_t)bin * c + gc] : 0.0f;
                vb[rr][e] = ok ? S[(int64_t)bin * c + gc + kLag] : 0.0f;
            }
        }
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int unit = tid + (r0 + rr) * kQThreads;
            if (r0 + rr >= kUnits || unit >= kQChunks * kQCols) break;
            unsigned w0[4] = {0, 0, 0, 0}, w1[4] = {0, 0, 0, 0}, w2[4] = {0, 0, 0, 0};
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float xa = va[rr][e], xb = vb[rr][e];
                if (FROM_T) {
                    xa -= ref;
                    xb -= ref;
                }
                // (values outside the slab's valid part were loaded as 0 on both sides: Du = 0, digits 0)
                const unsigned d = q_digit_bytes(q_fixed(xa) - q_fixed(xb));
                w0[e >> 2] |= (d & 255u) << (8 * (e & 3));
                w1[e >> 2] |= ((d >> 8) & 255u) << (8 * (e & 3));
                w2[e >> 2] |= ((d >> 16) & 255u) << (8 * (e & 3));
            }
            v4i *dst = reinterpret_cast<v4i *>(slab + (size_t)unit * kQUnit);
            dst[0] = v4i{(int)w0[0], (int)w0[1], (int)w0[2], (int)w0[3]};
            dst[1] = v4i{(int)w1[0], (int)w1[1], (int)w1[2], (int)w1[3]};
            dst[2] = v4i{(int)w2[0], (int)w2[1], (int)w2[2], (int)w2[3]};
        }
    }
#pragma unroll
    for (int e = 0; e < kAPer; ++e) abuf[tid + e * kQThreads] = areg[e];
    __syncthreads();
    v16i acc[2][5];
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
        for (int cl = 0; cl < 5; ++cl) acc[t2][cl] = v16i{0};
    const bool active = n0 + wave * 32 < nhp; // a wave whose hashprints lie past the clip only helps staging
    const int colw = wave * 32 + nl;          // this lane's hashprint inside the tile
    v4i a[2][2][3], b[2][3]; // two sets in turn: one per member of the chunk pair
    auto fetch = [&](int set, int hs, int m) {
        const v4i *ab = abuf + (hs & 1) * (kQHalfBytes / 16);
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int i = 0; i < 3; ++i) a[set][t2][i] = ab[((t2 * 3 + i) * 2 + m) * 64 + lane];
        const int cc = 2 * (hs & 1) + m, t = hs >> 1;
        const v4i *bu = reinterpret_cast<const v4i *>(slab + ((size_t)(2 * cc + h) * kQCols + colw + t) * kQUnit);
#pragma unroll
        for (int j = 0; j < 3; ++j) b[set][j] = bu[j];
    };
    auto mult = [&](int set) {
        // the two filter tiles in turn: instructions on the same accumulator (digit products of equal weight) stay apart
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2)
                    acc[t2][i + j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[set][t2][i], b[set][j], acc[t2][i + j], 0, 0, 0);
    };
    for (int hs = 0; hs < kQHalfSteps; ++hs) {
        if (hs + 1 < kQHalfSteps) { // next half step's filter digits: loads now, LDS writes after this one's matrix instructions
#pragma unroll
            for (int e = 0; e < kAPer; ++e) areg[e] = fq_image[(size_t)(hs + 1) * (kQHalfBytes / 16) + tid + e * kQThreads];
        }
        if (active) {
            fetch(0, hs, 0);
            fetch(1, hs, 1);
            mult(0);
            mult(1);
        }
        if (hs + 1 < kQHalfSteps) {
            v4i *an = abuf + ((hs + 1) & 1) * (kQHalfBytes / 16);
#pragma unroll
            for (int e = 0; e < kAPer; ++e) an[tid + e * kQThreads] = areg[e];
        }
        __syncthreads();
    }
    if (!active) return;
    // S10q: D = sum_c acc_c 2^(8c); D layout of the 32x32 tile: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    const int n = n0 + colw;
    uint64_t bits = 0;
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row = 32 * t2 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
            long long v = acc[t2][4][reg];
#pragma unroll
            for (int cl = 3; cl >= 0; --cl) v = v * 256 + acc[t2][cl][reg];
            bits |= (uint64_t)(v >= 0) << (63 - row);
            if (dbg && n < nhp) dbg[((int64_t)clip * kFilters + row) * nhp + n] = v;
        }
    // the two halves of the wave hold complementary rows of the same 32 hashprints
    const unsigned lo = (unsigned)bits, hi = (unsigned)(bits >> 32);
    const unsigned olo = (unsigned)__shfl_xor((int)lo, 32), ohi = (unsigned)__shfl_xor((int)hi, 32);
    bits |= ((uint64_t)ohi << 32) | olo;
    if (h == 0 && n < nhp) hp[(int64_t)clip * nhp + n] = bits;
}

// host: the filters' digits as the A operand of v_mfma_i32_32x32x32_i8, [t][pair p][tile][digit][member m][lane][16 bytes]:
// byte e of lane l = digit of fq[row = 32 tile + (l & 31)][k = 20 bin + t], bin = 32 (2 p + m) + 16 (l >> 5) + e (zero for bin >= 121)
void pack_filters_q(const float *f, std::vector<int8_t> &image)
{
    std::vector<int32_t> fq((size_t)kFilters * kFrame);
    for (int r = 0; r < kFilters; ++r) {
        float m = 0.0f;
        for (int k = 0; k < kFrame; ++k) m = std::fmax(m, std::fabs(f[(size_t)k * kFilters + r]));
        const int e = m > 0.0f ? 21 - std::ilogb(m) : 0;
        for (int k = 0; k < kFrame; ++k) fq[(size_t)r * kFrame + k] = (int32_t)std::rint(std::ldexp(f[(size_t)k * kFilters + r], e));
    }
    image.assign((size_t)kQHalfSteps * kQHalfBytes, 0);
    for (int t = 0; t < kCtx; ++t)
        for (int p = 0; p < 2; ++p)
            for (int t2 = 0; t2 < 2; ++t2)
                for (int m = 0; m < 2; ++m)
                    for (int l = 0; l < 64; ++l)
                        for (int e = 0; e < 16; ++e) {
                            const int bin = 32 * (2 * p + m) + 16 * (l >> 5) + e;
                            if (bin >= kBins) continue;
                            const int u = fq[(size_t)(32 * t2 + (l & 31)) * kFrame + bin * kCtx + t];
                            const int d0 = ((u + 128) & 255) - 128, u1 = (u - d0) >> 8, d1 = ((u1 + 128) & 255) - 128, d2 = (u1 - d1) >> 8;
                            const int d[3] = {d0, d1, d2};
                            for (int i = 0; i < 3; ++i)
                                image[(size_t)(2 * t + p) * kQHalfBytes + ((((size_t)t2 * 3 + i) * 2 + m) * 64 + l) * 16 + e] = (int8_t)d[i];
                        }
}

size_t project_q_image_bytes() { return (size_t)kQHalfSteps * kQHalfBytes; }

// dB terms (d_tmax != NULL) or dB spectrograms -> hashprints [n_clips][c - 99]; d_dbg: NULL, or D [n_clips][64][c - 99] (tests)
void launch_hashprints_q(const void *d_fq_image, const float *d_db, const float *d_tmax, int n_clips, int c, uint64_t *d_hp,
                         long long *d_dbg, hipStream_t s)
{
    static PerDeviceOnce attr_set;
    if (attr_set.need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(hashprint_q_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  kQLdsBytes);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(hashprint_q_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  kQLdsBytes);
        attr_set.mark();
    }
    const int nhp = c - (kCtx - 1) - kLag;
    if (nhp <= 0 || n_clips <= 0) return;
    dim3 grid((nhp + kQTileN - 1) / kQTileN, n_clips);
    if (d_tmax)
        hipLaunchKernelGGL(hashprint_q_kernel<true>, grid, dim3(kQThreads), kQLdsBytes, s, static_cast<const v4i *>(d_fq_image), d_db, d_tmax,
                           c, nhp, d_hp, d_dbg);
    else
        hipLaunchKernelGGL(hashprint_q_kernel<false>, grid, dim3(kQThreads), kQLdsBytes, s, static_cast<const v4i *>(d_fq_image), d_db, d_tmax,
                           c, nhp, d_hp, d_dbg);
}

} // namespace hpfw
