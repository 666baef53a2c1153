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
// matrix instruction are sixteen consecutive bins of one column.  A workgroup = 4 waves = 64 filters x 128 hashprints,
// and a wave = 16 FILTERS x all 128 hashprints on v_mfma_i32_16x16x64_i8: the digits of the workgroup's
// [121 bins][128 + 19 columns] slab of Du live in LDS as 16-byte units [bin chunk][column][digit], shared and read-only
// after the prologue, while every wave takes the digits of ITS filters straight from L2 into registers (3 KB per step of
// 64 k', two steps ahead) -- nothing is exchanged between the waves in the main loop, so it has no barrier: the two
// waves of a SIMD (two workgroups share a CU: 62 KB of LDS, 256 registers) interleave their matrix instructions freely,
// and one workgroup's staging (loads, quantisation: vector ALU) and epilogue run under the other's products.
#include <cmath>
#include <cstdint>
#include <vector>

#include "kernels.h"

namespace hpfw {

extern __shared__ __align__(16) unsigned char smem_raw[];

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int kQThreads = 256;                 // 4 waves; each owns 16 filters x 128 hashprints
constexpr int kQTileN = 128;                   // hashprints per workgroup
constexpr int kQCols = kQTileN + kCtx - 1;     // 147 columns of Du in the slab
constexpr int kQPitch = 160;                   // columns per chunk as stored: with this pitch the operand reads (ds_read_b128,
                                               // lanes = 16 columns x 4 chunks) meet no bank conflict (147 .. 157: two-way)
constexpr int kQChunks = 8;                    // bin chunks of 16 (121 bins padded to 128)
constexpr int kQUnit = 48;                     // bytes per (chunk, column): three digits x 16 bins
constexpr int kQSlabBytes = kQChunks * kQPitch * kQUnit;      // 61 440
constexpr int kQStepBytes = 3 * 64 * 16;                      // a wave's filter digits of one step (t, half of the chunks): [digit][lane][16] = 3 072
constexpr int kQSteps = 2 * kCtx;                             // 40 steps of 64 k'
constexpr int kQLdsBytes = kQSlabBytes + kQTileN * 4 * 2;     // + the waves' 16-bit parts of every hashprint: 62 464, two workgroups per CU
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
                                                                   const float *__restrict__ tmax, int c, int nhp, int n_tiles_x, int n_clips,
                                                                   uint64_t *__restrict__ hp, long long *__restrict__ dbg)
{
    unsigned char *slab = smem_raw;                                   // [chunk][column (pitch 160)][digit][16]
    unsigned short *parts = reinterpret_cast<unsigned short *>(smem_raw + kQSlabBytes); // [hashprint][wave]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kg = lane >> 4, cl = lane & 15;  // this lane's group of 16 k' inside a step; its column (B) / filter (A) in a tile
    // Workgroups go to the eight XCDs in turn (id mod 8), each with an L2 of its own: an XCD takes a contiguous run of
    // (clip, tile) pairs with the tile fastest, so that the 99 columns two neighbouring tiles share, and the columns a
    // tile reads twice (as c and as c + 80), come from HBM once: 2.34 -> 1.19 MB per clip by the counters (1.17 algorithmic; profiles/r04_pmc.json)
    const unsigned per_xcd = (gridDim.x + 7) / 8;
    const unsigned t = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (t >= (unsigned)(n_tiles_x * n_clips)) return;
    const int clip = t / n_tiles_x;
    const int n0 = (t - clip * n_tiles_x) * kQTileN;
#if defined(HPFW_Q_STAMPS)
    // diagnosis build (tools/q_stamps.py): s_memtime of wave 0 at the phase boundaries, written to dbg [workgroup][8]
    long long st[6];
    st[0] = __builtin_amdgcn_s_memtime();
#define Q_STAMP(k) st[k] = __builtin_amdgcn_s_memtime()
#else
#define Q_STAMP(k) ((void)0)
#endif
    const float *S = sdb + (int64_t)clip * kBins * c;
    const float ref = FROM_T ? tmax[clip] : 0.0f;
    // the wave's filter digits of the first two steps are on their way while the slab is quantised
    const v4i *img = fq_image + (size_t)wave * kQSteps * (kQStepBytes / 16) + lane;
    v4i a[3][3]; // three register sets in turn: the loads of step s + 2 are issued before the products of step s
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        a[0][d] = img[d * 64];
        a[1][d] = img[(kQStepBytes / 16) + d * 64];
    }
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
            // EVERY load is issued, from a valid address, and what lies outside the slab is zeroed afterwards: written as
            // `ok ? S[..] : 0` the loads sat in 64 branches of their own, and the compiler, short of registers for so many
            // merged values, waited for all but one of the loads in flight after every fourth -- seven memory round trips
            // in a row per round, 21 per workgroup: the staging took as long as the matrix loop (tools/q_stamps.py:
            // 69 k of a workgroup's 147 k cycles)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int bin = 16 * q + e;
                const bool ok = in && bin < kBins;
                const unsigned idx = ok ? (unsigned)(bin * c + gc) : 0u; // (clip-relative: < 121 * c, 32 bits; S[0] and S[kLag] exist)
                va[rr][e] = S[idx];
                vb[rr][e] = S[idx + kLag];
            }
        }
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int unit = tid + (r0 + rr) * kQThreads;
            const int q = unit / kQCols, col = unit - q * kQCols;
            const bool in = r0 + rr < kUnits && unit < kQChunks * kQCols && n0 + col + kLag < c;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const bool ok = in && 16 * q + e < kBins;
                va[rr][e] = ok ? va[rr][e] : 0.0f;
                vb[rr][e] = ok ? vb[rr][e] : 0.0f;
            }
        }
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int unit = tid + (r0 + rr) * kQThreads;
            if (r0 + rr >= kUnits || unit >= kQChunks * kQCols) break;
            const int q = unit / kQCols, col = unit - q * kQCols;
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
            v4i *dst = reinterpret_cast<v4i *>(slab + (size_t)(q * kQPitch + col) * kQUnit);
            dst[0] = v4i{(int)w0[0], (int)w0[1], (int)w0[2], (int)w0[3]};
            dst[1] = v4i{(int)w1[0], (int)w1[1], (int)w1[2], (int)w1[3]};
            dst[2] = v4i{(int)w2[0], (int)w2[1], (int)w2[2], (int)w2[3]};
        }
    }
    Q_STAMP(1);
    __syncthreads(); // the only barrier before the epilogue: the slab is read-only from here on
    Q_STAMP(2);
    // tiles of 16 hashprints that hold any (the last workgroup of a clip); the products of the others are skipped
    const int n_tiles = min(8, (nhp - n0 + 15) / 16);
    v4i acc[8][5];
#pragma unroll
    for (int f = 0; f < 8; ++f)
#pragma unroll
        for (int cls = 0; cls < 5; ++cls) acc[f][cls] = v4i{0, 0, 0, 0};
    // B operand of (step, tile f): chunk 4 (s & 1) + kg, column 16 f + cl + (s >> 1)
    const unsigned char *bl = slab + (size_t)(kg * kQPitch + cl) * kQUnit;
    auto step = [&](int s, const v4i (&aw)[3]) {
        const unsigned char *bs = bl + (size_t)((4 * (s & 1)) * kQPitch + (s >> 1)) * kQUnit;
        // the operand reads of tile f + 1 are issued BEFORE the nine products of tile f: read right where they are used, a
        // wave alone on its SIMD left the matrix pipe idle for an LDS round trip per tile (tools/q_stamps.py: 46 k cycles of
        // matrix instructions in a 69 k cycle loop)
        v4i nb0, nb1, nb2;
        {
            const v4i *bu = reinterpret_cast<const v4i *>(bs);
            nb0 = bu[0], nb1 = bu[1], nb2 = bu[2];
        }
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            if (f < n_tiles) {
                const v4i b0 = nb0, b1 = nb1, b2 = nb2;
                if (f + 1 < 8) { // (tile f + 1 of the slab exists whether or not it holds hashprints)
                    const v4i *bu = reinterpret_cast<const v4i *>(bs + (size_t)(16 * (f + 1)) * kQUnit);
                    nb0 = bu[0], nb1 = bu[1], nb2 = bu[2];
                }
                __builtin_amdgcn_sched_barrier(0); // (the scheduler would sink the reads back to where they are used)
                // filter digit i times spectrogram digit j goes to accumulator i + j
                acc[f][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(aw[0], b0, acc[f][0], 0, 0, 0);
                acc[f][1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(aw[0], b1, acc[f][1], 0, 0, 0);
                acc[f][2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(aw[0], b2, acc[f][2], 0, 0, 0);
                acc[f][3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(aw[1], b2, acc[f][3], 0, 0, 0);
                acc[f][4] = __builtin_amdgcn_mfma_i32_16x16x64_i8(aw[2], b2, acc[f][4], 0, 0, 0);
                acc[f][1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(aw[1], b0, acc[f][1], 0, 0, 0);
                acc[f][2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(aw[1], b1, acc[f][2], 0, 0, 0);
                acc[f][3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(aw[2], b1, acc[f][3], 0, 0, 0);
                acc[f][2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(aw[2], b0, acc[f][2], 0, 0, 0);
            }
        }
    };
    auto load_a = [&](int s, v4i (&aw)[3]) {
        if (s < kQSteps) {
#pragma unroll
            for (int d = 0; d < 3; ++d) aw[d] = img[(size_t)s * (kQStepBytes / 16) + d * 64];
        }
    };
    if (n_tiles > 0) {
#pragma unroll 1
        for (int s = 0; s < kQSteps - 1; s += 3) { // 40 steps = 13 x 3 + 1
            load_a(s + 2, a[2]);
            step(s, a[0]);
            load_a(s + 3, a[0]);
            step(s + 1, a[1]);
            load_a(s + 4, a[1]);
            step(s + 2, a[2]);
        }
        step(kQSteps - 1, a[0]);
    }
    Q_STAMP(3);
    // S10q: D = sum_c acc_c 2^(8c); D layout of the 16x16 tile: column (hashprint) = lane & 15, row (filter) = 4 (lane >> 4) + reg
#pragma unroll
    for (int f = 0; f < 8; ++f) {
        const int n = n0 + 16 * f + cl;
        (void)n;
        unsigned bits = 0; // this wave's 16 filters of hashprint n, filter 16 wave + row at bit 15 - row
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int row = 4 * kg + reg;
            long long v = acc[f][4][reg];
#pragma unroll
            for (int cls = 3; cls >= 0; --cls) v = v * 256 + acc[f][cls][reg];
            bits |= (unsigned)(v >= 0) << (15 - row);
#if !defined(HPFW_Q_STAMPS)
            if (dbg && f < n_tiles && n < nhp) dbg[((int64_t)clip * kFilters + 16 * wave + row) * nhp + n] = v;
#endif
        }
        // the four lanes of a column hold four filters each
        bits |= (unsigned)__shfl_xor((int)bits, 16);
        bits |= (unsigned)__shfl_xor((int)bits, 32);
        if (kg == 0) parts[(16 * f + cl) * 4 + wave] = (unsigned short)bits;
    }
    Q_STAMP(4);
    __syncthreads();
    if (tid < kQTileN && n0 + tid < nhp) {
        const unsigned short *p = parts + tid * 4;
        hp[(int64_t)clip * nhp + n0 + tid] = ((uint64_t)p[0] << 48) | ((uint64_t)p[1] << 32) | ((uint64_t)p[2] << 16) | (uint64_t)p[3];
    }
#if defined(HPFW_Q_STAMPS)
    Q_STAMP(5);
    if (dbg && tid == 0) {
        for (int k = 0; k < 6; ++k) dbg[(int64_t)t * 8 + k] = st[k];
        dbg[(int64_t)t * 8 + 6] = __builtin_amdgcn_s_getreg((15 << 11) | 4); // HW_ID: wave, SIMD, CU, SH, SE
        dbg[(int64_t)t * 8 + 7] = blockIdx.x;
    }
#endif
}

// host: the filters' digits as the A operand of v_mfma_i32_16x16x64_i8, [wave][step s = 2 t + p][digit][lane][16 bytes]:
// byte e of lane l = digit of fq[row = 16 wave + (l & 15)][k = 20 bin + t], bin = 64 p + 16 (l >> 4) + e (zero for bin >= 121)
void pack_filters_q(const float *f, std::vector<int8_t> &image)
{
    std::vector<int32_t> fq((size_t)kFilters * kFrame);
    for (int r = 0; r < kFilters; ++r) {
        float m = 0.0f;
        for (int k = 0; k < kFrame; ++k) m = std::fmax(m, std::fabs(f[(size_t)k * kFilters + r]));
        const int e = m > 0.0f ? 21 - std::ilogb(m) : 0;
        for (int k = 0; k < kFrame; ++k) fq[(size_t)r * kFrame + k] = (int32_t)std::rint(std::ldexp(f[(size_t)k * kFilters + r], e));
    }
    image.assign((size_t)4 * kQSteps * kQStepBytes, 0);
    for (int w = 0; w < 4; ++w)
        for (int t = 0; t < kCtx; ++t)
            for (int p = 0; p < 2; ++p)
                for (int l = 0; l < 64; ++l)
                    for (int e = 0; e < 16; ++e) {
                        const int bin = 64 * p + 16 * (l >> 4) + e;
                        if (bin >= kBins) continue;
                        const int u = fq[(size_t)(16 * w + (l & 15)) * kFrame + bin * kCtx + t];
                        const int d0 = ((u + 128) & 255) - 128, u1 = (u - d0) >> 8, d1 = ((u1 + 128) & 255) - 128, d2 = (u1 - d1) >> 8;
                        const int d[3] = {d0, d1, d2};
                        for (int i = 0; i < 3; ++i)
                            image[((size_t)w * kQSteps + (2 * t + p)) * kQStepBytes + ((size_t)i * 64 + l) * 16 + e] = (int8_t)d[i];
                    }
}

#if defined(HPFW_Q_STAMPS)
static long long *g_q_stamps = nullptr;
extern "C" void hpfw_gpu_debug_set_q_stamps(void *d) { g_q_stamps = static_cast<long long *>(d); }
#endif

size_t project_q_image_bytes() { return (size_t)4 * kQSteps * kQStepBytes; }

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
#if defined(HPFW_Q_STAMPS)
    if (!d_dbg) d_dbg = g_q_stamps; // (tools/q_stamps.py extract: the stamps of the extraction's own launches)
#endif
    const int tiles = (nhp + kQTileN - 1) / kQTileN;
    const dim3 grid(8 * (unsigned)(((int64_t)tiles * n_clips + 7) / 8)); // one-dimensional, in XCD-aware order
    if (d_tmax)
        hipLaunchKernelGGL(hashprint_q_kernel<true>, grid, dim3(kQThreads), kQLdsBytes, s, static_cast<const v4i *>(d_fq_image), d_db, d_tmax,
                           c, nhp, tiles, n_clips, d_hp, d_dbg);
    else
        hipLaunchKernelGGL(hashprint_q_kernel<false>, grid, dim3(kQThreads), kQLdsBytes, s, static_cast<const v4i *>(d_fq_image), d_db, d_tmax,
                           c, nhp, tiles, n_clips, d_hp, d_dbg);
}

} // namespace hpfw
