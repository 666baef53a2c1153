// k_forward.hip -- a1 + the forward half of a2: int16 PCM -> forward DFT bins [kmin, kmax), for 7-smooth clip lengths.
//
// Replaces essentia MonoLoader (PCM16 mono 44.1 kHz case) and the length-N FFT inside essentia
// NSGConstantQ::compute, called from CQT<>::spectrogram (reference include/hpfw/spectrum/cqt.h:45-52,
// 66-71).  N = n1 * n2 (n2 = 6300, n1 = 210 for a 30 s clip); the clip AS IT LIES is an [n1][n2]
// row-major matrix of samples x[n2 k1 + k2], and the transform runs its column stage first:
//   fwd_cols_q : G[q1][k2] = sum_k1 wq[(q1 k1) mod n1] pcm[n2 k1 + k2] for the rows q1 <= n1 / 2 (the rest are their
//                conjugates).  int16 samples times 23-bit fixed-point twiddles: an exact integer sum, so it runs on
//                v_mfma_i32_32x32x32_i8 as six digit products (samples: two bytes; twiddles: three balanced base-256
//                digits) -- a dense [2 hq x n1] . [n1 x n2] contraction per clip that needs no summation order.  The
//                epilogue rounds the integers ONCE to f32.
//   fwd_rows2  : one workgroup per row q1: times the twiddles between the stages on the way in (formed from every fourth
//                one, 13 KB per row instead of 50), FFT_n2 in LDS (fft_rows.h), of which only the outputs q2 that hold consumed
//                bins (a tenth) are stored, for the row itself and, conjugated, for its mirror n1 - q1.
// Nothing is re-laid-out on the way: the first kernel reads the PCM where the caller put it, the second writes the
// consumed bins.  (Clip lengths with a prime factor above 7: k_bluestein.hip.)
#include <cstdlib>

#include "kernels.h"

namespace hpfw {

extern __shared__ __align__(16) unsigned char smem_raw[];

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// ---- column stage ---------------------------------------------------------------------------------------------
// Two kernels: n1 <= 224 (clips up to 32 s at n2 = 6300: all k1 of a column fit the registers of a lane) runs
// fwd_cols_q3_kernel further down; longer clips run the kernel of this section, which stages the samples in LDS chunk
// by chunk of 224 k1.  Digits, accumulators, image layout and epilogue are common.
//
// One workgroup = 4 waves = 128 columns k2 of one clip x up to 7 row tiles (224 rows (Re, Im interleaved) = 112 q1);
// a wave owns 32 columns.  K = k1 in chunks of 224 samples (7 matrix-instruction steps): the samples of the chunk sit
// in LDS as two byte planes [16 k1][column][16 bytes] -- the transposition the matrix instruction's operand layout asks for (a lane
// holds 16 consecutive k1 of one column) happens on the way in, with byte permutes -- the twiddle digits of one row
// tile and chunk (21 KB) next to them.  79 KB of LDS: two workgroups per CU, one's staging and epilogue under the
// other's matrix instructions.
//
// Sample digits: x = 256 hi + lo + 128 with hi = x >> 8 (the high byte as it is) and lo = (x & 255) - 128 (the low
// byte with its top bit flipped), both in [-128, 127]; the constant 128 adds 128 sum_k1 wq[(q1 k1) mod n1] to every
// element of row q1, an exact integer the epilogue adds back (ColsQArgs::corr).
// Digit products of equal weight share an accumulator: c = i + j for sample digit i (0 lo, 1 hi) and twiddle digit j;
// |acc| <= 2 n1 2^14 < 2^31 for every admitted n1; G = sum_c acc_c 2^(8c) + corr, formed in double (exact), rounded
// once to f32.
constexpr int kCqThreads = 256;
constexpr int kCqCols = 128;                      // columns per workgroup
constexpr int kCqKSteps = 7;                      // matrix-instruction steps (of 32 samples) per chunk
constexpr int kCqKChunk = 32 * kCqKSteps;         // 224 samples k1 per chunk
constexpr int kCqTilesPerGroup = 7;               // row tiles per workgroup
constexpr int kCqPlaneBytes = kCqCols * kCqKChunk;            // 28 672
constexpr int kCqABytes = kCqKSteps * 3 * 1024;               // 21 504
constexpr int kCqLdsBytes = 2 * kCqPlaneBytes + kCqABytes;    // 78 848

// Layout of a plane: [unit j of 16 samples k1][slot of a column][16 bytes], 2 KB per unit.  Inside every block of 32
// columns the slot is a bit permutation of the column, pi(c) = (c2 ^ c0) | c3 << 1 | c4 << 2 | c1 << 3 | c0 << 4
// (c_i the bits of c), chosen so that BOTH access patterns are free of bank conflicts: the operand reads (ds_read_b128:
// the 16 lanes of a group hold columns of one parity class of (c4, c3, c2), which pi sends to 16 different slots mod 16)
// and the staging writes below (ds_write_b32 from lanes = 8 groups of 4 columns x 4 consecutive sample quads: for a
// fixed column inside its group pi mod 8 runs through 0..7).
__device__ __forceinline__ int cq_slot(int c)
{
    const int l = c & 31;
    return (c & ~31) | (((l >> 2) ^ l) & 1) | (((l >> 3) & 1) << 1) | (((l >> 4) & 1) << 2) | (((l >> 1) & 1) << 3) | ((l & 1) << 4);
}
// byte address of samples k1 = 4 rq .. 4 rq + 3 of column c inside a plane
__device__ __forceinline__ int cq_b_addr(int c, int rq) { return (rq >> 2) * (kCqCols * 16) + cq_slot(c) * 16 + 4 * (rq & 3); }

// four rows of two columns (one dword per row: [lo0 hi0 lo1 hi1] after the low bytes' top bits were flipped) ->
// per column and plane one dword of four consecutive k1
__device__ __forceinline__ void cq_store_pair(unsigned char *planes, int c, int rq, unsigned r0, unsigned r1, unsigned r2, unsigned r3)
{
    const unsigned x01 = __builtin_amdgcn_perm(r1, r0, 0x05010400u), x23 = __builtin_amdgcn_perm(r3, r2, 0x05010400u);
    const unsigned y01 = __builtin_amdgcn_perm(r1, r0, 0x07030602u), y23 = __builtin_amdgcn_perm(r3, r2, 0x07030602u);
    const int a0 = cq_b_addr(c, rq), a1 = cq_b_addr(c + 1, rq);
    *reinterpret_cast<unsigned *>(planes + a0) = __builtin_amdgcn_perm(x23, x01, 0x05040100u);                 // lo plane
    *reinterpret_cast<unsigned *>(planes + kCqPlaneBytes + a0) = __builtin_amdgcn_perm(x23, x01, 0x07060302u); // hi plane
    *reinterpret_cast<unsigned *>(planes + a1) = __builtin_amdgcn_perm(y23, y01, 0x05040100u);
    *reinterpret_cast<unsigned *>(planes + kCqPlaneBytes + a1) = __builtin_amdgcn_perm(y23, y01, 0x07060302u);
}

// The samples of chunk kc (224 k1 x 128 columns) from the PCM as it lies into the two planes.  A lane takes 4 columns
// x 4 consecutive k1; the 32 lanes of a half wave = 8 column groups (64 contiguous bytes of a row: one sector) x 4
// sample quads (one unit of 16 k1).  LOADW: samples per global load (4: n2 a multiple of 4 and the PCM 8-byte aligned;
// 2: n2 even, 4-byte aligned; 1 otherwise).
template <int LOADW>
__device__ __forceinline__ void cq_stage_samples(const ColsQArgs &a, const int16_t *__restrict__ clip_pcm, int col0, int kc,
                                                 unsigned char *planes, int tid)
{
    const int lane = tid & 63, wave = tid >> 6;
    const int cg = lane & 7, sub = (lane >> 3) & 3, half = lane >> 5;
    const int row_last = a.n1 - 1;
    constexpr int kItems = (kCqKChunk / 16) * (kCqCols / 32); // (unit, block of 32 columns) per half wave: 56
    constexpr int kPer = kItems / 8;                          // 7 per lane
    // columns past the clip's: any valid address (their results are not stored); a group never straddles the end when
    // LOADW divides n2
    const int last = a.n2 - 1;
    unsigned r[kPer][4][2];
    // every load of the lane is issued before the first value is used: one memory latency instead of seven
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
        const int it = 2 * wave + half + 8 * u;
        const int j = it >> 2, blk = it & 3;
        const int rq = 4 * j + sub;
        const int gc = col0 + 32 * blk + 4 * cg;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int row = kc * kCqKChunk + 4 * rq + i;
            if (row > row_last) row = row_last;      // samples past n1 meet zero twiddle digits: any valid address
            const int16_t *rowp = clip_pcm + (int64_t)row * a.n2;
            if (LOADW == 4) {
                const uint2 v = *reinterpret_cast<const uint2 *>(rowp + (gc + 3 <= last ? gc : last - 3));
                r[u][i][0] = v.x;
                r[u][i][1] = v.y;
            } else if (LOADW == 2) {
                r[u][i][0] = *reinterpret_cast<const unsigned *>(rowp + (gc + 1 <= last ? gc : last - 1));
                r[u][i][1] = *reinterpret_cast<const unsigned *>(rowp + (gc + 3 <= last ? gc + 2 : last - 1));
            } else {
                const int c0 = gc <= last ? gc : last, c1 = gc + 1 <= last ? gc + 1 : last, c2 = gc + 2 <= last ? gc + 2 : last,
                          c3 = gc + 3 <= last ? gc + 3 : last;
                r[u][i][0] = (unsigned)(unsigned short)rowp[c0] | ((unsigned)(unsigned short)rowp[c1] << 16);
                r[u][i][1] = (unsigned)(unsigned short)rowp[c2] | ((unsigned)(unsigned short)rowp[c3] << 16);
            }
        }
    }
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
        const int it = 2 * wave + half + 8 * u;
        const int j = it >> 2, blk = it & 3;
        const int rq = 4 * j + sub;
        const int c = 32 * blk + 4 * cg;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            r[u][i][0] ^= 0x00800080u;               // lo = (x & 255) - 128: the low bytes' top bits flipped
            r[u][i][1] ^= 0x00800080u;
        }
        cq_store_pair(planes, c, rq, r[u][0][0], r[u][1][0], r[u][2][0], r[u][3][0]);
        cq_store_pair(planes, c + 2, rq, r[u][0][1], r[u][1][1], r[u][2][1], r[u][3][1]);
    }
}

// the barriers of the column kernel order LDS traffic only: no wait for the global loads and stores in flight (the next
// tile's twiddle digits, the epilogue's table values, the previous tile's results), which __syncthreads() would drain
__device__ __forceinline__ void cq_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// z: the stage's output, PLANAR: [clip][q1][Re row, Im row][n2] floats -- with the samples as the matrix instruction's A operand
// (tile row = column k2) and the twiddle digits as B (tile column = output row), a lane ends up with ONE output row (Re or
// Im of one q1) and four runs of four consecutive k2: every global access of the epilogue is 16 bytes wide.
// SMALL: n1 <= 255, the digit-product sums stay below 2^23 and pair up in int32 (acc_0 + 2^8 acc_1, acc_2 + 2^8 acc_3)
// before the conversion.  HPFW_COLS_STAMPS (diagnostic builds only): cycle stamps per phase, tools/cols_stamps.py.
#ifdef HPFW_COLS_STAMPS
#define CQ_STAMP(k) stamp(k)
#else
#define CQ_STAMP(k) ((void)0)
#endif

template <int LOADW, bool SMALL>
__global__ __launch_bounds__(kCqThreads, 2) void fwd_cols_q_kernel(ColsQArgs a, const int16_t *__restrict__ pcm, int64_t clip_samples,
                                                                   float *__restrict__ z)
{
    unsigned char *planes = smem_raw;                                       // [2 planes][14 units][128 column slots][16 bytes]
    v4i *abuf = reinterpret_cast<v4i *>(smem_raw + 2 * kCqPlaneBytes);      // [7 steps][3 digits][64 lanes]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, nl = lane & 31;
    const int clip = blockIdx.x;                 // the fast index: workgroups resident together share a column block, hence
    const int col0 = blockIdx.y * kCqCols;       // the same slices of the twiddle tables in L2
    const int mt0 = blockIdx.z * kCqTilesPerGroup;
    const int mt1 = min(a.mt, mt0 + kCqTilesPerGroup);
    const int n_chunks = (a.ks + kCqKSteps - 1) / kCqKSteps;
    const int16_t *clip_pcm = pcm + (int64_t)clip * clip_samples;
    const v4i *image = static_cast<const v4i *>(a.image);
    constexpr int kAPer = (kCqABytes / 16 + kCqThreads - 1) / kCqThreads; // 6 pieces of 16 bytes per thread (the last partly)
    v4i areg[kAPer];
    auto load_a = [&](int mt, int kc) {          // the twiddle digits of (row tile, chunk): global -> registers
        const int steps = min(kCqKSteps, a.ks - kc * kCqKSteps);
        const v4i *src = image + ((int64_t)mt * a.ks + kc * kCqKSteps) * 3 * 64;
#pragma unroll
        for (int e = 0; e < kAPer; ++e) {
            const int i = tid + e * kCqThreads;
            areg[e] = i < steps * 3 * 64 ? src[i] : v4i{0, 0, 0, 0};
        }
    };
    auto store_a = [&]() {
#pragma unroll
        for (int e = 0; e < kAPer; ++e) {
            const int i = tid + e * kCqThreads;
            if (i < kCqABytes / 16) abuf[i] = areg[e];
        }
    };
#ifdef HPFW_COLS_STAMPS
    long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tprev = __builtin_amdgcn_s_memtime();
    auto stamp = [&](int k) {
        const long long t = __builtin_amdgcn_s_memtime();
        st[k] += t - tprev;
        tprev = t;
    };
#endif
    load_a(mt0, 0);
    if (n_chunks == 1) cq_stage_samples<LOADW>(a, clip_pcm, col0, 0, planes, tid); // the usual case: staged once per workgroup
    CQ_STAMP(0);
    const int s_base = cq_slot(wave * 32 + nl) * 16; // A operand: this lane's column (tile row) inside the block
    const int cbase = col0 + wave * 32 + 4 * h;      // D: registers 4 g .. 4 g + 3 are columns cbase + 8 g + (0..3), tile column = lane & 31
    const bool vec4 = (a.n2 & 3) == 0;               // then the rows of z and of the twiddle planes start 16-byte aligned
    for (int mt = mt0; mt < mt1; ++mt) {
        // this lane's output row, and what the epilogue adds
        const int row = 32 * mt + nl, q1 = row >> 1;
        const bool live = q1 < a.hq;
        const double corr = a.corr[2 * (live ? q1 : a.hq - 1) + (row & 1)];
        CQ_STAMP(1);
        v16i acc[4];
        for (int kc = 0; kc < n_chunks; ++kc) {
            cq_lds_barrier();                    // everybody has read the previous twiddle digits (and sample planes)
            CQ_STAMP(2);
            if (n_chunks > 1) cq_stage_samples<LOADW>(a, clip_pcm, col0, kc, planes, tid);
            store_a();
            CQ_STAMP(3);
            cq_lds_barrier();
            CQ_STAMP(4);
            {                                    // the next (tile, chunk)'s digits are on their way during the products
                int nmt = mt, nkc = kc + 1;
                if (nkc == n_chunks) {
                    nkc = 0;
                    ++nmt;
                }
                if (nmt < mt1) load_a(nmt, nkc);
            }
            const int steps = min(kCqKSteps, a.ks - kc * kCqKSteps);
            // operands of step s + 1 are read while the products of step s run: two register sets in turn
            v4i w[2][3], x[2][2];
            auto fetch = [&](int set, int s) {
#pragma unroll
                for (int d = 0; d < 3; ++d) w[set][d] = abuf[(s * 3 + d) * 64 + lane];
                const int off = s_base + (2 * s + h) * (kCqCols * 16);
                x[set][0] = *reinterpret_cast<const v4i *>(planes + off);
                x[set][1] = *reinterpret_cast<const v4i *>(planes + kCqPlaneBytes + off);
            };
            auto mult = [&](int set, bool first) {
                // sample digit i (0 lo, 1 hi) times twiddle digit j goes to accumulator i + j; the very first products of a
                // tile start from the instruction's zero operand instead of a cleared register
                const v16i zero = v16i{0};
                acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(x[set][0], w[set][0], first ? zero : acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(x[set][0], w[set][1], first ? zero : acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_i32_32x32x32_i8(x[set][0], w[set][2], first ? zero : acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_i32_32x32x32_i8(x[set][1], w[set][2], first ? zero : acc[3], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(x[set][1], w[set][0], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_i32_32x32x32_i8(x[set][1], w[set][1], acc[2], 0, 0, 0);
            };
            fetch(0, 0);
            if (steps > 1) fetch(1, 1);
            if (kc == 0) mult(0, true); else mult(0, false);
#pragma unroll 1
            for (int s = 1; s + 1 < steps; s += 2) {
                fetch(0, s + 1);
                mult(1, false);
                if (s + 2 < steps) fetch(1, s + 2);
                mult(0, false);
            }
            if ((steps & 1) == 0) mult(1, false);
            CQ_STAMP(5);
        }
        // D[tile row = column][tile column = output row]: this lane holds output row `row`, register r = column
        // cbase + 8 (r >> 2) + (r & 3).  G = sum_c acc_c 2^(8c) + corr, every term and partial sum exact in double; rounded once.
        float gm[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            double g;
            if (SMALL) {
                const int lo = acc[0][r] + (acc[1][r] << 8), hi = acc[2][r] + (acc[3][r] << 8); // |.| < 2^31: n1 <= 255
                g = __builtin_fma((double)hi, 65536.0, (double)lo) + corr;
            } else {
                g = corr;
#pragma unroll
                for (int c = 3; c >= 0; --c) g = __builtin_fma((double)acc[c][r], (double)(1 << (8 * c)), g);
            }
            gm[r] = (float)g;
        }
        CQ_STAMP(6);
        if (live) {
            // (blocked layout: this workgroup's column block, row `row`, kZBlock floats per row)
            float *zrow = z + (int64_t)clip * a.zclip + ((int64_t)blockIdx.y * 2 * a.hq + row) * kZBlock - col0;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = cbase + 8 * g;
                if (vec4 && c + 3 < a.n2) {
                    *reinterpret_cast<float4 *>(zrow + c) = float4{gm[4 * g], gm[4 * g + 1], gm[4 * g + 2], gm[4 * g + 3]};
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (c + e < a.n2) zrow[c + e] = gm[4 * g + e];
                }
            }
        }
        CQ_STAMP(7);
    }
#ifdef HPFW_COLS_STAMPS
    if (a.stamps && tid == 0) {
        long long *o = a.stamps + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = st[k];
    }
#endif
}

// ---- column stage, n1 <= 224: three workgroups per CU ----------------------------------------------------------
// The same contraction with the sample digits held in REGISTERS: a wave's 32 columns x 224 samples k1 are 56 registers
// per lane (7 steps x 2 planes x 16 bytes), read once per workgroup through a 2 KB scratch of the wave's own (the
// transposition with byte permutes as above, 32 rows at a time; no workgroup barrier: a wave's LDS traffic is in
// order).  What is left in LDS are the twiddle digits of a row tile, fetched by global_load_lds (no registers) into
// one of two buffers while the results of the previous tile are converted and stored: 43 KB + 8 KB per workgroup
// and at most 168 registers, so three workgroups = twelve waves share a CU, one barrier (behind a full vmcnt(0)) per tile.
constexpr int kCq3ScratchBytes = 2 * 2 * 32 * 16;                         // per wave: [plane][unit][slot][16 bytes]
constexpr int kCq3CorrRows = 256;                                         // row corrections of every tile (n1 <= 224: at most 226 rows), doubles
constexpr int cq3_lds_bytes(int waves) { return 2 * kCqABytes + waves * kCq3ScratchBytes + kCq3CorrRows * 8; }

__device__ __forceinline__ int cq3_w_addr(int c, int rq) { return (rq >> 2) * 512 + cq_slot(c) * 16 + 4 * (rq & 3); }

__device__ __forceinline__ void cq3_store_pair(unsigned char *sc, int c, int rq, unsigned r0, unsigned r1, unsigned r2, unsigned r3)
{
    const unsigned x01 = __builtin_amdgcn_perm(r1, r0, 0x05010400u), x23 = __builtin_amdgcn_perm(r3, r2, 0x05010400u);
    const unsigned y01 = __builtin_amdgcn_perm(r1, r0, 0x07030602u), y23 = __builtin_amdgcn_perm(r3, r2, 0x07030602u);
    const int a0 = cq3_w_addr(c, rq), a1 = cq3_w_addr(c + 1, rq);
    *reinterpret_cast<unsigned *>(sc + a0) = __builtin_amdgcn_perm(x23, x01, 0x05040100u);        // lo plane
    *reinterpret_cast<unsigned *>(sc + 1024 + a0) = __builtin_amdgcn_perm(x23, x01, 0x07060302u); // hi plane
    *reinterpret_cast<unsigned *>(sc + a1) = __builtin_amdgcn_perm(y23, y01, 0x05040100u);
    *reinterpret_cast<unsigned *>(sc + 1024 + a1) = __builtin_amdgcn_perm(y23, y01, 0x07060302u);
}

// the samples of the wave's 32 columns (from column cw0; n1 <= 224) into x[step][plane]: a lane loads 4 columns x 4 consecutive k1 per
// step (the 64 lanes of a load instruction: 8 rows x 64 contiguous bytes), every load issued before the first is used
template <int LOADW, class F>
__device__ __forceinline__ void cq3_load_samples(const ColsQArgs &a, const int16_t *__restrict__ clip_pcm, int cw0,
                                                 unsigned char *sc, int lane, v4i (&x)[kCqKSteps][2], F issued)
{
    const int cg = lane & 7, quad = lane >> 3;
    const int row_last = a.n1 - 1, last = a.n2 - 1;
    const int gc = cw0 + 4 * cg;
    // byte offsets inside the clip in 32 bits (a clip is below 2^31 samples): one register per address, the clip's base
    // in scalar registers
    const char *base = reinterpret_cast<const char *>(clip_pcm);
    const unsigned pitch = 2u * (unsigned)a.n2;
    unsigned cb[4];
    if (LOADW == 4) {
        cb[0] = 2u * (unsigned)(gc + 3 <= last ? gc : last - 3);
    } else if (LOADW == 2) {
        cb[0] = 2u * (unsigned)(gc + 1 <= last ? gc : last - 1);
        cb[1] = 2u * (unsigned)(gc + 3 <= last ? gc + 2 : last - 1);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) cb[e] = 2u * (unsigned)(gc + e <= last ? gc + e : last);
    }
    // LOADW = 4: every load of the chunk is issued before the first value is used (one memory latency instead of seven);
    // the narrower loads of odd shapes go step by step, or their addresses and values would not fit the registers
    constexpr int kBatch = LOADW == 4 ? kCqKSteps : 1;
    const int h = lane >> 5, nl = lane & 31;
    const int rd = h * 512 + cq_slot(nl) * 16;
#pragma unroll
    for (int s0 = 0; s0 < kCqKSteps; s0 += kBatch) {
    unsigned r[kBatch][4][2];
#pragma unroll
    for (int sb = 0; sb < kBatch; ++sb) {
        const int s = s0 + sb;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int row = 32 * s + 4 * quad + i;
            if (row > row_last) row = row_last;      // samples past n1 meet zero twiddle digits: any valid address
            const unsigned ro = (unsigned)row * pitch;
            if (LOADW == 4) {
                const uint2 v = *reinterpret_cast<const uint2 *>(base + (ro + cb[0]));
                r[sb][i][0] = v.x;
                r[sb][i][1] = v.y;
            } else if (LOADW == 2) {
                r[sb][i][0] = *reinterpret_cast<const unsigned *>(base + (ro + cb[0]));
                r[sb][i][1] = *reinterpret_cast<const unsigned *>(base + (ro + cb[1]));
            } else {
                r[sb][i][0] = (unsigned)*reinterpret_cast<const unsigned short *>(base + (ro + cb[0])) |
                              ((unsigned)*reinterpret_cast<const unsigned short *>(base + (ro + cb[1])) << 16);
                r[sb][i][1] = (unsigned)*reinterpret_cast<const unsigned short *>(base + (ro + cb[2])) |
                              ((unsigned)*reinterpret_cast<const unsigned short *>(base + (ro + cb[3])) << 16);
            }
        }
    }
    if (s0 == 0) issued();                         // what the caller wants in flight behind the first loads
#pragma unroll
    for (int sb = 0; sb < kBatch; ++sb) {
        const int s = s0 + sb;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            r[sb][i][0] ^= 0x00800080u;              // lo = (x & 255) - 128: the low bytes' top bits flipped
            r[sb][i][1] ^= 0x00800080u;
        }
        cq3_store_pair(sc, 4 * cg, quad, r[sb][0][0], r[sb][1][0], r[sb][2][0], r[sb][3][0]);
        cq3_store_pair(sc, 4 * cg + 2, quad, r[sb][0][1], r[sb][1][1], r[sb][2][1], r[sb][3][1]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the wave's own writes, in order before its reads
        x[s][0] = *reinterpret_cast<const v4i *>(sc + rd);
        x[s][1] = *reinterpret_cast<const v4i *>(sc + 1024 + rd);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // read before the next step's writes land
    }
    }
}

template <int LOADW, int kCq3Waves>                  // a wave = 32 columns
__global__ __launch_bounds__(64 * kCq3Waves, 3) void fwd_cols_q3_kernel(ColsQArgs a, const int16_t *__restrict__ pcm, int64_t clip_samples,
                                                                    float *__restrict__ z)
{
    constexpr int kCq3Cols = 32 * kCq3Waves;
    unsigned char *abytes = smem_raw;                                       // [2 buffers][7 steps][3 digits][64 lanes][16 bytes]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned char *sc = smem_raw + 2 * kCqABytes + wave * kCq3ScratchBytes;
    const int h = lane >> 5, nl = lane & 31;
    // Workgroups go to the eight XCDs in turn (id mod 8), each with an L2 of its own: an XCD takes a contiguous run of
    // (clip, column block) pairs with the column block fastest, so that the 128-byte lines of a PCM row that two
    // neighbouring column blocks share (a row is 12 600 bytes: no block starts on a line) are fetched from HBM once.
    const int ncb = (a.n2 + kCq3Cols - 1) / kCq3Cols;
    const unsigned per_xcd = (gridDim.x + 7) / 8;
    const unsigned t = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (t >= (unsigned)(a.n_clips * ncb)) return;
    const int cb = t % ncb, clip = t / ncb;
    const int col0 = cb * kCq3Cols;
    const int16_t *clip_pcm = pcm + (int64_t)clip * clip_samples;
    const v4i *image = static_cast<const v4i *>(a.image);
    const int steps = a.ks, pieces = 3 * steps;      // <= 7 steps: every sample of the columns is in registers
    // the twiddle digits of row tile mt into buffer `buf`: 21 pieces of 1 KB, every wave its share
    auto issue_a = [&](int mt, int buf) {
        const v4i *src = image + (int64_t)mt * a.ks * 3 * 64 + lane;
#pragma unroll
        for (int e = 0; e < (kCqKSteps * 3 + kCq3Waves - 1) / kCq3Waves; ++e) {
            const int p = wave + kCq3Waves * e;
            if (p < pieces)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + p * 64),
                                                 (__attribute__((address_space(3))) void *)(abytes + buf * kCqABytes + p * 1024),
                                                 16, 0, 0);
        }
    };
#ifdef HPFW_COLS_STAMPS
    long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tprev = __builtin_amdgcn_s_memtime();
    auto stamp = [&](int k) {
        const long long t = __builtin_amdgcn_s_memtime();
        st[k] += t - tprev;
        tprev = t;
    };
#endif
    v4i x[kCqKSteps][2];
    // the samples first (they come from HBM), the first tile's digits (from L2) behind them
    cq3_load_samples<LOADW>(a, clip_pcm, col0 + wave * 32, sc, lane, x, [&] { issue_a(0, 0); });
    CQ_STAMP(0);
    // D[tile row = output row][tile column = column]: the twiddle digits are the A operand and the samples the B operand,
    // so a lane holds ONE column (col0 + 32 wave + (lane & 31)) and its register r the output row
    // 32 mt + 8 (r >> 2) + (r & 3) + 4 h; a 4 x 4 exchange inside the quads of lanes then gives every lane four consecutive
    // columns of one row, so that a store instruction writes eight whole 128-byte lines, 16 bytes per lane.  (Until round 4
    // the operands were the other way round -- a lane held a row and wrote 16-byte pieces of it, 32 lines touched per
    // instruction: the vector-memory path, not HBM, bound the kernel.)
    const int c = col0 + wave * 32 + nl;
    const bool c_in = c < a.n2;
    const bool vec4 = (a.n2 & 3) == 0;               // rows of z then start 16-byte aligned and columns come in whole fours
    // z [clip][column block cb][row = 2 q1 + (Re: 0, Im: 1)][kZBlock]: this workgroup's block, the wave's 32 columns of it
    static_assert(32 * kCq3Waves == kZBlock, "a workgroup's columns are one block of z");
    float *zblk = z + (int64_t)clip * a.zclip + (int64_t)cb * 2 * a.hq * kZBlock + wave * 32;
    const int rows_live = 2 * a.hq;
    // what the samples' +128 digit offset adds to every element of a row, for the rows of every tile: in LDS, 16 per lane and tile
    double *corr_lds = reinterpret_cast<double *>(smem_raw + 2 * kCqABytes + kCq3Waves * kCq3ScratchBytes);
    for (int i = tid; i < kCq3CorrRows; i += 64 * kCq3Waves) corr_lds[i] = a.corr[i < rows_live ? i : rows_live - 1];
    for (int mt = 0; mt < a.mt; ++mt) {
        // this wave's pieces of the tile's digits have landed; after the barrier everybody's have, and everybody is done
        // with the other buffer.  A full wait: loads and stores share the counter and are not guaranteed to retire in
        // order with respect to each other, so "at most four outstanding" (the previous tile's stores) would not prove
        // that the older global_load_lds are done.  (Round 3 blamed a counted wait here for wrong hashprints with two
        // processes on the GPU; round 4 found that cause elsewhere -- packed FP32 in the row stage, device_math.h -- and
        // this kernel's output was never wrong in any of the runs that localised it: tools/rows_error_shape.py.)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        CQ_STAMP(1);
        const v4i *wb = reinterpret_cast<const v4i *>(abytes + (mt & 1) * kCqABytes) + lane;
        v16i acc[4];
        const v16i zero = v16i{0};
        auto step = [&](int s) {
            // sample digit i (0 lo, 1 hi) times twiddle digit j goes to accumulator i + j; the first products of a tile
            // start from the instruction's zero operand instead of a cleared register
            const v4i w0 = wb[(s * 3 + 0) * 64], w1 = wb[(s * 3 + 1) * 64], w2 = wb[(s * 3 + 2) * 64];
            acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(w0, x[s][0], s == 0 ? zero : acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(w1, x[s][0], s == 0 ? zero : acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_i32_32x32x32_i8(w2, x[s][0], s == 0 ? zero : acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_i32_32x32x32_i8(w2, x[s][1], s == 0 ? zero : acc[3], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(w0, x[s][1], acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_i32_32x32x32_i8(w1, x[s][1], acc[2], 0, 0, 0);
        };
        if (steps == kCqKSteps) {                // the usual case, straight through: the operand reads run ahead of the products
#pragma unroll
            for (int s = 0; s < kCqKSteps; ++s) step(s);
        } else {
#pragma unroll
            for (int s = 0; s < kCqKSteps; ++s)
                if (s < steps) step(s);
        }
        CQ_STAMP(2);
        // the next tile's digits, on their way under the conversion and the stores (no wait for a vector-memory LOAD may
        // follow while they are in flight -- the compiler would make it a wait for everything: the row corrections come from LDS)
        CQ_STAMP(3);
        if (mt + 1 < a.mt) issue_a(mt + 1, (mt + 1) & 1);
        asm volatile("" ::: "memory");
        CQ_STAMP(4);
        const int row0 = 32 * mt + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            // G = sum_c acc_c 2^(8c) + corr: n1 <= 224, so the digit-product sums pair up in int32 (|.| < 2^31) and the rest
            // is exact in double; rounded once.  Four rows (registers 4 g .. 4 g + 3) at a time: few live registers.
            float q[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * g + e;
                const int lo = acc[0][r] + (acc[1][r] << 8), hi = acc[2][r] + (acc[3][r] << 8);
                q[e] = (float)(__builtin_fma((double)hi, 65536.0, (double)lo) + corr_lds[row0 + 8 * g + e]);
            }
            if (vec4) {
                // 4 x 4 transposition inside every quad of lanes (two exchange steps on the data-parallel-primitive path):
                // lane 4 i + j then holds row 8 g + j (+ 4 h), columns 4 i .. 4 i + 3 of the wave's 32 -- one 16-byte store, and
                // a store instruction covers eight whole 128-byte lines
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {     // lanes differing in bit 0 exchange registers differing in bit 0
                    const float send = (lane & 1) ? q[2 * pr] : q[2 * pr + 1];
                    const float got = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, send), 0xB1, 0xF, 0xF, true));
                    if (lane & 1) q[2 * pr] = got; else q[2 * pr + 1] = got;
                }
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {     // bit 1
                    const float send = (lane & 2) ? q[pr] : q[pr + 2];
                    const float got = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, send), 0x4E, 0xF, 0xF, true));
                    if (lane & 2) q[pr] = got; else q[pr + 2] = got;
                }
                const int row = row0 + 8 * g + (nl & 3);
                const int c4 = col0 + wave * 32 + (nl & ~3);
                if (c4 < a.n2 && row < rows_live) *reinterpret_cast<float4 *>(zblk + row * kZBlock + (nl & ~3)) = float4{q[0], q[1], q[2], q[3]};
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int row = row0 + 8 * g + e;
                    if (c_in && row < rows_live) zblk[row * kZBlock + nl] = q[e];
                }
            }
        }
        CQ_STAMP(6);
    }
#ifdef HPFW_COLS_STAMPS
    if (a.stamps && tid == 0) {
        long long *o = a.stamps + (int64_t)blockIdx.x * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = st[k];
    }
#endif
}

#ifdef HPFW_COLS_STAMPS
long long *g_cols_stamps = nullptr;
extern "C" long long *hpfw_debug_cols_stamps() { return g_cols_stamps; }
#endif

// ---- row stage ------------------------------------------------------------------------------------------------
// one workgroup = one row q1 of one clip; the body (fft_rows.h) is shared with tests/emu.
// 512 threads x 128 VGPRs and 50 KB of LDS: two workgroups per CU, every pass one butterfly per thread.
constexpr int kFwdThreads = 512;
#ifndef HPFW_ROWS_WAVES
#define HPFW_ROWS_WAVES 6 // waves per SIMD the compile-time sequence is held to (6: three workgroups per CU)
#endif

template <class Groups, int WAVES>
__global__ __launch_bounds__(kFwdThreads, WAVES) void fwd_rows2_kernel(RowsArgs a, Rows2Out o, const float *__restrict__ z,
                                                                   cf *__restrict__ x)
{
    cf *lds = reinterpret_cast<cf *>(smem_raw);
    // clip is the fast grid index: workgroups resident at the same time run the same row, whose butterfly twiddles
    // they share in L2
    const int clip = blockIdx.x, q1 = blockIdx.y;
    rows2_body<Groups>(lds, a, kFwdThreads, z + (int64_t)clip * o.zclip + (int64_t)2 * q1 * o.zrow, q1, o, x + (int64_t)clip * o.n1 * o.q2w);
}

__global__ __launch_bounds__(256) void gather_bins_kernel(CqPlanDev cp, const cf *__restrict__ x, cf *__restrict__ out)
{
    const int clip = blockIdx.y;
    const XsView v = cp.view(x, clip);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < cp.nk; i += gridDim.x * 256) out[(int64_t)clip * cp.nk + i] = v(cp.kmin + i);
}

template <int LOADW>
static void launch_cols_q_t(ColsQArgs a, const int16_t *d_pcm, int64_t clip_samples, dim3 grid, float *d_z, hipStream_t s)
{
    static PerDeviceOnce attr_set;
    if (attr_set.need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(fwd_cols_q_kernel<LOADW, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kCqLdsBytes);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(fwd_cols_q_kernel<LOADW, false>), hipFuncAttributeMaxDynamicSharedMemorySize, kCqLdsBytes);
        attr_set.mark();
    }
    if (a.ks <= kCqKSteps && !(a.variant & 1)) {  // n1 <= 224: the register-resident kernel, a one-dimensional grid
        a.n_clips = grid.x;
        // (workgroups of six waves = 192 columns, two per CU: a third less of the digit traffic per column, the kernel
        // itself 2.63 -> 2.9 ms, the step within 0.6 %: DESIGN.md section 9)
        constexpr int kWaves = 4;
        const int ncb = (a.n2 + 32 * kWaves - 1) / (32 * kWaves);
        hipLaunchKernelGGL((fwd_cols_q3_kernel<LOADW, kWaves>), dim3(8 * ((grid.x * ncb + 7) / 8)), dim3(64 * kWaves), cq3_lds_bytes(kWaves), s,
                           a, d_pcm, clip_samples, d_z);
        return;
    }
    if (a.n1 <= 255)
        hipLaunchKernelGGL((fwd_cols_q_kernel<LOADW, true>), grid, dim3(kCqThreads), kCqLdsBytes, s, a, d_pcm, clip_samples, d_z);
    else
        hipLaunchKernelGGL((fwd_cols_q_kernel<LOADW, false>), grid, dim3(kCqThreads), kCqLdsBytes, s, a, d_pcm, clip_samples, d_z);
}

void launch_fwd_cols_q(const ColsQArgs &a_in, const int16_t *d_pcm, int64_t clip_samples, int n_clips, float *d_z, hipStream_t s)
{
    if (n_clips <= 0) return;
    ColsQArgs a = a_in;
#ifdef HPFW_COLS_STAMPS
    static long long *d_stamps = nullptr;
    if (!d_stamps) (void)hipMalloc(&d_stamps, (size_t)64 * 1024 * 1024);
    a.stamps = d_stamps;
    g_cols_stamps = d_stamps;
#endif
    dim3 grid(n_clips, (a.n2 + kCqCols - 1) / kCqCols, (a.mt + kCqTilesPerGroup - 1) / kCqTilesPerGroup);
    const uintptr_t addr = reinterpret_cast<uintptr_t>(d_pcm);
    if (a.n2 % 4 == 0 && a.n2 >= 4 && addr % 8 == 0 && clip_samples % 4 == 0)
        launch_cols_q_t<4>(a, d_pcm, clip_samples, grid, d_z, s);
    else if (a.n2 % 2 == 0 && addr % 4 == 0 && clip_samples % 2 == 0)
        launch_cols_q_t<2>(a, d_pcm, clip_samples, grid, d_z, s);
    else
        launch_cols_q_t<1>(a, d_pcm, clip_samples, grid, d_z, s);
}

size_t fwd_rows_lds_bytes(const RowsArgs &a) { return (size_t)a.n2 * sizeof(cf); }

template <class Groups, int WAVES>
static void launch_rows2_t(const RowsArgs &a, const Rows2Out &o, const float *d_z, int n_clips, cf *d_x, hipStream_t s)
{
    static PerDeviceOnce attr_set;
    if (attr_set.need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(fwd_rows2_kernel<Groups, WAVES>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set.mark();
    }
    dim3 grid(n_clips, o.hq);
    hipLaunchKernelGGL((fwd_rows2_kernel<Groups, WAVES>), grid, dim3(kFwdThreads), fwd_rows_lds_bytes(a), s, a, o, d_z, d_x);
}

void launch_fwd_rows2(const RowsArgs &a, const Rows2Out &o, const float *d_z, int n_clips, cf *d_x, hipStream_t s)
{
    if (n_clips <= 0) return;
    // the compile-time sequence runs its last two groups one butterfly per thread
    // (68 registers under the bound of six waves per SIMD: three workgroups share a CU, as the 50 KB of LDS allow --
    // 2.30 -> 2.18 ms per 1000 clips against the 82 registers and two workgroups the compiler settles on by itself)
    if (Groups6300::matches_plan(a) && Groups6300::min_threads(a.n2) <= kFwdThreads)
        launch_rows2_t<Groups6300, HPFW_ROWS_WAVES>(a, o, d_z, n_clips, d_x, s);
    else
        launch_rows2_t<RuntimeGroups, 4>(a, o, d_z, n_clips, d_x, s);
}

void launch_gather_bins(const CqPlanDev &cp, const cf *d_x, int n_clips, cf *d_out, hipStream_t s)
{
    if (n_clips <= 0) return;
    hipLaunchKernelGGL(gather_bins_kernel, dim3((cp.nk + 255) / 256 < 64 ? (cp.nk + 255) / 256 : 64, n_clips), dim3(256), 0, s, cp, d_x, d_out);
}

} // namespace hpfw
