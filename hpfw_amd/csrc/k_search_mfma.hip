// k_search_mfma.hip -- a9 on the matrix cores: the sliding Hamming scan as an exact +-1 contraction.
//
// Same result as hamming_scan_kernel (k_search.hip; reference MemoryStorage::find,
// include/hpfw/audioproblems/live-song-id/storage.h:27-64), bit for bit: with every hashprint bit b
// written as the E2M1 (fp4) number 1 - 2b, sum_j <Q_j, R_{off+j}> = 64 k - 2 sum_j popcount(q[j] ^
// r[off+j]); the products are +-1 and the f32 accumulator holds integers of magnitude <= 64 k < 2^24,
// so v_mfma_scale_f32_32x32x64_f8f6f4 (both operands fp4, unit scales) computes it exactly
// (tools/fp4_probe.hip checks the instruction against popcount).  One MFMA = 32 queries x 32 offsets x
// one hashprint position (K = 64 bits): 1024 pairs in 32 cycles per SIMD, against 4 VALU lane-ops
// per pair for xor/popcount -- the scan is issue-bound, not memory-bound, so this is where the time is.
//
// Layout of the contraction (no diagonal sums, no lane shuffles in the loop):
//   M = 32 queries of a group, N = offsets, K = (position j, bit): the operand B of step j for offset
//   column n is the reference hashprint off + j -- a window of the clip sliding through LDS, one
//   ds_read_b128 per MFMA; the operand A of step j (the 32 queries' hashprint j) is read once per
//   step and used by the wave's 4 offset tiles.  Workgroup = 8 waves = one query group x 1024
//   offsets of one clip; queries shorter than the group's longest are zero-padded (fp4 0 adds
//   nothing), windows past the end of the clip are zeros, and k = min(k, n) (storage.h:37-39) falls
//   out of that: a query longer than the clip meets zeros beyond n.
#include "kernels.h"

#include <algorithm>

namespace hpfw {

extern __shared__ __align__(16) unsigned char smem_raw[];

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kSmThreads = 512;            // 8 waves
constexpr int kSmTiles = 4;                // offset tiles of 32 per wave
constexpr int kSmWaveOffs = 32 * kSmTiles; // 128
constexpr int kSmWgOffs = kSmWaveOffs * (kSmThreads / 64); // 1024 offsets per workgroup
constexpr int kSmChunk = 16;               // positions j per staged chunk of the A operand (16 KB)

// 8 bits -> 8 E2M1 nibbles, bit i in nibble i: 0 -> +1.0 (0x2), 1 -> -1.0 (0xA)
__device__ __forceinline__ uint32_t expand8(uint32_t x)
{
    uint32_t t = (x | (x << 12)) & 0x000F000Fu;
    t = (t | (t << 6)) & 0x03030303u;
    t = (t | (t << 3)) & 0x11111111u;
    return 0x22222222u | (t << 3);
}

__device__ __forceinline__ v4i expand32(uint32_t w)
{
    v4i r;
    r.x = (int)expand8(w & 0xff);
    r.y = (int)expand8((w >> 8) & 0xff);
    r.z = (int)expand8((w >> 16) & 0xff);
    r.w = (int)expand8(w >> 24);
    return r;
}

// qa [n_groups][kt_pad][64 lanes][16 B]: lane (m, h) of step j holds bits [32 h, 32 h + 32) of
// hashprint j of query 32 g + m, or zeros past the query's end / past the last query.
__global__ __launch_bounds__(256) void expand_queries_kernel(const uint64_t *__restrict__ q,
                                                             const int64_t *__restrict__ q_off, int n_q, int kt_pad,
                                                             v4i *__restrict__ qa)
{
    const int g = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x; // (j, lane)
    if (idx >= kt_pad * 64) return;
    const int j = idx >> 6, lane = idx & 63, m = lane & 31, h = lane >> 5;
    const int qi = g * 32 + m;
    v4i v = {0, 0, 0, 0};
    if (qi < n_q) {
        const int64_t o = q_off[qi];
        if (j < (int)(q_off[qi + 1] - o)) v = expand32((uint32_t)(q[o + j] >> (32 * h)));
    }
    qa[((int64_t)g * kt_pad + j) * 64 + lane] = v;
}

// the same image for windows of one or more queries: row w of group w / 32 is the `win` hashprints
// starting at q[w_start[w]] (annoy_storage.h:23,32: 64 consecutive words per item)
__global__ __launch_bounds__(256) void expand_windows_kernel(const uint64_t *__restrict__ q,
                                                             const int64_t *__restrict__ w_start, int n_win, int win,
                                                             int kt_pad, v4i *__restrict__ qa)
{
    const int g = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= kt_pad * 64) return;
    const int j = idx >> 6, lane = idx & 63, m = lane & 31, h = lane >> 5;
    const int w = g * 32 + m;
    v4i v = {0, 0, 0, 0};
    if (w < n_win && j < win) v = expand32((uint32_t)(q[w_start[w] + j] >> (32 * h)));
    qa[((int64_t)g * kt_pad + j) * 64 + lane] = v;
}

struct SearchMfmaArgs {
    const uint64_t *db;
    const int64_t *db_off;
    int n_clips;
    const int64_t *q_off; // [n_q + 1] (device), this launch's first query at q_off[0]
    int n_q;
    const v4i *qa;        // expanded queries of this launch
    int kt_pad;           // rows of qa per group (multiple of kSmChunk)
    const int *gk;        // [2 g]: longest query of group g; [2 g + 1]: its shortest non-empty one
    uint64_t *best;       // [n_q][n_clips], initialised to ~0
    int chunks;           // workgroups (of 1024 offsets) per clip
    // nearest-window mode (KNN): rows are windows of `win` hashprints, every row has length win, and the
    // nn smallest (distance, global position) keys of each row are kept in slots [n_q][8]
    int win, nn;
    unsigned long long *slots;
};

template <bool KNN>
__global__ __launch_bounds__(kSmThreads, 4) void hamming_mfma_kernel(SearchMfmaArgs a)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_lane = lane & 31, h = lane >> 5;
    const int clip = blockIdx.x / a.chunks, g = blockIdx.y;
    const int64_t r0 = a.db_off[clip];
    const int n = (int)(a.db_off[clip + 1] - r0);
    const int kt = KNN ? a.win : a.gk[2 * g];
    if (n <= 0 || kt <= 0) return;
    const int o0 = (blockIdx.x - clip * a.chunks) * kSmWgOffs;
    const int kmin = KNN ? a.win : min(a.gk[2 * g + 1], n);
    if (o0 > n - kmin) return; // no query of the group has an offset in this chunk (off <= n - k)

    const int win = kSmWgOffs + kt;                      // window slots (one spare)
    v4i *ldsB = reinterpret_cast<v4i *>(smem_raw);       // [2][win]: plane h = bits [32 h, 32 h + 32) of each hashprint
                                                         // (lanes of a half-wave read 16 B apart: no bank conflicts)
    v4i *ldsA = ldsB + 2 * win;                          // [2][kSmChunk][64]
    int *kq_s = reinterpret_cast<int *>(ldsA + 2 * kSmChunk * 64); // [32] effective k of the group's queries
    unsigned *red = reinterpret_cast<unsigned *>(kq_s + 32);       // [8 waves][32 queries]

    // the clip's hashprints o0 .. o0 + win - 1, expanded; zeros past the end of the clip
    for (int i = tid; i < win; i += kSmThreads) {
        const int gi = o0 + i;
        v4i lo = {0, 0, 0, 0}, hi = {0, 0, 0, 0};
        if (gi < n) {
            const uint64_t w = a.db[r0 + gi];
            lo = expand32((uint32_t)w);
            hi = expand32((uint32_t)(w >> 32));
        }
        ldsB[i] = lo;
        ldsB[win + i] = hi;
    }
    if (tid < 32) {
        const int qi = g * 32 + tid;
        int k = 0;
        if (KNN) {
            k = qi < a.n_q ? a.win : 0; // a window is never shortened: clips below win hashprints hold no item
        } else {
            if (qi < a.n_q) k = (int)(a.q_off[qi + 1] - a.q_off[qi]);
            k = k < n ? k : n; // storage.h:37-39
        }
        kq_s[tid] = k;
    }
    // first chunk of the A operand
    const v4i *qa = a.qa + (int64_t)g * a.kt_pad * 64;
    const int n_chunks = (kt + kSmChunk - 1) / kSmChunk;
    v4i st0 = qa[tid], st1 = qa[tid + kSmThreads]; // 16 steps x 64 lanes = 1024 entries: two per thread
    ldsA[tid] = st0;
    ldsA[tid + kSmThreads] = st1;
    __syncthreads();

    f32x16 acc[kSmTiles];
#pragma unroll
    for (int t = 0; t < kSmTiles; ++t) acc[t] = f32x16{0};
    const int one = 0x7f7f7f7f; // E8M0 scale 2^0
    const v4i *bp = ldsB + h * win + wave * kSmWaveOffs + n_lane;

    for (int c = 0; c < n_chunks; ++c) {
        const bool more = c + 1 < n_chunks;
        if (more) { // next chunk of A into registers while this one multiplies
            st0 = qa[(c + 1) * (kSmChunk * 64) + tid];
            st1 = qa[(c + 1) * (kSmChunk * 64) + tid + kSmThreads];
        }
        const v4i *ap = ldsA + (c & 1) * (kSmChunk * 64) + lane;
        const int jn = min(kSmChunk, kt - c * kSmChunk);
        const v4i *bj = bp + c * kSmChunk;
        for (int j = 0; j < jn; ++j) {
            const v4i a4 = ap[j * 64];
            const v8i av = {a4.x, a4.y, a4.z, a4.w, 0, 0, 0, 0};
#pragma unroll
            for (int t = 0; t < kSmTiles; ++t) {
                const v4i b4 = bj[j + 32 * t];
                const v8i bv = {b4.x, b4.y, b4.z, b4.w, 0, 0, 0, 0};
                acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc[t], 4, 4, 0, one, 0, one);
            }
        }
        if (more) {
            v4i *dst = ldsA + ((c + 1) & 1) * (kSmChunk * 64);
            dst[tid] = st0;
            dst[tid + kSmThreads] = st1;
        }
        __syncthreads();
    }

    if constexpr (KNN) {
        // the nn nearest windows of each row among this wave's 128 offsets: nn rounds of "smallest
        // (distance, offset) key across the half-wave", each winner offered to the row's global slots --
        // a chain of atomicMin that keeps the nn smallest keys seen by anyone (unsorted; the host sorts)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int m = (reg & 3) + 8 * (reg >> 2) + 4 * h;
            const int k = kq_s[m];
            unsigned cand[kSmTiles];
#pragma unroll
            for (int t = 0; t < kSmTiles; ++t) {
                const int lo = wave * kSmWaveOffs + t * 32 + n_lane;
                const int dist = (64 * k - (int)acc[t][reg]) >> 1;
                cand[t] = (k > 0 && o0 + lo <= n - k) ? (((unsigned)dist << 12) | (unsigned)lo) : 0xffffffffu;
            }
            const int row = g * 32 + m;
            for (int r = 0; r < a.nn; ++r) {
                unsigned mine = cand[0];
#pragma unroll
                for (int t = 1; t < kSmTiles; ++t) mine = cand[t] < mine ? cand[t] : mine;
                unsigned best = mine;
#pragma unroll
                for (int s2 = 16; s2 >= 1; s2 >>= 1) {
                    const unsigned o = (unsigned)__shfl_xor((int)best, s2);
                    best = o < best ? o : best;
                }
                if (best == 0xffffffffu) break; // uniform across the half-wave
#pragma unroll
                for (int t = 0; t < kSmTiles; ++t)
                    if (cand[t] == best) cand[t] = 0xffffffffu; // keys are unique: exactly one lane and tile
                if (n_lane == 0 && row < a.n_q) {
                    unsigned long long *sl = a.slots + (int64_t)row * 8;
                    unsigned long long cur = ((unsigned long long)(best >> 12) << 40) |
                                             (unsigned long long)(r0 + o0 + (int)(best & 0xfff));
                    if (cur < __atomic_load_n(sl + a.nn - 1, __ATOMIC_RELAXED)) {
                        for (int s2 = 0; s2 < a.nn; ++s2) {
                            const unsigned long long old = atomicMin(sl + s2, cur);
                            cur = old > cur ? old : cur; // the larger of the two moves on
                        }
                    }
                }
            }
        }
        return;
    } else {
        // acc[t][reg]: query row m = (reg & 3) + 8 (reg >> 2) + 4 h, offset o0 + wave 128 + t 32 + n_lane.
        // key = dist << 12 | offset inside the workgroup's 1024 (dist <= 64 k < 2^20): smaller distance, then
        // smaller offset -- the first strict minimum of storage.h:50.
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int m = (reg & 3) + 8 * (reg >> 2) + 4 * h;
            const int k = kq_s[m];
            unsigned key = 0xffffffffu;
#pragma unroll
            for (int t = 0; t < kSmTiles; ++t) {
                const int lo = wave * kSmWaveOffs + t * 32 + n_lane;
                const int dist = (64 * k - (int)acc[t][reg]) >> 1;
                const unsigned cand = ((unsigned)dist << 12) | (unsigned)lo;
                if (k > 0 && o0 + lo <= n - k && cand < key) key = cand;
            }
#pragma unroll
            for (int s = 16; s >= 1; s >>= 1) { // minimum over the 32 offset columns of this half-wave
                const unsigned o = (unsigned)__shfl_xor((int)key, s);
                key = o < key ? o : key;
            }
            if (n_lane == 0) red[wave * 32 + m] = key;
        }
        __syncthreads();
        if (tid < 32) {
            unsigned key = red[tid];
            for (int w = 1; w < kSmThreads / 64; ++w) key = red[w * 32 + tid] < key ? red[w * 32 + tid] : key;
            const int qi = g * 32 + tid;
            if (qi < a.n_q && key != 0xffffffffu) {
                const unsigned long long full = ((unsigned long long)(key >> 12) << 32) | (unsigned)(o0 + (int)(key & 0xfff));
                atomicMin(reinterpret_cast<unsigned long long *>(a.best) + (int64_t)qi * a.n_clips + clip, full);
            }
        }
    }
}

// ---- one query (or a few, one launch each): the rows of the MFMA tiles are shifts of the query ----
// acc_tu(m, n) = sum_j <Q[j - 32 t - m], R[t0 + S (n + 32 u) + j]> = the dot product at offset t0 + 1024 MT u + S n +
// 32 t + m, S = 32 MT: MT tiles of 32 shifts stacked in M, NT tiles of 32 columns (S offsets apart) side by side in N,
// cover 1024 MT NT consecutive offsets of one clip and every row does useful work (a tile of hamming_mfma_kernel would
// carry 31 rows of padding).  The MT NT matrix instructions of a step share MT query operands and NT column operands.
// Measured at 125 000 clips of 2320, query 304: (MT, NT) = (1, 2) 2.65 ms, (2, 1) 3.47 ms, (1, 1) 4.3 ms -- the query
// operands' register moves (vector ALU, which the SIMD shares with the issue of the matrix instructions) cost more than
// the column operands' LDS reads.
// Workgroup = one such item; its waves split the k + S - 1 steps and add their accumulators through LDS.  The clip window
// is stored transposed (slot i at row i mod S) so that the columns' operands, S slots apart, are read from consecutive
// addresses.
// The query operands never touch LDS: row m of step j + 1 is row m - 1 of step j, so a wave keeps each in four
// registers and moves it down one lane per step (DPP wave_shr:1); the step's new hashprint enters at row 0 of both
// halves of K from a feeder register that holds the next 32 steps' hashprints, one per lane, and rotates with it.
struct SearchShiftArgs {
    const uint64_t *db;
    const int64_t *db_off;
    int n_clips;
    const v4i *qexp;   // [2][k + 128 MT]: slot i = half h of Q[i - 32 MT] expanded (expand32p), zeros outside the query
    int k;             // the query's length
    uint64_t *best;    // [n_clips], initialised to ~0
    int chunks;        // workgroups (of 1024 MT NT offsets) per clip
    int ws;            // row stride of the transposed window (odd)
};

// 32 bits -> 32 E2M1 nibbles (0 -> +1.0 = 0x2, 1 -> -1.0 = 0xA) by byte permutes: a selector byte holds one 2-bit field
// of the word and picks the byte with its two nibbles out of a four-byte table.  The nibbles come out in a permuted
// order of the bits (dword c = the fields at bits 2c, 2c + 1 of the four bytes), which the contraction over K does not
// see as long as both operands are expanded alike -- this kernel's query and window both are.
__device__ __forceinline__ v4i expand32p(uint32_t w)
{
    const uint32_t pool = 0xAAA22A22u; // field 0 -> 0x22, 1 -> 0x2A, 2 -> 0xA2, 3 -> 0xAA
    const uint32_t m = 0x03030303u;
    v4i r;
    r.x = (int)__builtin_amdgcn_perm(0u, pool, w & m);
    r.y = (int)__builtin_amdgcn_perm(0u, pool, (w >> 2) & m);
    r.z = (int)__builtin_amdgcn_perm(0u, pool, (w >> 4) & m);
    r.w = (int)__builtin_amdgcn_perm(0u, pool, (w >> 6) & m);
    return r;
}

__global__ __launch_bounds__(256) void expand_query_shift_kernel(const uint64_t *__restrict__ q, int k, int lead, int qlen,
                                                                 v4i *__restrict__ qexp)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= qlen) return;
    const int j = i - lead;
    const bool in = j >= 0 && j < k;
    const uint64_t w = in ? q[j] : 0ull;
    qexp[i] = in ? expand32p((uint32_t)w) : v4i{0, 0, 0, 0};
    qexp[qlen + i] = in ? expand32p((uint32_t)(w >> 32)) : v4i{0, 0, 0, 0};
}

// One step of a query operand: row m <- row m - 1 (DPP wave_shr:1), row 0 of either half of K <- the value at the head
// of the feeder f (lanes 0 and 32, picked by the mask; lane 0 has no lane to take from and reads zero under bound_ctrl,
// which the mask discards); then the feeder moves up one lane (wave_rol:1), bringing the next step's rows to its lanes
// 0 and 32.  The feeder is loaded once per 32 steps -- lane L of a half holds the hashprint of step L -- so the loop
// issues no global load: nothing but the window's LDS reads feeds the matrix instructions.
// (The result goes to registers of its own: the matrix instructions of the step before may still be reading a.)
__device__ __forceinline__ v4i shift_rows_feed(const v4i &a, v4i &f)
{
    const unsigned long long heads = 1ull | (1ull << 32);
    v4i r;
    asm volatile("s_nop 1\n\t"
                 "s_mov_b64 vcc, %12\n\t"
                 "v_cndmask_b32_dpp %0, %8, %4, vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "v_cndmask_b32_dpp %1, %9, %5, vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "v_cndmask_b32_dpp %2, %10, %6, vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "v_cndmask_b32_dpp %3, %11, %7, vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "v_mov_b32_dpp %4, %4 wave_rol:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_mov_b32_dpp %5, %5 wave_rol:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_mov_b32_dpp %6, %6 wave_rol:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_mov_b32_dpp %7, %7 wave_rol:1 row_mask:0xf bank_mask:0xf"
                 : "=&v"(r.x), "=&v"(r.y), "=&v"(r.z), "=&v"(r.w), "+v"(f.x), "+v"(f.y), "+v"(f.z), "+v"(f.w)
                 : "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w), "s"(heads)
                 : "vcc");
    return r;
}

// WAVES waves split the steps; KST steps of column operands are in flight behind the matrix instructions of the KST
// before them.  (registers: two workgroups of eight waves on a CU are four waves per SIMD, 128 registers each)
template <int MT, int NT, int WAVES, int KST>
__global__ __launch_bounds__(64 * WAVES, WAVES == 8 ? 4 : 3) void hamming_shift_kernel(SearchShiftArgs a)
{
    constexpr int kThr = 64 * WAVES, S = 32 * MT;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m_lane = lane & 31, h = lane >> 5;
    const int clip = blockIdx.x / a.chunks;
    const int64_t r0 = a.db_off[clip];
    const int n = (int)(a.db_off[clip + 1] - r0);
    if (n <= 0 || a.k <= 0) return;
    const int keff = a.k < n ? a.k : n; // storage.h:37-39
    const int t0 = (blockIdx.x - clip * a.chunks) * 1024 * MT * NT;
    if (t0 > n - keff) return;
    const int steps = a.k + S - 1;
    const int win = 1024 * MT * NT + steps;        // window slots used: S (n + 32 u) + j <= S (32 NT - 1) + steps - 1
    const int plane = S * a.ws;                    // v4i per half-plane of the transposed window
    v4i *wB = reinterpret_cast<v4i *>(smem_raw);   // [2][S][ws]
    // this wave's steps; its query operands of the step before its first (row m of tile t = Q[j0 - 1 - 32 t - m]) and
    // the feeders of its first segment (lane L of tile t = Q[j0 - 32 t + L]); slot of Q[j] = j + S
    const int per = (steps + WAVES - 1) / WAVES, j0 = wave * per, j1 = min(steps, j0 + per);
    const int qlen = a.k + 4 * S;
    const v4i *qh = a.qexp + h * qlen;
    v4i av[MT], feed[MT], feed_next[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        av[t] = qh[min(j0, steps) - 1 - 32 * t - m_lane + S];
        feed_next[t] = qh[min(j0 - 32 * t + m_lane + S, qlen - 1)];
    }
    // every load of a staging round (up to 8 window slots per thread) is issued before the first value is expanded,
    // so their latencies overlap instead of adding up
    constexpr int kLd = 8;
    for (int i0 = tid; i0 < win; i0 += kLd * kThr) {
        uint64_t w[kLd];
#pragma unroll
        for (int e = 0; e < kLd; ++e) {
            const int gi = t0 + i0 + e * kThr;
            w[e] = (i0 + e * kThr < win && gi < n) ? a.db[r0 + gi] : 0ull;
        }
        if (t0 + win <= n) { // the whole window lies inside the clip (all but the last item of a clip)
#pragma unroll
            for (int e = 0; e < kLd; ++e) {
                const int i = i0 + e * kThr;
                if (i < win) {
                    const int slot = (i % S) * a.ws + i / S;
                    wB[slot] = expand32p((uint32_t)w[e]);
                    wB[plane + slot] = expand32p((uint32_t)(w[e] >> 32));
                }
            }
        } else {
#pragma unroll
            for (int e = 0; e < kLd; ++e) {
                const int i = i0 + e * kThr;
                if (i < win) {
                    const bool in = t0 + i < n; // past the end of the clip: fp4 zeros, not the expansion of 0 = all +1
                    const int slot = (i % S) * a.ws + i / S;
                    wB[slot] = in ? expand32p((uint32_t)w[e]) : v4i{0, 0, 0, 0};
                    wB[plane + slot] = in ? expand32p((uint32_t)(w[e] >> 32)) : v4i{0, 0, 0, 0};
                }
            }
        }
    }
    __syncthreads();
    f32x16 acc[MT][NT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int u = 0; u < NT; ++u) acc[t][u] = f32x16{0};
    const int one = 0x7f7f7f7f;
    const v4i *bp = wB + h * plane + m_lane;       // lane index = column n here
    v4i cb[NT][KST], nb[NT][KST];
    auto fetch = [&](int j, int jend, v4i (&fb)[NT][KST]) {
#pragma unroll
        for (int e = 0; e < KST; ++e) {
            const int jj = j + e < jend ? j + e : jend - 1; // the tail re-reads the last step (not used)
#pragma unroll
            for (int u = 0; u < NT; ++u) fb[u][e] = bp[(jj % S) * a.ws + jj / S + 32 * u];
        }
    };
    auto mult = [&](int j, int jend, const v4i (&fb)[NT][KST]) {
#pragma unroll
        for (int e = 0; e < KST; ++e) {
            if (j + e < jend) {
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    av[t] = shift_rows_feed(av[t], feed[t]);
                    const v8i a8 = {av[t].x, av[t].y, av[t].z, av[t].w, 0, 0, 0, 0};
#pragma unroll
                    for (int u = 0; u < NT; ++u) {
                        const v8i bv = {fb[u][e].x, fb[u][e].y, fb[u][e].z, fb[u][e].w, 0, 0, 0, 0};
                        acc[t][u] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, bv, acc[t][u], 4, 4, 0, one, 0, one);
                    }
                }
            }
        }
    };
    // segments of at most 32 steps, one feeder per tile each (fetched one segment ahead)
    for (int ja = j0; ja < j1; ja += 32) {
        const int jb = min(j1, ja + 32);
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            feed[t] = feed_next[t];
            feed_next[t] = qh[min(ja + 32 - 32 * t + m_lane + S, qlen - 1)];
        }
        fetch(ja, jb, cb);
        for (int j = ja; j < jb; j += 2 * KST) { // two register sets in turn, no copies
            if (j + KST < jb) fetch(j + KST, jb, nb);
            __builtin_amdgcn_sched_barrier(0);
            mult(j, jb, cb);
            __builtin_amdgcn_sched_barrier(0);
            if (j + 2 * KST < jb) fetch(j + 2 * KST, jb, cb);
            __builtin_amdgcn_sched_barrier(0);
            if (j + KST < jb) mult(j + KST, jb, nb);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __syncthreads(); // the window is no longer read: its memory takes the partial sums
    constexpr int TILES = MT * NT;
    static_assert(16 * TILES / WAVES == 4, "a thread sums one group of four accumulator registers");
    f32x4 *red = reinterpret_cast<f32x4 *>(smem_raw); // [WAVES][MT][NT][4 groups of 4 regs][64 lanes]
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int u = 0; u < NT; ++u)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                red[((wave * TILES + t * NT + u) * 4 + g) * 64 + lane] =
                    f32x4{acc[t][u][4 * g], acc[t][u][4 * g + 1], acc[t][u][4 * g + 2], acc[t][u][4 * g + 3]};
    __syncthreads();
    // wave w sums group (w mod 4) of tile (w / 4) over the waves: four sums per thread
    unsigned key = 0xffffffffu;
    const int tu = wave >> 2, g4 = wave & 3, t = tu / NT, u = tu % NT;
    f32x4 dots = {0.0f, 0.0f, 0.0f, 0.0f}; // (integers below 2^24: any order of the partial sums gives the same value)
#pragma unroll
    for (int w = 0; w < WAVES; ++w) dots += red[((w * TILES + tu) * 4 + g4) * 64 + lane];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int reg = 4 * g4 + e;
        const float dot = dots[e];
        const int m = (reg & 3) + 8 * (reg >> 2) + 4 * h; // row of the tile
        const int lo = S * (m_lane + 32 * u) + 32 * t + m; // column m_lane of tile u: base offset S (n + 32 u)
        const int dist = (64 * keff - (int)dot) >> 1;
        const unsigned cand = ((unsigned)dist << 12) | (unsigned)lo;
        if (t0 + lo <= n - keff && cand < key) key = cand;
    }
#pragma unroll
    for (int s2 = 32; s2 >= 1; s2 >>= 1) {
        const unsigned o = (unsigned)__shfl_xor((int)key, s2);
        key = o < key ? o : key;
    }
    // one atomic per wave: no further pass through LDS
    if (lane == 0 && key != 0xffffffffu) {
        const unsigned long long full = ((unsigned long long)(key >> 12) << 32) | (unsigned)(t0 + (int)(key & 0xfff));
        atomicMin(reinterpret_cast<unsigned long long *>(a.best) + clip, full);
    }
}

static int shift_ws(int k, int mt, int nt)
{
    const int s = 32 * mt;
    const int w = (1024 * mt * nt + k + s - 1 + s - 1) / s; // rows of s slots that hold the window
    return w | 1;
}

static bool shift_two_tiles()
{
    static const bool on = std::getenv("HPFW_SHIFT_ONE_TILE") == nullptr;
    return on;
}

static size_t shift_lds_bytes(int k, int mt, int nt, int waves)
{
    const size_t win = (size_t)2 * 32 * mt * shift_ws(k, mt, nt) * 16;
    return std::max(win, (size_t)waves * mt * nt * 4096); // (the partial sums overlay the window)
}

size_t hamming_shift_lds_bytes(int k) { return shift_lds_bytes(k, 1, 1, 4); }
size_t hamming_shift_image_bytes(int k) { return (size_t)2 * (k + 4 * 32) * 16; }

// one query of k hashprints at d_q against the whole index: best[clip] (preset to ~0) gets (dist << 32) | offset.
// d_qexp: hamming_shift_image_bytes(k) of scratch for the query's expanded image.
void launch_hamming_shift(const uint64_t *d_db, const int64_t *d_db_off, int n_clips, int n_off_max, const uint64_t *d_q,
                          int k, void *d_qexp, uint64_t *d_best, hipStream_t s)
{
    static PerDeviceOnce attr_set;
    if (attr_set.need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(hamming_shift_kernel<1, 1, 4, 4>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(hamming_shift_kernel<1, 2, 8, 2>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set.mark();
    }
    SearchShiftArgs a;
    a.db = d_db;
    a.db_off = d_db_off;
    a.n_clips = n_clips;
    a.qexp = reinterpret_cast<const v4i *>(d_qexp);
    a.k = k;
    a.best = d_best;
    const int tiles1 = (n_off_max + 1023) / 1024;
    // two tiles of columns per workgroup share every step's query operand; one tile when the clips have no more than
    // 1024 offsets, or when the doubled window does not fit the LDS
    const bool two = shift_two_tiles() && tiles1 > 1 && shift_lds_bytes(k, 1, 2, 8) <= 160 * 1024;
    const int lead = 32, qlen = k + 4 * lead;
    hipLaunchKernelGGL(expand_query_shift_kernel, dim3((qlen + 255) / 256), dim3(256), 0, s, d_q, k, lead, qlen,
                       reinterpret_cast<v4i *>(d_qexp));
    const int nt = two ? 2 : 1;
    a.chunks = (tiles1 + nt - 1) / nt;
    a.ws = shift_ws(k, 1, nt);
    const dim3 grid((unsigned)a.chunks * (unsigned)n_clips);
    if (two)
        hipLaunchKernelGGL((hamming_shift_kernel<1, 2, 8, 2>), grid, dim3(512), shift_lds_bytes(k, 1, 2, 8), s, a);
    else
        hipLaunchKernelGGL((hamming_shift_kernel<1, 1, 4, 4>), grid, dim3(256), shift_lds_bytes(k, 1, 1, 4), s, a);
}

size_t hamming_mfma_lds_bytes(int kt)
{
    return (size_t)2 * (kSmWgOffs + kt) * 16 + (size_t)2 * kSmChunk * 64 * 16 + 32 * 4 + (kSmThreads / 64) * 32 * 4;
}

int hamming_mfma_kt_pad(int k_max) { return (k_max + kSmChunk - 1) / kSmChunk * kSmChunk + kSmChunk; }

void launch_expand_queries(const uint64_t *d_q, const int64_t *d_q_off, int n_q, int kt_pad, void *d_qa, hipStream_t s)
{
    const int n_groups = (n_q + 31) / 32;
    dim3 grid((kt_pad * 64 + 255) / 256, n_groups);
    hipLaunchKernelGGL(expand_queries_kernel, grid, dim3(256), 0, s, d_q, d_q_off, n_q, kt_pad,
                       reinterpret_cast<v4i *>(d_qa));
}

static void mfma_scan_attrs()
{
    static PerDeviceOnce attr_set;
    if (attr_set.need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(hamming_mfma_kernel<false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(hamming_mfma_kernel<true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set.mark();
    }
}

void launch_expand_windows(const uint64_t *d_q, const int64_t *d_w_start, int n_win, int win, int kt_pad, void *d_qa,
                           hipStream_t s)
{
    dim3 grid((kt_pad * 64 + 255) / 256, (n_win + 31) / 32);
    hipLaunchKernelGGL(expand_windows_kernel, grid, dim3(256), 0, s, d_q, d_w_start, n_win, win, kt_pad,
                       reinterpret_cast<v4i *>(d_qa));
}

// nearest windows: rows = n_win windows of `win` hashprints (image d_qa), slots [n_win][8] preset to ~0
void launch_knn_windows(const uint64_t *d_db, const int64_t *d_db_off, int n_clips, int n_off_max, const void *d_qa,
                        int kt_pad, int n_win, int win, int nn, void *d_slots, hipStream_t s)
{
    mfma_scan_attrs();
    SearchMfmaArgs m = {};
    m.db = d_db;
    m.db_off = d_db_off;
    m.n_clips = n_clips;
    m.n_q = n_win;
    m.qa = reinterpret_cast<const v4i *>(d_qa);
    m.kt_pad = kt_pad;
    m.win = win;
    m.nn = nn;
    m.slots = reinterpret_cast<unsigned long long *>(d_slots);
    m.chunks = (n_off_max + kSmWgOffs - 1) / kSmWgOffs;
    dim3 grid((unsigned)m.chunks * (unsigned)n_clips, (n_win + 31) / 32);
    hipLaunchKernelGGL(hamming_mfma_kernel<true>, grid, dim3(kSmThreads), hamming_mfma_lds_bytes(win), s, m);
}

void launch_hamming_mfma(const SearchArgs &a, const void *d_qa, int kt_pad, const int *d_gk, int n_max, hipStream_t s)
{
    mfma_scan_attrs();
    SearchMfmaArgs m = {};
    m.db = a.db;
    m.db_off = a.db_off;
    m.n_clips = a.n_clips;
    m.q_off = a.q_off;
    m.n_q = a.n_q;
    m.qa = reinterpret_cast<const v4i *>(d_qa);
    m.kt_pad = kt_pad;
    m.gk = d_gk;
    m.best = a.best;
    const int n_groups = (a.n_q + 31) / 32;
    m.chunks = (n_max + kSmWgOffs - 1) / kSmWgOffs;
    dim3 grid((unsigned)m.chunks * (unsigned)a.n_clips, n_groups);
    hipLaunchKernelGGL(hamming_mfma_kernel<false>, grid, dim3(kSmThreads), hamming_mfma_lds_bytes(a.k_max), s, m);
}

} // namespace hpfw
