// k_forward.hip -- a1 + the forward half of a2: int16 PCM -> forward DFT bins [kmin, kmax).
//
// Replaces essentia MonoLoader (PCM16 mono 44.1 kHz case) and the length-N FFT inside essentia
// NSGConstantQ::compute, called from CQT<>::spectrogram (reference include/hpfw/spectrum/cqt.h:45-52,
// 66-71).  N = n1 * n2 (n2 = 6300, n1 = 210 for a 30 s clip):
//   pcm_pairs: coalescing pre-pass.  The clip is an [n2][n1] row-major matrix of samples; residue pair
//              (2p, 2p+1) is a 4-byte column of it.  A workgroup transposes a tile of 64 time steps
//              through LDS so that fwd_rows reads its pair stream contiguously.
//   fwd_rows : one workgroup per pair of residues (a, a+1) mod n1: the two real sequences
//              x[a + n1 t], x[a+1 + n1 t] ride one complex length-n2 FFT held in LDS (fft_rows.h), are
//              split by Hermitian symmetry, multiplied by T_N[a k2] and written as planar half spectra.
//   fwd_cols : the remaining length-n1 DFT, only for the rows k1 that hold consumed bins (about 10 % of
//              the N/2 bins feed the 121 bands): a dense [4 K1 x 2 n1] . [2 n1 x h] real contraction on
//              v_mfma_f32_32x32x2_f32.  The MFMA chains its k index in ascending order, which is the
//              specification's fma chain over residues (Re then Im part of each), bit for bit.
#include "kernels.h"

namespace hpfw {

extern __shared__ __align__(16) unsigned char smem_raw[];

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kPairsTileMax = 128; // time steps per tile; fewer when n1 is large (the tile lives in LDS)

// VEC: the tile (nt * n1 samples) is copied in 16-byte pieces -- the launcher checks that every tile
// starts 16-byte aligned and holds a multiple of 8 samples; otherwise sample by sample.
// n: samples between consecutive clips; n_valid: samples of a clip that exist (the chirp-z transform pads the
// [n2][n1] matrix with zeros beyond them: n_valid < n1 n2)
template <bool VEC>
__global__ __launch_bounds__(256) void pcm_pairs_kernel(int64_t n, int64_t n_valid, int n1, int n2, int kPairsTile,
                                                        const int16_t *__restrict__ pcm, i16x2 *__restrict__ pairs)
{
    int16_t *tile = reinterpret_cast<int16_t *>(smem_raw); // [kPairsTile][n1]
    const int tid = threadIdx.x;
    const int clip = blockIdx.y;
    const int t0 = blockIdx.x * kPairsTile;
    const int nt = min(kPairsTile, n2 - t0);
    const int np = (n1 + 1) / 2;
    const int16_t *src = pcm + (int64_t)clip * n + (int64_t)t0 * n1;
    if (VEC) {
        const uint4 *s4 = reinterpret_cast<const uint4 *>(src);
        uint4 *t4 = reinterpret_cast<uint4 *>(tile);
        for (int i = tid; i < nt * n1 / 8; i += 256) t4[i] = s4[i];
    } else {
        const int64_t left = n_valid - (int64_t)t0 * n1; // samples of this tile that exist
        for (int i = tid; i < nt * n1; i += 256) tile[i] = i < left ? src[i] : (int16_t)0;
    }
    __syncthreads();
    i16x2 *dst = pairs + (int64_t)clip * np * n2 + t0;
    const bool even = (n1 & 1) == 0; // then a pair is one aligned 32-bit word of the tile
    // the tile length is a power of two: a thread keeps its time step and walks the pairs (no division per word)
    const int lg = 31 - __builtin_clz((unsigned)kPairsTile);
    const int tt = tid & (kPairsTile - 1), pstep = 256 >> lg;
    if (pstep >= 1) {
        if (tt < nt) {
            const int16_t *row = tile + tt * n1;
            i16x2 *out = dst + tt;
            for (int p = tid >> lg; p < np; p += pstep) {
                i16x2 v;
                if (even) {
                    v = *reinterpret_cast<const i16x2 *>(row + 2 * p);
                } else {
                    v.x = row[2 * p];
                    v.y = (2 * p + 1 < n1) ? row[2 * p + 1] : (short)0;
                }
                out[(int64_t)p * n2] = v;
            }
        }
    } else { // a tile longer than the workgroup (not produced by the launcher): the general walk
        for (int i = tid; i < np * kPairsTile; i += 256) {
            const int p = i >> lg, t2 = i & (kPairsTile - 1);
            if (t2 < nt) {
                i16x2 v;
                v.x = tile[t2 * n1 + 2 * p];
                v.y = (2 * p + 1 < n1) ? tile[t2 * n1 + 2 * p + 1] : (short)0;
                dst[(int64_t)p * n2 + t2] = v;
            }
        }
    }
}

// one workgroup = one residue pair of one clip; the body (fft_rows.h) is shared with tests/emu.
// 512 threads x 128 VGPRs and 50 KB of LDS: two workgroups per CU, every pass one butterfly per thread.
constexpr int kFwdThreads = 512;

// WAVES: waves per SIMD the register allocation is held to (4: two workgroups per CU; 6: three)
template <class Groups, int WAVES>
__global__ __launch_bounds__(kFwdThreads, WAVES) void fwd_rows_kernel(RowsArgs a, const i16x2 *__restrict__ pairs,
                                                                  int64_t clip_pitch, int pair_pitch,
                                                                  float *__restrict__ yp)
{
    cf *lds = reinterpret_cast<cf *>(smem_raw);
    // clip is the fast grid index: workgroups resident at the same time then share one residue
    // pair, so its two rows of the T_N[a k2] table (50 KB of the 5.3 MB) stay in L2 instead of the
    // whole table cycling through it once per clip
    const int p = blockIdx.y;
    const int clip = blockIdx.x;
    const int a0 = 2 * p;
    float *ya = yp + ((int64_t)clip * 2 * a.n1 + 2 * a0) * a.hpad;
    float *yb = (a0 + 1 < a.n1) ? ya + 2 * (int64_t)a.hpad : nullptr;
    rows_body<Groups>(lds, a, kFwdThreads, pairs + clip * clip_pitch + (int64_t)p * pair_pitch, a0, ya, yb);
}

// ---- column DFT on the matrix cores --------------------------------------------------------
// D[row][k2] = sum_k A[row][k] B[k][k2], k = 2 a + part: B = planar Y' (row 2a = Re, 2a+1 = Im of
// residue a); A rows come in (Re, Im) pairs per wanted k1, "complex row" cr:
//   cr <  K1 : k1 = k1lo + cr           -> X[n2 k1 + k2]
//   cr >= K1 : k1' = n1 - 1 - k1        -> X[n2 k1 + (n2 - k2)] = conj(.)      (X[k] = conj X[N - k])
// One wave = one tile of 32 columns x NT row tiles of 32 (16 complex rows each); no LDS, no barriers.
// kColsStep = MFMA k-steps (residues) per register block, a template argument: the loop runs an even number
// of blocks, so the launcher picks the step that pads n1 least (n1 = 210: 14 blocks of 15, where 16 pads to 224)

// Operands come through buffer loads: lane part of the address in one VGPR that never changes, the
// residue step in an SGPR, the row-tile step in the instruction's immediate -- no vector address
// arithmetic between the MFMAs -- and a residue past n1 (the padded tail of the last block) is
// out of range of the descriptor and reads as 0.
__device__ __forceinline__ float cols_ld(__amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}

template <int NT, int kColsStep>
__global__ __launch_bounds__(256, 2) void fwd_cols_kernel(ColsArgs ca, int tile0, const float *__restrict__ yp,
                                                          cf *__restrict__ x)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ctile = blockIdx.x * 4 + wave;
    const int clip = blockIdx.y;
    if (ctile * 32 >= ca.h) return;
    const int hb = lane >> 5, j = lane & 31;
    const int64_t clip_floats = (int64_t)2 * ca.n1 * ca.hpad;
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(yp + clip * clip_floats), (short)0, (int)(clip_floats * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(ca.apack), (short)0, ca.n1 * ca.n_tiles * 256, 0x00020000); // [a][tile][lane]
    const int vb = (hb * ca.hpad + ctile * 32 + j) * 4;
    const int va = (tile0 * 64 + lane) * 4;
    const int sb = 2 * ca.hpad * 4;                          // bytes per residue in Y' (Re row, Im row)
    const int sa = ca.n_tiles * 256;                         // bytes per residue in the coefficient image
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x16{0};
    float a0[kColsStep][NT], b0[kColsStep], a1[kColsStep][NT], b1[kColsStep];
    const int nblocks = (ca.n1 + kColsStep - 1) / kColsStep;
#pragma unroll
    for (int s = 0; s < kColsStep; ++s) {
        b0[s] = cols_ld(rb, vb, s * sb);
#pragma unroll
        for (int t = 0; t < NT; ++t) a0[s][t] = cols_ld(ra, va + t * 256, s * sa);
    }
    // two register sets in turn: the loads of the next block are issued between the MFMAs of this one
#pragma unroll 1
    for (int blk = 0; blk < nblocks; blk += 2) {
        const int r1 = (blk + 1) * kColsStep, r2 = (blk + 2) * kColsStep;
#pragma unroll
        for (int s = 0; s < kColsStep; ++s) {
            b1[s] = cols_ld(rb, vb, (r1 + s) * sb);
#pragma unroll
            for (int t = 0; t < NT; ++t) a1[s][t] = cols_ld(ra, va + t * 256, (r1 + s) * sa);
#pragma unroll
            for (int t = 0; t < NT; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s][t], b0[s], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0); // keep the loads up here: the scheduler would sink them to save registers
        }
#pragma unroll
        for (int s = 0; s < kColsStep; ++s) {
            b0[s] = cols_ld(rb, vb, (r2 + s) * sb);
#pragma unroll
            for (int t = 0; t < NT; ++t) a0[s][t] = cols_ld(ra, va + t * 256, (r2 + s) * sa);
#pragma unroll
            for (int t = 0; t < NT; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s][t], b1[s], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // D layout: column = lane & 31; registers (2q, 2q+1) of a tile are rows 2p, 2p+1 with
    // p = (q & 1) + 4 (q >> 1) + 2 (lane >> 5): the Re / Im rows of complex row tile * 16 + p
    const int k2 = ctile * 32 + j;
    cf *xo = x + (int64_t)clip * (ca.kmax - ca.kmin);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int cr = (tile0 + t) * 16 + (q & 1) + 4 * (q >> 1) + 2 * hb;
            const float re = acc[t][2 * q], im = acc[t][2 * q + 1];
            if (cr < ca.k1n) {
                const int64_t k = (int64_t)ca.n2 * (ca.k1lo + cr) + k2;
                if (k2 < ca.h && k >= ca.kmin && k < ca.kmax) xo[k - ca.kmin] = {re, im};
            } else if (cr < 2 * ca.k1n) {
                const int64_t k = (int64_t)ca.n2 * (ca.k1lo + cr - ca.k1n) + (ca.n2 - k2);
                if (k2 >= 1 && k2 <= ca.n2 - ca.h && k >= ca.kmin && k < ca.kmax) xo[k - ca.kmin] = {re, -im};
            }
        }
    }
}

void launch_pcm_pairs(int64_t n, int n1, int n2, const int16_t *d_pcm, int n_clips, i16x2 *d_pairs, hipStream_t s)
{
    int kPairsTile = kPairsTileMax;
    while (kPairsTile > 1 && (size_t)kPairsTile * n1 * sizeof(int16_t) > 60 * 1024) kPairsTile /= 2;
    dim3 grid((n2 + kPairsTile - 1) / kPairsTile, n_clips);
    const int tail = n2 % kPairsTile;
    const bool vec = (reinterpret_cast<uintptr_t>(d_pcm) % 16 == 0) && (n * 2 % 16 == 0) && (kPairsTile * n1 % 8 == 0) &&
                     (tail * n1 % 8 == 0);
    const size_t lds = ((size_t)kPairsTile * n1 * sizeof(int16_t) + 15) / 16 * 16;
    if (vec)
        hipLaunchKernelGGL(pcm_pairs_kernel<true>, grid, dim3(256), lds, s, n, n, n1, n2, kPairsTile, d_pcm, d_pairs);
    else
        hipLaunchKernelGGL(pcm_pairs_kernel<false>, grid, dim3(256), lds, s, n, n, n1, n2, kPairsTile, d_pcm, d_pairs);
}

size_t fwd_rows_lds_bytes(const RowsArgs &a) { return (size_t)a.n2 * sizeof(cf); }

template <class Groups, int WAVES>
static void launch_rows_t(const RowsArgs &a, const i16x2 *d_pairs, int n_clips, float *d_yp, hipStream_t s)
{
    static PerDeviceOnce attr_set;
    if (attr_set.need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(fwd_rows_kernel<Groups, WAVES>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set.mark();
    }
    dim3 grid(n_clips, (a.n1 + 1) / 2);
    // pre-passed stream: [clip][pair][n2]; in place: pair p of time step t at pcm word t n1/2 + p
    const int64_t clip_pitch = a.pair_stride == 1 ? (int64_t)((a.n1 + 1) / 2) * a.n2 : (int64_t)a.n1 * a.n2 / 2;
    const int pair_pitch = a.pair_stride == 1 ? a.n2 : 1;
    hipLaunchKernelGGL((fwd_rows_kernel<Groups, WAVES>), grid, dim3(kFwdThreads), fwd_rows_lds_bytes(a), s, a, d_pairs,
                       clip_pitch, pair_pitch, d_yp);
}

// d_src: the pcm_pairs output when a.pair_stride == 1, otherwise the PCM itself (n1 even: every
// residue pair is an aligned 4-byte word of the [n2][n1] sample matrix)
void launch_fwd_rows(const RowsArgs &a, const i16x2 *d_src, int n_clips, float *d_yp, hipStream_t s)
{
    // the compile-time sequence runs its last two groups one butterfly per thread
    // 4 waves per SIMD (two workgroups per CU): held to 6 (80 VGPRs, three workgroups) the static kernel measured
    // slower, 4.42 against 4.17 ms for pairs + rows per 1000 clips
    if (Groups6300::matches_plan(a) && Groups6300::min_threads(a.n2) <= kFwdThreads)
        launch_rows_t<Groups6300, 4>(a, d_src, n_clips, d_yp, s);
    else
        launch_rows_t<RuntimeGroups, 4>(a, d_src, n_clips, d_yp, s);
}

template <int STEP>
static void launch_cols_step(const ColsArgs &ca, const float *d_yp, int n_clips, cf *d_x, hipStream_t s)
{
    dim3 grid(((ca.h + 31) / 32 + 3) / 4, n_clips);
    for (int t0 = 0; t0 < ca.n_tiles; t0 += 3) {
        const int nt = ca.n_tiles - t0 < 3 ? ca.n_tiles - t0 : 3;
        if (nt == 3)
            hipLaunchKernelGGL((fwd_cols_kernel<3, STEP>), grid, dim3(256), 0, s, ca, t0, d_yp, d_x);
        else if (nt == 2)
            hipLaunchKernelGGL((fwd_cols_kernel<2, STEP>), grid, dim3(256), 0, s, ca, t0, d_yp, d_x);
        else
            hipLaunchKernelGGL((fwd_cols_kernel<1, STEP>), grid, dim3(256), 0, s, ca, t0, d_yp, d_x);
    }
}

void launch_fwd_cols(const ColsArgs &ca, const float *d_yp, int n_clips, cf *d_x, hipStream_t s)
{
    auto padded = [&](int step) { return ((ca.n1 + step - 1) / step + 1) / 2 * 2 * step; }; // residues the loop walks
    int best = 16;
    for (int step : {15, 14})
        if (padded(step) < padded(best)) best = step;
    if (best == 16)
        launch_cols_step<16>(ca, d_yp, n_clips, d_x, s);
    else if (best == 15)
        launch_cols_step<15>(ca, d_yp, n_clips, d_x, s);
    else
        launch_cols_step<14>(ca, d_yp, n_clips, d_x, s);
}

// host: coefficient image of the column DFT for the MFMA A operand, [a][tile][lane]:
// lane l supplies A[row = 32 tile + (l & 31)][k = 2 a + (l >> 5)]; row 2 cr = Re row, 2 cr + 1 = Im row.
void pack_cols_coefficients(int n1, int k1lo, int k1n, const float *tw_n1_ri, int n_tiles, float *apack)
{
    for (int a = 0; a < n1; ++a)
        for (int t = 0; t < n_tiles; ++t)
            for (int l = 0; l < 64; ++l) {
                const int row = 32 * t + (l & 31), cr = row >> 1, part = l >> 5;
                float v = 0.0f;
                if (cr < 2 * k1n) {
                    const int k1 = cr < k1n ? k1lo + cr : n1 - 1 - (k1lo + cr - k1n);
                    const int64_t idx = ((int64_t)a * k1) % n1;
                    const float dr = tw_n1_ri[2 * idx], di = tw_n1_ri[2 * idx + 1];
                    if ((row & 1) == 0) v = part == 0 ? dr : -di; // Re: fma(dr, yr), fma(-di, yi)
                    else v = part == 0 ? di : dr;                  // Im: fma(di, yr), fma(dr, yi)
                }
                apack[((size_t)a * n_tiles + t) * 64 + l] = v;
            }
}

} // namespace hpfw
