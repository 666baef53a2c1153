// k_bluestein.hip -- the forward DFT bins [kmin, kmax) of a clip whose length N has a prime factor above 7
// (DESIGN.md S15): essentia hands the file's exact sample count to FFTW (reference
// include/hpfw/spectrum/cqt.h:54-55, inputSize = audioBuffer.size()), so N is whatever the file holds.
//
//   X[k] = w[k] sum_n (x[n] w[n]) conj(w[k - n]),  w[n] = e^{-i pi n^2 / N}         (Bluestein / chirp-z)
//
// as a cyclic convolution of length L = n1 * n2 >= N + (kmax - kmin) - 1 with n2 = 6300 (the row transform
// of the 7-smooth path, fft_rows.h, pass order [7,3,5,3,5,4]) and any n1 (the transform across residues is a
// dense DFT on the matrix cores and does not care about n1's factors):
//   pcm_pairs    (k_forward.hip) PCM -> residue streams, zeros beyond N
//   bz_rows<0>   per residue r: a[r + n1 t] = x w, FFT_n2 in LDS, times T_L[r k2]           -> Y'  planar
//   bz_cols2<0>  A[n2 k1 + k2] = sum_r T_n1[r k1] Y'[r][k2] for all n1 = 16 a rows, in two stages on f32 MFMA (length a
//                over r2, length 16 over r1, r = 16 r2 + r1), C = conj(A Bhat)                 -> C planar [k1][k2]
//   bz_transpose the flat index j = n2 k1 + k2 = r + n1 t regrouped by residue                  -> C' planar [r][t]
//   bz_rows<1>   per residue r: FFT_n2 over t in LDS, times T_L[r k2]                          -> Y'' planar
//   bz_cols<1>   F[n2 k1 + k2] = sum_r T_n1[r k1] Y''[r][k2] for the rows k1 that hold consumed bins only (about a
//                tenth of them, as in fwd_cols); X[k] = conj(F[k]) w[k] / L
// Only the first transform's column stage needs all n1 output rows: a dense n1 x n1 contraction there took more time
// than every other kernel of the path together, hence its two stages.  Arithmetic order = the oracle's, bit for
// bit: MFMA chains its k index (Re then Im part of each residue) in ascending order (S6).
#include "kernels.h"

namespace hpfw {

extern __shared__ __align__(16) unsigned char smem_raw[];

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kBzThreads = 512;

// MODE 0: first transform's rows (input x w from the pair streams); MODE 1: second transform's rows (input C', planar)
template <int MODE>
__global__ __launch_bounds__(kBzThreads, 4) void bz_rows_kernel(RowsArgs a, BzArgs bz, const i16x2 *__restrict__ pairs,
                                                                const float *__restrict__ in, float *__restrict__ out)
{
    cf *lds = reinterpret_cast<cf *>(smem_raw);
    const int row = blockIdx.y; // the residue r of the flat index r + n1 t (of the samples, or of C)
    const int clip = blockIdx.x;
    const int n2 = a.n2, nthreads = kBzThreads, tid = threadIdx.x;
    const int64_t plane = (int64_t)2 * bz.n1 * bz.n2pad; // floats per clip of a planar buffer
    if (MODE == 0) {
        // the pair word as a plain 32-bit integer (an array of 16-bit vectors ends up in scratch memory)
        const int *__restrict__ src = reinterpret_cast<const int *>(pairs + ((int64_t)clip * ((bz.n1 + 1) / 2) + (row >> 1)) * n2);
        const cf *__restrict__ w = bz.w + (int64_t)row * n2;
        const int sh = (row & 1) ? 0 : 16; // odd residue: the high half (arithmetic shift); even: the low half moved up first
        constexpr int kLd = 13;
        for (int t0 = tid; t0 < n2; t0 += kLd * nthreads) {
            int p[kLd];
            cf ww[kLd];
#pragma unroll
            for (int e = 0; e < kLd; ++e) {
                const int t = t0 + e * nthreads;
                p[e] = src[t < n2 ? t : 0];
                ww[e] = w[t < n2 ? t : 0];
            }
#pragma unroll
            for (int e = 0; e < kLd; ++e) {
                const int t = t0 + e * nthreads;
                if (t < n2) {
                    const float xs = (float)((p[e] << sh) >> 16) / 32768.0f;
                    lds[t] = {xs * ww[e].r, xs * ww[e].i};
                }
            }
        }
    } else {
        const float *__restrict__ re = in + clip * plane + (int64_t)2 * row * bz.n2pad;
        const float *__restrict__ im = re + bz.n2pad;
        for (int t = tid; t < n2; t += nthreads) lds[t] = {re[t], im[t]};
    }
    __syncthreads();
    Groups6300::run(lds, a, nthreads); // outputs in natural order
    // both transforms: times T_L[row k2], planar rows (2 row, 2 row + 1) of the output
    const cf *__restrict__ tl = bz.tl + (int64_t)row * n2;
    float *__restrict__ ore = out + clip * plane + (int64_t)2 * row * bz.n2pad;
    float *__restrict__ oim = ore + bz.n2pad;
    for (int k2 = tid; k2 < n2; k2 += nthreads) {
        const cf o = c_mul(lds[k2], tl[k2]);
        ore[k2] = o.r;
        oim[k2] = o.i;
    }
}

// C [k1][k2] (planar, n2pad columns) -> C' [r][t] (the same planar shape) with n2 k1 + k2 = r + n1 t: a workgroup takes
// the flat range of T time steps t (T n1 consecutive flat elements: coalesced reads), one plane at a time through
// LDS, and writes for every residue its T values side by side.
__global__ __launch_bounds__(256) void bz_transpose_kernel(BzArgs bz, int lg_t, const float *__restrict__ in, float *__restrict__ out)
{
    float *tile = reinterpret_cast<float *>(smem_raw); // [tsteps][n1]
    const int tsteps = 1 << lg_t;
    const int clip = blockIdx.y, t0 = blockIdx.x * tsteps;
    const int nt = min(tsteps, bz.n2 - t0);
    const int64_t plane = (int64_t)2 * bz.n1 * bz.n2pad;
    const int64_t j0 = (int64_t)bz.n1 * t0;
    const int count = bz.n1 * nt;
    // this thread's first flat element as (k1, k2); every further one is 256 on (no division per element)
    int k1 = (int)((j0 + threadIdx.x) / bz.n2), k2 = (int)((j0 + threadIdx.x) - (int64_t)k1 * bz.n2);
    const int tt = threadIdx.x & (tsteps - 1), rstep = 256 >> lg_t;
    for (int part = 0; part < 2; ++part) { // Re plane, then Im plane
        const float *src = in + clip * plane + (int64_t)part * bz.n2pad;
        int a1 = k1, a2 = k2;
        for (int i = threadIdx.x; i < count; i += 256) {
            tile[i] = src[(int64_t)2 * a1 * bz.n2pad + a2];
            a2 += 256;
            while (a2 >= bz.n2) {
                a2 -= bz.n2;
                ++a1;
            }
        }
        __syncthreads();
        float *dst = out + clip * plane + (int64_t)part * bz.n2pad + t0 + tt;
        // a thread keeps its time step and walks the residues: runs of tsteps floats per residue and wave pass
        if (tt < nt)
            for (int r = threadIdx.x >> lg_t; r < bz.n1; r += rstep) dst[(int64_t)2 * r * bz.n2pad] = tile[tt * bz.n1 + r];
        __syncthreads();
    }
}

__device__ __forceinline__ float bz_ld(__amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}

// The second transform's column stage.  D[row][col] = sum_k A[row][k] B[k][col], k = 2 r + part: B = a planar buffer
// (row 2r = Re, 2r + 1 = Im of residue r, n2pad columns), A = the DFT coefficients of the rows k1lo .. k1lo + k1n - 1
// that hold consumed bins (Re / Im row pairs), packed by bz_pack_kernel.  One wave = 32 columns x 3 row tiles of 32
// (16 complex rows each); blockIdx.z = the group of 3 row tiles.  Same register-blocked operand stream as
// fwd_cols_kernel (k_forward.hip).  x[k - kmin] = conj(D) w[k] / L for k = n2 k1 + k2 in [kmin, kmax).
template <int MODE, int kStep>
__global__ __launch_bounds__(256, 2) void bz_cols_kernel(BzArgs bz, const float *__restrict__ in, float *__restrict__ out,
                                                         cf *__restrict__ x)
{
    constexpr int NT = 3;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ctile = blockIdx.x * 4 + wave;
    const int clip = blockIdx.y;
    const int tile0 = blockIdx.z * NT;
    if (ctile * 32 >= bz.n2) return;
    const int hb = lane >> 5, j = lane & 31;
    const int64_t plane = (int64_t)2 * bz.n1 * bz.n2pad;
    const __amdgpu_buffer_rsrc_t rb =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in + clip * plane), (short)0, (int)(plane * 4), 0x00020000);
    static_assert(MODE == 1, "the full column stage is bz_cols2_kernel");
    const int n_tiles = bz.n_tiles2;
    const __amdgpu_buffer_rsrc_t ra =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(bz.apack2), (short)0, bz.n1 * n_tiles * 256, 0x00020000); // [r][tile][lane]
    const int vb = (hb * bz.n2pad + ctile * 32 + j) * 4;
    const int va = (tile0 * 64 + lane) * 4;
    const int sb = 2 * bz.n2pad * 4; // bytes per residue in the planar buffer
    const int sa = n_tiles * 256;    // bytes per residue in the coefficient image
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x16{0};
    float a0[kStep][NT], b0[kStep], a1[kStep][NT], b1[kStep];
    const int nblocks = (bz.n1 + kStep - 1) / kStep;
#pragma unroll
    for (int s = 0; s < kStep; ++s) {
        b0[s] = bz_ld(rb, vb, s * sb);
#pragma unroll
        for (int t = 0; t < NT; ++t) a0[s][t] = bz_ld(ra, va + t * 256, s * sa);
    }
#pragma unroll 1
    for (int blk = 0; blk < nblocks; blk += 2) {
        const int r1 = (blk + 1) * kStep, r2 = (blk + 2) * kStep;
#pragma unroll
        for (int s = 0; s < kStep; ++s) {
            b1[s] = bz_ld(rb, vb, (r1 + s) * sb);
#pragma unroll
            for (int t = 0; t < NT; ++t) a1[s][t] = bz_ld(ra, va + t * 256, (r1 + s) * sa);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s][t], b0[s], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int s = 0; s < kStep; ++s) {
            b0[s] = bz_ld(rb, vb, (r2 + s) * sb);
#pragma unroll
            for (int t = 0; t < NT; ++t) a0[s][t] = bz_ld(ra, va + t * 256, (r2 + s) * sa);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s][t], b1[s], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // D layout: column = lane & 31; registers (2q, 2q+1) of a tile = Re / Im of complex row
    // tile * 16 + (q & 1) + 4 (q >> 1) + 2 (lane >> 5)
    const int k2 = ctile * 32 + j;
    if (k2 >= bz.n2) return;
    {
        cf *__restrict__ xo = x + (int64_t)clip * (bz.kmax - bz.kmin);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int cr = (tile0 + t) * 16 + (q & 1) + 4 * (q >> 1) + 2 * hb;
                const int64_t k = (int64_t)bz.n2 * (bz.k1lo + cr) + k2;
                if (cr < bz.k1n && k >= bz.kmin && k < bz.kmax)
                    xo[k - bz.kmin] = c_mul(cf{acc[t][2 * q], -acc[t][2 * q + 1]}, bz.wk[k - bz.kmin]);
            }
        }
    }
}

// stage 2 of bz_cols2_kernel for the outputs k_a = 16 tile + I, 16 tile + I + 1 (see there); c3 = the tile's slice of
// the stage-2 coefficient image in LDS [16 k_a][16 r1][64 lanes]
template <int MODE>
struct Cols2Out {
    const BzArgs &bz;
    float *__restrict__ o;
    cf *__restrict__ x;
    int k2, hb, lane, tile;
    const float *c3;

    // Bhat of pair I's 2 x 8 outputs (issued one pair ahead: in flight while the previous pair's chains run)
    template <int I>
    __device__ __forceinline__ void fetch(cf (&bh)[2][8]) const
    {
        if (MODE != 0 || I >= 16) return;
        const int ka = tile * 16 + I;
        const bool live = k2 < bz.n2;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int k1 = ka + u + bz.a * ((q & 1) + 4 * (q >> 1) + 2 * hb);
                bh[u][q] = (live && ka + u < bz.a) ? bz.bhat[(int64_t)k1 * bz.n2 + k2] : cf{0.0f, 0.0f};
            }
    }

    template <int I>
    __device__ __forceinline__ void pair(const f32x16 (&z)[16], const cf (&bh)[2][8]) const
    {
        const int ka = tile * 16 + I;
        if (ka >= bz.a) return; // uniform
        const bool two = ka + 1 < bz.a;
        f32x16 d0 = f32x16{0}, d1 = f32x16{0};
#pragma unroll
        for (int r1 = 0; r1 < 16; ++r1) {
            d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(c3[(I * 16 + r1) * 64 + lane], z[r1][I], d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(c3[((I + 1) * 16 + r1) * 64 + lane], z[r1][I + 1], d1, 0, 0, 0);
        }
        // D layout: column = lane & 31; registers (2q, 2q+1) = Re / Im of k_b = (q & 1) + 4 (q >> 1) + 2 (lane >> 5)
        if (k2 >= bz.n2) return;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (u == 1 && !two) continue;
            const f32x16 &d = u ? d1 : d0;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int k1 = ka + u + bz.a * ((q & 1) + 4 * (q >> 1) + 2 * hb);
                if (MODE == 0) {
                    const cf v = c_mul(cf{d[2 * q], d[2 * q + 1]}, bh[u][q]);
                    o[(int64_t)2 * k1 * bz.n2pad + k2] = v.r;
                    o[(int64_t)(2 * k1 + 1) * bz.n2pad + k2] = -v.i;
                } else {
                    x[(int64_t)k1 * bz.n2 + k2] = cf{d[2 * q], d[2 * q + 1]};
                }
            }
        }
    }
};

// The first transform's column stage: all n1 = 16 a output rows, A[k1] = sum_r T_n1[r k1] Y'[r], per column k2.
// With r = 16 r2 + r1 (r1 < 16, r2 < a) and k1 = k_a + a k_b (k_a < a, k_b < 16):
//   stage 1   Z[r1][k_a] = sum_{r2} T_a[r2 k_a] Y'[16 r2 + r1]              (16 transforms of length a)
//   stage 2   A[k_a + a k_b] = sum_{r1} T_n1[r1 (k_a + a k_b)] Z[r1][k_a]   (a transforms of length 16, twiddles folded in)
// -- n1 (a + 16) complex products per column where the dense contraction takes n1^2, both stages as fma chains on
// v_mfma_f32_32x32x2_f32 in ascending order of the summed index (Re, then Im part of each term), nothing between
// them: stage 1's rows are laid out so that accumulator register i of the tile of r1 holds Re Z[r1][k_a = 16 T + i] in
// lanes 0-31 and Im in lanes 32-63, which is exactly the B operand of stage 2's k-step r1 (k = 2 r1, 2 r1 + 1).
// One wave = 32 columns x one tile T of 16 outputs k_a; the 16 stage-1 tiles (r1) stay in registers (256 of them:
// one wave per SIMD).  MODE 0: out = conj(A Bhat[k1][k2]) planar [k1][k2]; MODE 2 (table generation): x[n2 k1 + k2] = A.
template <int MODE>
__global__ __launch_bounds__(256, 1) void bz_cols2_kernel(BzArgs bz, const float *__restrict__ in, float *__restrict__ out,
                                                          cf *__restrict__ x)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // column groups along blockIdx.x (measured against clips along x, which would share Bhat in L2: 3.25 vs 3.41 ms)
    const int ctile = blockIdx.x * 4 + wave;
    const int clip = blockIdx.y;
    const int tile = blockIdx.z;
    // the tile's slice of the stage-2 image (zeros past k_a = a) into LDS while stage 1 runs; all four waves share it
    float *c3 = reinterpret_cast<float *>(smem_raw);
    {
        // 16 x 16 bytes per thread, all in flight before the first is stored
        const float4 *src = reinterpret_cast<const float4 *>(bz.apack3) + (int64_t)tile * 16 * 256;
        const int limit = (bz.a - tile * 16) * 256; // float4 pieces that exist (a k_a = 1024 floats)
        float4 v[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int i = threadIdx.x + 256 * e;
            v[e] = i < limit ? src[i] : float4{0.0f, 0.0f, 0.0f, 0.0f};
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) reinterpret_cast<float4 *>(c3)[threadIdx.x + 256 * e] = v[e];
    }
    const bool active = ctile * 32 < bz.n2; // (every wave reaches the barrier below)
    const int hb = lane >> 5, j = lane & 31;
    const int64_t plane = (int64_t)2 * bz.n1 * bz.n2pad;
    const __amdgpu_buffer_rsrc_t rb =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in + clip * plane), (short)0, (int)(plane * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t ra1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(bz.apack1), (short)0,
                                                                        bz.a * bz.n_tiles1 * 256, 0x00020000); // [r2][tile][lane]
    const int vb = (hb * bz.n2pad + ctile * 32 + j) * 4;
    const int va = (tile * 64 + lane) * 4;
    const int sb = 2 * bz.n2pad * 4;   // bytes per residue in the planar buffer
    const int sa = bz.n_tiles1 * 256;  // bytes per r2 in the stage-1 image
    f32x16 z[16];
#pragma unroll
    for (int r1 = 0; r1 < 16; ++r1) z[r1] = f32x16{0};
    if (active) {
        // stage 1: k-step r2 feeds the sixteen chains (r1) with one coefficient operand; the operands of the next
        // kDepth - 1 steps are in flight (one wave per SIMD: nothing else hides the latency).  Loads past the last
        // residue fall outside the buffers and return 0: the steps that pad a to a multiple of kDepth add 0 * 0 to
        // every chain, which changes no value (an accumulator that started at +0 is never -0).
        constexpr int kDepth = 4;
        float ca[kDepth], cb[kDepth][16];
#pragma unroll
        for (int d = 0; d < kDepth - 1; ++d) {
            ca[d] = bz_ld(ra1, va, d * sa);
#pragma unroll
            for (int r1 = 0; r1 < 16; ++r1) cb[d][r1] = bz_ld(rb, vb, (16 * d + r1) * sb);
        }
#pragma unroll 1
        for (int r2 = 0; r2 < bz.a; r2 += kDepth) {
#pragma unroll
            for (int d = 0; d < kDepth; ++d) {
                constexpr int kAhead = kDepth - 1;
                const int nd = (d + kAhead) % kDepth, nr = r2 + d + kAhead;
                ca[nd] = bz_ld(ra1, va, nr * sa);
#pragma unroll
                for (int r1 = 0; r1 < 16; ++r1) cb[nd][r1] = bz_ld(rb, vb, (16 * nr + r1) * sb);
#pragma unroll
                for (int r1 = 0; r1 < 16; ++r1) z[r1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ca[d], cb[d][r1], z[r1], 0, 0, 0);
            }
        }
    }
    __syncthreads();
    if (!active) return;
    // stage 2: per output k_a = 16 tile + i the chain over r1, two outputs side by side (independent accumulators);
    // i is a template constant: z must be indexed by compile-time constants to stay in registers
    const int k2 = ctile * 32 + j;
    Cols2Out<MODE> epi{bz, out + clip * plane, x, k2, hb, lane, tile, c3};
    cf bha[2][8], bhb[2][8];
    epi.template fetch<0>(bha);
    epi.template fetch<2>(bhb);
    epi.template pair<0>(z, bha);
    epi.template fetch<4>(bha);
    epi.template pair<2>(z, bhb);
    epi.template fetch<6>(bhb);
    epi.template pair<4>(z, bha);
    epi.template fetch<8>(bha);
    epi.template pair<6>(z, bhb);
    epi.template fetch<10>(bhb);
    epi.template pair<8>(z, bha);
    epi.template fetch<12>(bha);
    epi.template pair<10>(z, bhb);
    epi.template fetch<14>(bhb);
    epi.template pair<12>(z, bha);
    epi.template pair<14>(z, bhb);
}

// coefficient images of bz_cols2_kernel, from T_n1 (T_a[j] = T_n1[16 j] exactly: S2 reduces the same fraction):
// apack1 [r2][tile][lane]: lane l supplies A[row = l & 31 of the tile][k = 2 r2 + (l >> 5)]; row 8 g + u of tile T:
//   u < 4: the Re row of k_a = 16 T + 4 g + u -> (dr, -di); u >= 4: the Im row of k_a = 16 T + 4 g + u - 4 -> (di, dr),
//   d = T_a[r2 k_a]; zero rows for k_a >= a
// apack3 [k_a][r1][lane]: row l & 31 = 2 k_b (Re) / 2 k_b + 1 (Im), d = T_n1[r1 (k_a + a k_b)]
__global__ __launch_bounds__(256) void bz_pack_stages_kernel(BzArgs bz, const cf *__restrict__ tw_n1, float *__restrict__ apack1,
                                                             float *__restrict__ apack3)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t n_a1 = (int64_t)bz.a * bz.n_tiles1 * 64, n_a3 = (int64_t)bz.a * 16 * 64;
    const int l = (int)(i & 63), part = l >> 5, row = l & 31;
    if (i < n_a1) {
        const int t = (int)((i >> 6) % bz.n_tiles1), r2 = (int)((i >> 6) / bz.n_tiles1);
        const int g = row >> 3, u = row & 7, ka = 16 * t + 4 * g + (u & 3);
        float v = 0.0f;
        if (ka < bz.a) {
            const cf d = tw_n1[16 * (int)(((int64_t)r2 * ka) % bz.a)];
            if (u < 4) v = part == 0 ? d.r : -d.i;
            else v = part == 0 ? d.i : d.r;
        }
        apack1[i] = v;
    }
    if (i < n_a3) {
        const int r1 = (int)((i >> 6) & 15), ka = (int)(i >> 10), kb = row >> 1;
        const cf d = tw_n1[((int64_t)r1 * (ka + (int64_t)bz.a * kb)) % bz.n1];
        float v;
        if ((row & 1) == 0) v = part == 0 ? d.r : -d.i;
        else v = part == 0 ? d.i : d.r;
        apack3[i] = v;
    }
}

void launch_bz_pack_stages(const BzArgs &bz, const cf *d_tw_n1, float *d_apack1, float *d_apack3, hipStream_t s)
{
    const int64_t n_a1 = (int64_t)bz.a * bz.n_tiles1 * 64, n_a3 = (int64_t)bz.a * 16 * 64, count = n_a1 > n_a3 ? n_a1 : n_a3;
    hipLaunchKernelGGL(bz_pack_stages_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, bz, d_tw_n1, d_apack1, d_apack3);
}

// coefficient image [r][tile][lane] of the length-n1 DFT for the MFMA A operand: lane l supplies
// A[row = 32 tile + (l & 31)][k = 2 r + (l >> 5)]; row 2 k1 = Re row (dr, -di), 2 k1 + 1 = Im row (di, dr);
// rows k1_first .. k1_first + k1_count - 1; n_tiles is a multiple of 3, rows past 2 k1_count are zero.  Filled on the
// device from T_n1 (n1^2 entries: milliseconds of host time for a long clip, and every file brings its own length).
__global__ __launch_bounds__(256) void bz_pack_kernel(int n1, int k1_first, int k1_count, const cf *__restrict__ tw_n1, int n_tiles,
                                                      float *__restrict__ apack)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)n1 * n_tiles * 64) return;
    const int l = (int)(i & 63), t = (int)((i >> 6) % n_tiles), r = (int)((i >> 6) / n_tiles);
    const int row = 32 * t + (l & 31), cr = row >> 1, part = l >> 5;
    float v = 0.0f;
    if (cr < k1_count) {
        const cf d = tw_n1[((int64_t)r * (k1_first + cr)) % n1];
        if ((row & 1) == 0) v = part == 0 ? d.r : -d.i;
        else v = part == 0 ? d.i : d.r;
    }
    apack[i] = v;
}

void launch_bz_pack_coefficients(int n1, int k1_first, int k1_count, const cf *d_tw_n1, int n_tiles, float *d_apack, hipStream_t s)
{
    const int64_t count = (int64_t)n1 * n_tiles * 64;
    hipLaunchKernelGGL(bz_pack_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, n1, k1_first, k1_count, d_tw_n1, n_tiles,
                       d_apack);
}

// ---- the tables of S15, generated where they are used -------------------------------------------------------------
// A corpus of real recordings has a length of its own per file, so the tables of a length are built once per file:
// on the host (double DFT of length L) that cost more than the file's whole extraction.  Here they take one pass of
// double arithmetic per element and one forward transform on the kernels above.
//
// S2b: cosine and sine of alpha in [0, pi / 4] in double, as Taylor polynomials in alpha^2 evaluated by explicit fma
// chains (truncation below 1e-19; coefficients = the correctly rounded 1 / k!): every operation is an IEEE double
// multiply or fma, so the oracle's C restatement is identical bit for bit, which a libm call would not promise.
__device__ __forceinline__ void bz_cos_sin(double x, double &c, double &s)
{
    const double z = x * x;
    double ps = 0x1.952c77030ad4ap-49;                 // 1 / 17!
    ps = __builtin_fma(ps, z, -0x1.ae7f3e733b81fp-41); // -1 / 15!
    ps = __builtin_fma(ps, z, 0x1.6124613a86d09p-33);
    ps = __builtin_fma(ps, z, -0x1.ae64567f544e4p-26);
    ps = __builtin_fma(ps, z, 0x1.71de3a556c734p-19);
    ps = __builtin_fma(ps, z, -0x1.a01a01a01a01ap-13);
    ps = __builtin_fma(ps, z, 0x1.1111111111111p-7);
    ps = __builtin_fma(ps, z, -0x1.5555555555555p-3); // -1 / 3!
    s = __builtin_fma(x * z, ps, x);
    double pc = -0x1.6827863b97d97p-53;               // -1 / 18!
    pc = __builtin_fma(pc, z, 0x1.ae7f3e733b81fp-45); // 1 / 16!
    pc = __builtin_fma(pc, z, -0x1.93974a8c07c9dp-37);
    pc = __builtin_fma(pc, z, 0x1.1eed8eff8d898p-29);
    pc = __builtin_fma(pc, z, -0x1.27e4fb7789f5cp-22);
    pc = __builtin_fma(pc, z, 0x1.a01a01a01a01ap-16);
    pc = __builtin_fma(pc, z, -0x1.6c16c16c16c17p-10);
    pc = __builtin_fma(pc, z, 0x1.5555555555555p-5);
    pc = __builtin_fma(pc, z, -0.5);
    c = __builtin_fma(z, pc, 1.0);
}

// e^{-2 pi i m / n}, 0 <= m < n: the octant reduction of S2 (plan.cpp twiddle_d) in integers, S2b for the octant's angle
__device__ __forceinline__ void bz_unit(int64_t m, int64_t n, double &re, double &im)
{
    const int64_t a = 8 * m;
    const int oct = (int)(a / n);
    const int64_t r = a - (int64_t)oct * n;
    const int64_t t = (oct & 1) ? (n - r) : r;
    const double alpha = 3.14159265358979323846 * (double)t / (double)(4 * n);
    double ca, sa, c, s;
    bz_cos_sin(alpha, ca, sa);
    switch (oct) {
    case 0: c = ca; s = sa; break;
    case 1: c = sa; s = ca; break;
    case 2: c = -sa; s = ca; break;
    case 3: c = -ca; s = sa; break;
    case 4: c = -ca; s = -sa; break;
    case 5: c = -sa; s = -ca; break;
    case 6: c = sa; s = -ca; break;
    default: c = ca; s = -sa; break;
    }
    re = c;
    im = -s;
}

struct BzGen {
    int64_t n, big_l; // samples, convolution length n1 n2
    cf *w, *tl, *wk;  // the tables of BzArgs, written here
    float *b;         // planar [2 r][n2pad]: the lags b[m mod L] = conj(w[m]), m in [kmin - (N - 1), kmax - 1], by residue
};

// element i = r n2 + t of the [n1][n2] tables stands for the flat index idx = r + n1 t
__global__ __launch_bounds__(256) void bz_tables_kernel(BzArgs bz, BzGen g)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double c, s;
    if (i < bz.kmax - bz.kmin) { // w[k] / L
        const int64_t k = bz.kmin + i;
        bz_unit((k * k) % (2 * g.n), 2 * g.n, c, s);
        g.wk[i] = cf{(float)(c / (double)g.big_l), (float)(s / (double)g.big_l)};
    }
    if (i >= g.big_l) return;
    const int r = (int)(i / bz.n2), t = (int)(i - (int64_t)r * bz.n2);
    const int64_t idx = r + (int64_t)bz.n1 * t;
    cf wv{0.0f, 0.0f};
    float bre = 0.0f, bim = 0.0f;
    if (idx < g.n) { // w[idx] = e^{-i pi idx^2 / N} = T_2N[idx^2 mod 2N]; idx < 2^26, the square is exact in 64 bits
        bz_unit((idx * idx) % (2 * g.n), 2 * g.n, c, s);
        wv = cf{(float)c, (float)s};
        if (idx <= bz.kmax - 1) { // the lag m = idx
            bre = (float)c;
            bim = (float)(-s);
        }
    }
    const int64_t neg = g.big_l - idx; // the lag m = idx - L = -neg; disjoint from the lags m = idx as L >= N + nk - 1
    if (neg <= g.n - 1 - bz.kmin) {
        bz_unit((neg * neg) % (2 * g.n), 2 * g.n, c, s);
        bre = (float)c;
        bim = (float)(-s);
    }
    g.w[i] = wv;
    g.b[(int64_t)2 * r * bz.n2pad + t] = bre;
    g.b[(int64_t)(2 * r + 1) * bz.n2pad + t] = bim;
    bz_unit(((int64_t)r * t) % g.big_l, g.big_l, c, s);
    g.tl[i] = cf{(float)c, (float)s};
}

size_t bz_plane_bytes(const BzArgs &bz, int n_clips) { return (size_t)n_clips * 2 * bz.n1 * bz.n2pad * sizeof(float); }

template <int MODE, int STEP>
static void launch_bz_cols_step(const BzArgs &bz, const float *in, float *out, cf *x, int n_clips, hipStream_t s)
{
    dim3 grid(((bz.n2 + 31) / 32 + 3) / 4, n_clips, bz.n_tiles2 / 3);
    hipLaunchKernelGGL((bz_cols_kernel<MODE, STEP>), grid, dim3(256), 0, s, bz, in, out, x);
}

template <int MODE>
static void launch_bz_cols_t(const BzArgs &bz, const float *in, float *out, cf *x, int n_clips, hipStream_t s)
{
    auto padded = [&](int step) { return ((bz.n1 + step - 1) / step + 1) / 2 * 2 * step; }; // residues the loop walks
    int best = 16;
    for (int step : {15, 14})
        if (padded(step) < padded(best)) best = step;
    if (best == 16)
        launch_bz_cols_step<MODE, 16>(bz, in, out, x, n_clips, s);
    else if (best == 15)
        launch_bz_cols_step<MODE, 15>(bz, in, out, x, n_clips, s);
    else
        launch_bz_cols_step<MODE, 14>(bz, in, out, x, n_clips, s);
}

static void bz_rows_attr()
{
    static PerDeviceOnce attr_set;
    if (attr_set.need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(bz_rows_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(bz_rows_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(bz_transpose_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(bz_cols2_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(bz_cols2_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        attr_set.mark();
    }
}

// d_pairs: pcm_pairs output for the [n2][n1] matrix padded with zeros -> planar Y' (bz_plane_bytes)
void launch_bz_rows_first(const RowsArgs &rows, const BzArgs &bz, const i16x2 *d_pairs, int n_clips, float *d_out, hipStream_t s)
{
    bz_rows_attr();
    hipLaunchKernelGGL(bz_rows_kernel<0>, dim3(n_clips, bz.n1), dim3(kBzThreads), (size_t)rows.n2 * sizeof(cf), s, rows, bz,
                       d_pairs, (const float *)nullptr, d_out);
}

// Y' -> C = conj(A Bhat), planar [k1][k2]; in and out distinct
void launch_bz_cols_full(const BzArgs &bz, const float *d_in, float *d_out, int n_clips, hipStream_t s)
{
    bz_rows_attr();
    hipLaunchKernelGGL(bz_cols2_kernel<0>, dim3(((bz.n2 + 31) / 32 + 3) / 4, n_clips, bz.n_tiles1), dim3(256), 64 * 1024, s, bz, d_in, d_out,
                       (cf *)nullptr);
}

// C [k1][k2] -> C' [r][t]; in and out distinct
void launch_bz_transpose(const BzArgs &bz, const float *d_in, float *d_out, int n_clips, hipStream_t s)
{
    bz_rows_attr();
    // time steps per workgroup (a power of two, at most 256): 128-byte runs with several workgroups per CU; fewer
    // when n1 is large (the tile lives in LDS)
    int lg = 5;
    while (lg > 0 && ((size_t)1 << lg) * bz.n1 * sizeof(float) > 48 * 1024) --lg;
    const int tsteps = 1 << lg;
    hipLaunchKernelGGL(bz_transpose_kernel, dim3((bz.n2 + tsteps - 1) / tsteps, n_clips), dim3(256),
                       (size_t)tsteps * bz.n1 * sizeof(float), s, bz, lg, d_in, d_out);
}

// C' -> Y'' planar; in and out distinct
void launch_bz_rows_second(const RowsArgs &rows, const BzArgs &bz, const float *d_in, int n_clips, float *d_out, hipStream_t s)
{
    bz_rows_attr();
    hipLaunchKernelGGL(bz_rows_kernel<1>, dim3(n_clips, bz.n1), dim3(kBzThreads), (size_t)rows.n2 * sizeof(cf), s, rows, bz,
                       (const i16x2 *)nullptr, d_in, d_out);
}

// Fills bz.w, bz.tl, bz.wk and bz.bhat (device memory of the plan, sizes as in kernels.h) for clips of n samples:
// chirp, T_L and the lags by bz_tables_kernel, then Bhat = the first transform (rows, T_L, columns) of the lags.
// d_b and d_y: two planar buffers of bz_plane_bytes(bz, 1) each.  bz.apack must be in place.
void launch_bz_make_tables(const RowsArgs &rows, const BzArgs &bz, int64_t n, float *d_b, float *d_y, hipStream_t s)
{
    bz_rows_attr();
    BzGen g;
    g.n = n;
    g.big_l = (int64_t)bz.n1 * bz.n2;
    g.w = const_cast<cf *>(bz.w);
    g.tl = const_cast<cf *>(bz.tl);
    g.wk = const_cast<cf *>(bz.wk);
    g.b = d_b;
    const int64_t nk = bz.kmax - bz.kmin, count = g.big_l > nk ? g.big_l : nk;
    hipLaunchKernelGGL(bz_tables_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, bz, g);
    hipLaunchKernelGGL(bz_rows_kernel<1>, dim3(1, bz.n1), dim3(kBzThreads), (size_t)rows.n2 * sizeof(cf), s, rows, bz,
                       (const i16x2 *)nullptr, (const float *)d_b, d_y);
    hipLaunchKernelGGL(bz_cols2_kernel<2>, dim3(((bz.n2 + 31) / 32 + 3) / 4, 1, bz.n_tiles1), dim3(256), 64 * 1024, s, bz, (const float *)d_y,
                       (float *)nullptr, const_cast<cf *>(bz.bhat));
}

// Y'' -> x [n_clips][kmax - kmin]
void launch_bz_cols_last(const BzArgs &bz, const float *d_in, int n_clips, cf *d_x, hipStream_t s)
{
    launch_bz_cols_t<1>(bz, d_in, nullptr, d_x, n_clips, s);
}

} // namespace hpfw
