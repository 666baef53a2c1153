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
//   bz_cols<0>   A[n2 k1 + k2] = sum_r T_n1[r k1] Y'[r][k2] (f32 MFMA, all n1 rows), C = conj(A Bhat) -> C planar [k1][k2]
//   bz_transpose the flat index j = n2 k1 + k2 = r + n1 t regrouped by residue                  -> C' planar [r][t]
//   bz_rows<1>   per residue r: FFT_n2 over t in LDS, times T_L[r k2]                          -> Y'' planar
//   bz_cols<1>   F[n2 k1 + k2] = sum_r T_n1[r k1] Y''[r][k2] for the rows k1 that hold consumed bins only (about a
//                tenth of them, as in fwd_cols); X[k] = conj(F[k]) w[k] / L
// Only the first transform's column stage is a full n1 x n1 contraction.  Arithmetic order = the oracle's, bit for
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
        const i16x2 *__restrict__ src = pairs + ((int64_t)clip * ((bz.n1 + 1) / 2) + (row >> 1)) * n2;
        const cf *__restrict__ w = bz.w + (int64_t)row * n2;
        const bool odd = row & 1;
        constexpr int kLd = 13;
        for (int t0 = tid; t0 < n2; t0 += kLd * nthreads) {
            i16x2 p[kLd];
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
                    const float xs = (float)(odd ? p[e].y : p[e].x) / 32768.0f;
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

// D[row][col] = sum_k A[row][k] B[k][col], k = 2 r + part: B = a planar buffer (row 2r = Re, 2r + 1 = Im of
// residue r, n2pad columns), A = the DFT coefficients of ALL n1 output rows (Re / Im row pairs), packed by
// bz_pack_kernel.  One wave = 32 columns x 3 row tiles of 32 (16 complex rows each); blockIdx.z = the
// group of 3 row tiles.  Same register-blocked operand stream as fwd_cols_kernel (k_forward.hip).
// MODE 0: all n1 rows, out = conj(D Bhat[k1][k2]) planar; MODE 1: the rows k1lo .. k1lo + k1n - 1 that hold consumed
// bins, x[k - kmin] = conj(D) w[k] / L for k = n2 k1 + k2 in [kmin, kmax); MODE 2 (table generation, one "clip"): all
// n1 rows, x[n2 k1 + k2] = D as it stands -- Bhat, the transform of the conjugate chirp's lags.
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
    const int n_tiles = MODE == 1 ? bz.n_tiles2 : bz.n_tiles;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(MODE == 1 ? bz.apack2 : bz.apack), (short)0, bz.n1 * n_tiles * 256, 0x00020000); // [r][tile][lane]
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
    if (MODE == 0) {
        float *__restrict__ o = out + clip * plane;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int k1 = (tile0 + t) * 16 + (q & 1) + 4 * (q >> 1) + 2 * hb;
                if (k1 < bz.n1) {
                    const cf v = c_mul(cf{acc[t][2 * q], acc[t][2 * q + 1]}, bz.bhat[(int64_t)k1 * bz.n2 + k2]);
                    o[(int64_t)2 * k1 * bz.n2pad + k2] = v.r;
                    o[(int64_t)(2 * k1 + 1) * bz.n2pad + k2] = -v.i;
                }
            }
        }
    } else if (MODE == 2) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int k1 = (tile0 + t) * 16 + (q & 1) + 4 * (q >> 1) + 2 * hb;
                if (k1 < bz.n1) x[(int64_t)k1 * bz.n2 + k2] = cf{acc[t][2 * q], acc[t][2 * q + 1]};
            }
        }
    } else {
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
    dim3 grid(((bz.n2 + 31) / 32 + 3) / 4, n_clips, (MODE == 1 ? bz.n_tiles2 : bz.n_tiles) / 3);
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
    launch_bz_cols_t<0>(bz, d_in, d_out, nullptr, n_clips, s);
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
    launch_bz_cols_t<2>(bz, d_y, nullptr, const_cast<cf *>(bz.bhat), 1, s);
}

// Y'' -> x [n_clips][kmax - kmin]
void launch_bz_cols_last(const BzArgs &bz, const float *d_in, int n_clips, cf *d_x, hipStream_t s)
{
    launch_bz_cols_t<1>(bz, d_in, nullptr, d_x, n_clips, s);
}

} // namespace hpfw
