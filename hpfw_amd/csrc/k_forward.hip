// k_forward.hip -- a1 + the forward half of a2: int16 PCM -> forward DFT bins [kmin, kmax).
//
// Replaces essentia MonoLoader (PCM16 mono 44.1 kHz case) and the length-N FFT inside essentia
// NSGConstantQ::compute, called from CQT<>::spectrogram (reference include/hpfw/spectrum/cqt.h:45-52,
// 66-71).  N = n1 * n2:
//   fwd_rows : one workgroup per pair of residues (a, a+1) mod n1: the two real sequences
//              x[a + n1 t], x[a+1 + n1 t] ride one complex length-n2 FFT held in LDS, are split by
//              Hermitian symmetry, multiplied by T_N[a k2] and written as half spectra.
//   fwd_cols : the remaining length-n1 DFT, only for the rows k1 that hold consumed bins, as one fma
//              chain per output over a ascending (only ~10 % of the N/2 bins feed the 121 bands).
//   pcm_pairs: coalescing pre-pass.  The clip is an [n2][n1] row-major matrix of samples; residue pair
//              (2p, 2p+1) is a 4-byte column of it.  A workgroup transposes a tile of 64 time steps
//              through LDS so that fwd_rows reads its pair stream contiguously.
#include "kernels.h"

namespace hpfw {

extern __shared__ __align__(16) unsigned char smem_raw[];

constexpr int kPairsTile = 64;

__global__ __launch_bounds__(256) void pcm_pairs_kernel(int64_t n, int n1, int n2, const int16_t *__restrict__ pcm,
                                                        i16x2 *__restrict__ pairs)
{
    int16_t *tile = reinterpret_cast<int16_t *>(smem_raw); // [kPairsTile][n1]
    const int tid = threadIdx.x;
    const int clip = blockIdx.y;
    const int t0 = blockIdx.x * kPairsTile;
    const int nt = min(kPairsTile, n2 - t0);
    const int np = (n1 + 1) / 2;
    const int16_t *src = pcm + (int64_t)clip * n + (int64_t)t0 * n1;
    for (int i = tid; i < nt * n1; i += 256) tile[i] = src[i];
    __syncthreads();
    i16x2 *dst = pairs + (int64_t)clip * np * n2 + t0;
    for (int i = tid; i < np * kPairsTile; i += 256) {
        const int p = i / kPairsTile, tt = i - p * kPairsTile;
        if (tt < nt) {
            i16x2 v;
            v.x = tile[tt * n1 + 2 * p];
            v.y = (2 * p + 1 < n1) ? tile[tt * n1 + 2 * p + 1] : (short)0;
            dst[(int64_t)p * n2 + tt] = v;
        }
    }
}

constexpr int kFwdThreads = 768;

// one workgroup = one residue pair of one clip; the body (fft_rows.h) is shared with tests/emu
__global__ __launch_bounds__(kFwdThreads) void fwd_rows_kernel(RowsArgs a, const i16x2 *__restrict__ pairs,
                                                               cf *__restrict__ yp)
{
    cf *lds = reinterpret_cast<cf *>(smem_raw);
    const int p = blockIdx.x;
    const int clip = blockIdx.y;
    const int np = (a.n1 + 1) / 2;
    const int a0 = 2 * p;
    cf *ya = yp + ((int64_t)clip * a.n1 + a0) * a.h;
    cf *yb = (a0 + 1 < a.n1) ? ya + a.h : nullptr;
    rows_body(lds, a, (int)blockDim.x, pairs + ((int64_t)clip * np + p) * a.n2, a0, ya, yb);
}

constexpr int kColsThreads = 256;
constexpr int kColsRows = 4; // direct rows per thread (+ the same number of mirrored rows)

__global__ __launch_bounds__(kColsThreads) void fwd_cols_kernel(FwdPlanDev fp, const cf *__restrict__ yp,
                                                                cf *__restrict__ x)
{
    cf *tw1 = reinterpret_cast<cf *>(smem_raw);
    const int tid = threadIdx.x;
    for (int i = tid; i < fp.n1; i += kColsThreads) tw1[i] = fp.tw_n1[i];
    __syncthreads();
    const int k2 = blockIdx.x * kColsThreads + tid;
    const int clip = blockIdx.z;
    if (k2 >= fp.h) return;
    const int n1 = fp.n1;
    int k1d[kColsRows], k1m[kColsRows], id[kColsRows], im[kColsRows];
    float dr[kColsRows], di[kColsRows], mr[kColsRows], mi[kColsRows];
#pragma unroll
    for (int i = 0; i < kColsRows; ++i) {
        int k1 = fp.k1lo + blockIdx.y * kColsRows + i;
        if (k1 > fp.k1hi) k1 = fp.k1hi; // surplus rows repeat the last one and are not stored
        k1d[i] = k1;
        k1m[i] = n1 - 1 - k1;
        id[i] = 0;
        im[i] = 0;
        dr[i] = di[i] = mr[i] = mi[i] = 0.0f;
    }
    const cf *y = yp + (int64_t)clip * n1 * fp.h + k2;
    for (int a = 0; a < n1; ++a) {
        const cf yv = y[(int64_t)a * fp.h];
#pragma unroll
        for (int i = 0; i < kColsRows; ++i) {
            const cf d = tw1[id[i]];
            dr[i] = __builtin_fmaf(d.r, yv.r, dr[i]);
            dr[i] = __builtin_fmaf(-d.i, yv.i, dr[i]);
            di[i] = __builtin_fmaf(d.r, yv.i, di[i]);
            di[i] = __builtin_fmaf(d.i, yv.r, di[i]);
            id[i] += k1d[i];
            if (id[i] >= n1) id[i] -= n1;
            const cf e = tw1[im[i]];
            mr[i] = __builtin_fmaf(e.r, yv.r, mr[i]);
            mr[i] = __builtin_fmaf(-e.i, yv.i, mr[i]);
            mi[i] = __builtin_fmaf(e.r, yv.i, mi[i]);
            mi[i] = __builtin_fmaf(e.i, yv.r, mi[i]);
            im[i] += k1m[i];
            if (im[i] >= n1) im[i] -= n1;
        }
    }
    const int nk = fp.kmax - fp.kmin;
    cf *xo = x + (int64_t)clip * nk;
    const bool has_mirror = (k2 >= 1) && (k2 <= fp.n2 - fp.h);
#pragma unroll
    for (int i = 0; i < kColsRows; ++i) {
        const int k1 = fp.k1lo + blockIdx.y * kColsRows + i;
        if (k1 > fp.k1hi) continue;
        const int64_t kd = (int64_t)fp.n2 * k1 + k2;
        if (kd >= fp.kmin && kd < fp.kmax) xo[kd - fp.kmin] = {dr[i], di[i]};
        if (has_mirror) { // X[k] = conj(X[N - k])
            const int64_t km = (int64_t)fp.n2 * k1 + (fp.n2 - k2);
            if (km >= fp.kmin && km < fp.kmax) xo[km - fp.kmin] = {mr[i], -mi[i]};
        }
    }
}

static int g_rows_lds_set = 0;

void launch_pcm_pairs(int64_t n, int n1, int n2, const int16_t *d_pcm, int n_clips, i16x2 *d_pairs, hipStream_t s)
{
    dim3 grid((n2 + kPairsTile - 1) / kPairsTile, n_clips);
    hipLaunchKernelGGL(pcm_pairs_kernel, grid, dim3(256), (size_t)kPairsTile * n1 * sizeof(int16_t), s, n, n1, n2,
                       d_pcm, d_pairs);
}

size_t fwd_rows_lds_bytes(const RowsArgs &a) { return ((size_t)a.n2 + (a.quad ? a.n2 / 4 : 0)) * sizeof(cf); }

void launch_fwd_rows(const RowsArgs &a, const i16x2 *d_pairs, int n_clips, cf *d_yp, hipStream_t s)
{
    if (!g_rows_lds_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(fwd_rows_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        g_rows_lds_set = 1;
    }
    dim3 grid((a.n1 + 1) / 2, n_clips);
    hipLaunchKernelGGL(fwd_rows_kernel, grid, dim3(kFwdThreads), fwd_rows_lds_bytes(a), s, a, d_pairs, d_yp);
}

void launch_fwd_cols(const FwdPlanDev &fp, const cf *d_yp, int n_clips, cf *d_x, hipStream_t s)
{
    const int rows = fp.k1hi - fp.k1lo + 1;
    dim3 grid((fp.h + kColsThreads - 1) / kColsThreads, (rows + kColsRows - 1) / kColsRows, n_clips);
    hipLaunchKernelGGL(fwd_cols_kernel, grid, dim3(kColsThreads), (size_t)fp.n1 * sizeof(cf), s, fp, d_yp, d_x);
}

} // namespace hpfw
