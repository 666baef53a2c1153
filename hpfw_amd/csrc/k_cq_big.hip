// k_cq_big.hip -- chirp-z bands whose length P exceeds the LDS (clips longer than ~50 s): the same
// transform as cq_kernel (k_cq.hip), pass for pass, in three parts:
//   cq_big_dif_kernel    the first `outer` radix-4 DIF passes through global memory, one butterfly per
//                        thread (the first one forms a[i] = X[s_j + i] G_j[i] on the fly)
//   cq_big_local_kernel  every block of len0 = P / 4^outer points in LDS: the remaining DIF passes, the
//                        product with V, the first inverse passes -- fft_lds.h's cq_transform on the
//                        block, with butterfly tables that hold the twiddles of length P
//   cq_big_idit_kernel   the last `outer` inverse DIT passes through global memory (the last one
//                        stores |b[c]| or its dB term for c < C)
// A radix-4 pass at sub-length L >= 4 len0 only combines elements len0 or more apart, and the passes
// below act inside blocks of len0, so the split changes no operand and no rounding.
#include "kernels.h"
#include "db_spec.h"

namespace hpfw {

extern __shared__ __align__(16) unsigned char smem_raw[];

__device__ __forceinline__ float wave_max_f(float v)
{
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) v = fmaxf(v, __shfl_xor(v, s));
    return v;
}

struct CqBigArgs {
    int p;            // transform length
    int len;          // sub-length L of this pass
    const cf *tw;     // T_p [p]
    const int *band;  // bands of the class
    int64_t work_clip_pitch; // complex elements per clip in work (n_bands * p)
};

// one DIF radix-4 pass at sub-length len over every band of the class and every clip.
// FIRST: the input is the windowed slice of the forward bins instead of the work array.
template <bool FIRST>
__global__ __launch_bounds__(256) void cq_big_dif_kernel(CqBigArgs a, CqPlanDev cp, const cf *__restrict__ x,
                                                         cf *__restrict__ work)
{
    const int bf = blockIdx.x * 256 + threadIdx.x; // butterfly index: p / 4 of them
    if (bf >= a.p / 4) return;
    const int bi = blockIdx.y, clip = blockIdx.z;
    const int quarter = a.len / 4;
    const int blk = bf / quarter, j = bf - blk * quarter;
    const int i0 = blk * a.len + j;
    cf *w = work + (int64_t)clip * a.work_clip_pitch + (int64_t)bi * a.p;
    cf u[4];
    if (FIRST) {
        const int jb = a.band[bi], lg = cp.lg[jb];
        const XsBand xs{cp.view(x, clip), cp.start[jb]};
        const cf *g = cp.g + cp.g_off[jb];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = i0 + q * quarter;
            u[q] = i < lg ? c_mul(xs(i), g[i]) : cf{0.0f, 0.0f};
        }
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) u[q] = w[i0 + q * quarter];
    }
    Dft<4>::run(u);
    const int64_t ts = a.p / a.len;
    w[i0] = u[0];
#pragma unroll
    for (int s = 1; s < 4; ++s) w[i0 + s * quarter] = c_mul(u[s], a.tw[ts * j * s]);
}

// one inverse DIT radix-4 pass at sub-length len.  LAST (len == p): the outputs with index < c go to
// the spectrogram as magnitudes or dB terms, nothing is written back.
template <bool LAST, bool DBT>
__global__ __launch_bounds__(256) void cq_big_idit_kernel(CqBigArgs a, CqPlanDev cp, cf *__restrict__ work,
                                                          float *__restrict__ mag)
{
    const int bf = blockIdx.x * 256 + threadIdx.x;
    if (bf >= a.p / 4) return;
    const int bi = blockIdx.y, clip = blockIdx.z;
    const int quarter = a.len / 4;
    const int blk = bf / quarter, j = bf - blk * quarter;
    const int i0 = blk * a.len + j;
    if (LAST && j >= cp.c) return; // all four outputs j + s p / 4 lie past the kept samples
    cf *w = work + (int64_t)clip * a.work_clip_pitch + (int64_t)bi * a.p;
    const int64_t ts = a.p / a.len;
    cf v[4];
    v[0] = w[i0];
#pragma unroll
    for (int q = 1; q < 4; ++q) v[q] = c_mulc(w[i0 + q * quarter], a.tw[ts * j * q]);
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = {v[q].i, v[q].r};
    Dft<4>::run(v);
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = {v[q].i, v[q].r};
    if (LAST) {
        float *out = mag + ((int64_t)clip * kBins + a.band[bi]) * cp.c;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int i = i0 + s * quarter;
            if (i < cp.c) {
                const float m = __builtin_sqrtf(__builtin_fmaf(v[s].r, v[s].r, v[s].i * v[s].i));
                out[i] = DBT ? db_term(m * m) : m;
            }
        }
    } else {
#pragma unroll
        for (int s = 0; s < 4; ++s) w[i0 + s * quarter] = v[s];
    }
}

constexpr int cq_big_threads(int len0) { return len0 % 3 == 0 ? len0 / 12 : len0 / 16; } // 256 .. 512

// one block of LEN0 points of one band of one clip: DIF passes below LEN0, times V, inverse passes up to LEN0
template <int LEN0>
__global__ __launch_bounds__(cq_big_threads(LEN0)) void cq_big_local_kernel(int p, CqTwiddles tw, const cf *__restrict__ vrev,
                                                                            int64_t work_clip_pitch, cf *__restrict__ work)
{
    using S = Size<LEN0>;
    cf *lds = reinterpret_cast<cf *>(smem_raw);
    const int tid = threadIdx.x;
    constexpr int nthreads = cq_big_threads(LEN0);
    const int blk = blockIdx.x, bi = blockIdx.y, clip = blockIdx.z;
    cf *w = work + (int64_t)clip * work_clip_pitch + (int64_t)bi * p + (int64_t)blk * LEN0;
    for (int i = tid; i < LEN0; i += nthreads) lds[pad16(i)] = w[i];
    __syncthreads();
    cq_transform<LEN0, LEN0, 0>(lds, tw, nthreads, LEN0, LEN0, vrev + (int64_t)blk * LEN0);
    for (int i = tid; i < LEN0; i += nthreads) w[i] = lds[pad16(i)];
    (void)S::N;
}

// largest value of each band of the class -> wavemax[clip][band][0], the other slots -inf
__global__ __launch_bounds__(256) void cq_big_max_kernel(const float *__restrict__ mag, int c, const int *__restrict__ band,
                                                         float *__restrict__ wavemax)
{
    __shared__ float part[4];
    const int jb = band[blockIdx.x], clip = blockIdx.y;
    const float *m = mag + ((int64_t)clip * kBins + jb) * c;
    float mx = -INFINITY;
    for (int i = threadIdx.x; i < c; i += 256) mx = fmaxf(mx, m[i]);
    mx = wave_max_f(mx);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = mx;
    __syncthreads();
    float *slot = wavemax + ((int64_t)clip * kBins + jb) * kCqMaxWaves;
    if (threadIdx.x == 0) slot[0] = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
    if (threadIdx.x >= 1 && threadIdx.x < kCqMaxWaves) slot[threadIdx.x] = -INFINITY;
}

template <int LEN0>
static void launch_local_t(const CqClassDev &cc, int n_clips, int64_t pitch, cf *d_work, hipStream_t s)
{
    static PerDeviceOnce attr_set;
    if (attr_set.need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(cq_big_local_kernel<LEN0>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set.mark();
    }
    dim3 grid(cc.p / LEN0, cc.n_bands, n_clips);
    hipLaunchKernelGGL(cq_big_local_kernel<LEN0>, grid, dim3(cq_big_threads(LEN0)), (size_t)Size<LEN0>::DATA * sizeof(cf), s,
                       cc.p, cc.gtw, cc.vrev, pitch, d_work);
}

size_t cq_big_work_bytes(const CqClassDev &cc, int n_clips)
{
    return cc.outer ? (size_t)n_clips * cc.n_bands * cc.p * sizeof(cf) : 0;
}

void launch_cq_big_class(const CqPlanDev &cp, const CqClassDev &cc, const cf *d_x, int n_clips, cf *d_work, float *d_mag,
                         float *d_wavemax, bool db_term_out, hipStream_t s)
{
    CqBigArgs a;
    a.p = cc.p;
    a.tw = cc.tw;
    a.band = cc.band;
    a.work_clip_pitch = (int64_t)cc.n_bands * cc.p;
    dim3 grid((cc.p / 4 + 255) / 256, cc.n_bands, n_clips);
    int len = cc.p;
    for (int q = 0; q < cc.outer; ++q, len /= 4) {
        a.len = len;
        if (q == 0)
            hipLaunchKernelGGL(cq_big_dif_kernel<true>, grid, dim3(256), 0, s, a, cp, d_x, d_work);
        else
            hipLaunchKernelGGL(cq_big_dif_kernel<false>, grid, dim3(256), 0, s, a, cp, d_x, d_work);
    }
    switch (cc.len0) {
    case 3072: launch_local_t<3072>(cc, n_clips, a.work_clip_pitch, d_work, s); break;
    case 4096: launch_local_t<4096>(cc, n_clips, a.work_clip_pitch, d_work, s); break;
    case 6144: launch_local_t<6144>(cc, n_clips, a.work_clip_pitch, d_work, s); break;
    default: launch_local_t<8192>(cc, n_clips, a.work_clip_pitch, d_work, s); break; // plan.cpp admits only these four
    }
    len *= 4;
    for (int q = cc.outer - 1; q >= 0; --q, len *= 4) {
        a.len = len;
        if (q > 0)
            hipLaunchKernelGGL((cq_big_idit_kernel<false, false>), grid, dim3(256), 0, s, a, cp, d_work, d_mag);
        else if (db_term_out)
            hipLaunchKernelGGL((cq_big_idit_kernel<true, true>), grid, dim3(256), 0, s, a, cp, d_work, d_mag);
        else
            hipLaunchKernelGGL((cq_big_idit_kernel<true, false>), grid, dim3(256), 0, s, a, cp, d_work, d_mag);
    }
    hipLaunchKernelGGL(cq_big_max_kernel, dim3(cc.n_bands, n_clips), dim3(256), 0, s, d_mag, cp.c, cc.band, d_wavemax);
}

} // namespace hpfw
