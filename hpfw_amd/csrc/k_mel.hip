// k_mel.hip -- f3: the Mel front-end, hpfw::spectrum::MelSpectrogram<44100, 33, 4410, 441>::spectrogram
// (reference include/hpfw/spectrum/mel.h:34-104): essentia FrameCutter(4410, 441) -> Windowing(hann)
// -> Spectrum -> MelBands(33) per frame, silent frames dropped (mel.h:94-96), power_to_db over the
// kept columns (mel.h:103).  essentia is not vendored; DESIGN.md appendix B restates the algorithms
// (parity unpinned, like the constant-Q).
//
//   mel_blocksum_kernel   sum of pcm^2 over blocks of one hop (441 samples): a frame is exactly ten
//                         blocks, so its instant power -- the silence test -- is exact in integers
//   mel_keep_kernel       keep[f], the column pos[f] of every kept frame, their count
//   stft_rows_kernel      frames in pairs through the row transform of the forward FFT (fft_rows.h:
//                         4410 = 7 2 7 3 5 3 in LDS, Hermitian split); the loader applies the
//                         zero-phase Hann window on the fly -- the "framed audio" is never stored
//   mel_bands_kernel      [33 x 2206] . [2206 x frames] on v_mfma_f32_32x32x2_f32, the power spectrum
//                         formed from the split spectra on the way into LDS
//   mel_max_kernel, mel_db_kernel   power_to_db of the kept columns, compacted to the front
#include "kernels.h"
#include "db_spec.h"

namespace hpfw {

extern __shared__ __align__(16) unsigned char smem_raw[];

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kMelHalf = kMelFrame / 2;          // 2205 = 5 hops
constexpr int kMelHpad = 2208;                   // row stride of the split spectra (>= 2206, multiple of 32)
constexpr int kMelRowsPad = 64;                  // 33 bands padded to two MFMA row tiles
constexpr int kMelThreads = 512;

__global__ __launch_bounds__(256) void mel_blocksum_kernel(const int16_t *__restrict__ pcm, int64_t n, int n_blk,
                                                           int64_t *__restrict__ blk)
{
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), clip = blockIdx.y, lane = threadIdx.x & 63;
    if (b >= n_blk) return;
    const int16_t *x = pcm + (int64_t)clip * n;
    long long s = 0;
    for (int i = lane; i < kMelHop; i += 64) {
        const int64_t idx = (int64_t)b * kMelHop + i;
        if (idx < n) s += (long long)x[idx] * x[idx];
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) blk[(int64_t)clip * n_blk + b] = s;
}

// one workgroup per clip: keep[f] = the frame is not silent (essentia isSilent: instant power < 1e-10,
// i.e. sum pcm^2 <= 473), pos[f] = its column among the kept ones, count[clip]
__global__ __launch_bounds__(256) void mel_keep_kernel(const int64_t *__restrict__ blk, int n_blk, int n_frames,
                                                       int *__restrict__ pos, int *__restrict__ count)
{
    __shared__ int part[256];
    __shared__ int base_s;
    const int clip = blockIdx.x, tid = threadIdx.x;
    const int64_t *e = blk + (int64_t)clip * n_blk;
    int *p = pos + (int64_t)clip * n_frames;
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (int f0 = 0; f0 < n_frames; f0 += 256) {
        const int f = f0 + tid;
        int k = 0;
        if (f < n_frames) {
            long long s = 0;
            for (int b = f - 5; b < f + 5; ++b) // the frame starts at sample 441 f - 2205
                if (b >= 0 && b < n_blk) s += e[b];
            k = s > 473 ? 1 : 0;
        }
        part[tid] = k;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) { // inclusive scan
            const int v = tid >= o ? part[tid - o] : 0;
            __syncthreads();
            part[tid] += v;
            __syncthreads();
        }
        if (f < n_frames) p[f] = k ? base_s + part[tid] - 1 : -1;
        __syncthreads();
        if (tid == 255) base_s += part[255];
        __syncthreads();
    }
    if (tid == 0) count[clip] = base_s;
}

// the two frames of a pair as the row transform's two real sequences: time step t of the transform is
// windowed sample (t + 2205) mod 4410 of the frame (essentia Windowing, zeroPhase = true)
struct FrameLoad {
    const int16_t *pcm;
    int64_t n, start0;
    bool has1;
    const float *win;
    struct Raw {
        short a, b;
    };
    __device__ __forceinline__ Raw raw(int t) const
    {
        const int i = t < kMelFrame - kMelHalf ? t + kMelHalf : t - (kMelFrame - kMelHalf);
        const int64_t s0 = start0 + i, s1 = s0 + kMelHop;
        Raw r;
        r.a = (s0 >= 0 && s0 < n) ? pcm[s0] : (short)0;
        r.b = (has1 && s1 >= 0 && s1 < n) ? pcm[s1] : (short)0;
        return r;
    }
    __device__ __forceinline__ cf conv(Raw r, int t) const
    {
        const int i = t < kMelFrame - kMelHalf ? t + kMelHalf : t - (kMelFrame - kMelHalf);
        const float w = win[i];
        return {((float)r.a / 32768.0f) * w, ((float)r.b / 32768.0f) * w};
    }
};

using Groups4410 = StaticGroups<7, 2, 7, 3, 5, 3>;

// workgroup = one pair of frames of one clip; output rows (Re, Im) of frame f at y[clip][2 f .. 2 f + 1][hpad]
template <class Groups>
__global__ __launch_bounds__(kMelThreads, 4) void stft_rows_kernel(RowsArgs a, const int16_t *__restrict__ pcm, int64_t n,
                                                                   int n_frames, int frames_pad,
                                                                   const float *__restrict__ win, float *__restrict__ y)
{
    cf *lds = reinterpret_cast<cf *>(smem_raw);
    const int f0 = 2 * blockIdx.x, clip = blockIdx.y;
    FrameLoad ld;
    ld.pcm = pcm + (int64_t)clip * n;
    ld.n = n;
    ld.start0 = (int64_t)f0 * kMelHop - kMelHalf;
    ld.has1 = f0 + 1 < n_frames;
    ld.win = win;
    float *ya = y + ((int64_t)clip * frames_pad + f0) * 2 * kMelHpad;
    rows_body_from<Groups>(lds, a, kMelThreads, ld, 0, ya, ya + 2 * kMelHpad);
}

// power[clip][band][f] = fma chain over the bins j of coeff[band][j] (Re[f][j]^2 (+) Im[f][j]^2), all frames.
// Workgroup = 64 band rows (33 used) x 128 frames; wave = 64 x 32; K in chunks of 32 bins through LDS.
__global__ __launch_bounds__(256) void mel_bands_kernel(const float *__restrict__ y, const float *__restrict__ cpack,
                                                        int n_frames, int frames_pad, float *__restrict__ power)
{
    __shared__ float pt[32 * 129]; // [bin in chunk][frame], odd stride
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hb = lane >> 5, li = lane & 31;
    const int fw0 = blockIdx.x * 128, clip = blockIdx.y;
    const float *yc = y + (int64_t)clip * frames_pad * 2 * kMelHpad;
    f32x16 acc[2] = {f32x16{0}, f32x16{0}};
    for (int j0 = 0; j0 < kMelHpad; j0 += 32) {
        __syncthreads();
#pragma unroll 4
        for (int e = 0; e < 16; ++e) {
            const int idx = tid + 256 * e, fl = idx >> 5, j = idx & 31;
            const int f = fw0 + fl;
            float p = 0.0f;
            if (f < n_frames && j0 + j < kMelBins) {
                const float re = yc[((int64_t)2 * f) * kMelHpad + j0 + j], im = yc[((int64_t)2 * f + 1) * kMelHpad + j0 + j];
                const float m = __builtin_sqrtf(__builtin_fmaf(re, re, im * im)); // essentia Spectrum: the magnitude
                p = m * m;                                                         // MelBands, type "power"
            }
            pt[j * 129 + fl] = p;
        }
        __syncthreads();
#pragma unroll 4
        for (int st = 0; st < 16; ++st) {
            const int j = 2 * st + hb;
            const float b = pt[j * 129 + wave * 32 + li];
            const float a0 = cpack[(int64_t)(j0 + j) * kMelRowsPad + li], a1 = cpack[(int64_t)(j0 + j) * kMelRowsPad + 32 + li];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc[1], 0, 0, 0);
        }
    }
    const int f = fw0 + wave * 32 + li;
    float *pc = power + (int64_t)clip * kMelBands * n_frames;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int band = t * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * hb;
            if (band < kMelBands && f < n_frames) pc[(int64_t)band * n_frames + f] = acc[t][reg];
        }
}

__device__ __forceinline__ float wave_max_mel(float v)
{
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) v = fmaxf(v, __shfl_xor(v, s));
    return v;
}

// pmax[clip] = the largest band power over the kept frames (0 when nothing is kept)
__global__ __launch_bounds__(256) void mel_max_kernel(const float *__restrict__ power, const int *__restrict__ pos,
                                                      int n_frames, float *__restrict__ pmax)
{
    __shared__ float part[4];
    const int clip = blockIdx.x, tid = threadIdx.x;
    const float *pc = power + (int64_t)clip * kMelBands * n_frames;
    const int *p = pos + (int64_t)clip * n_frames;
    float mx = 0.0f;
    for (int64_t i = tid; i < (int64_t)kMelBands * n_frames; i += 256)
        if (p[i % n_frames] >= 0) mx = fmaxf(mx, pc[i]);
    mx = wave_max_mel(mx);
    if ((tid & 63) == 0) part[tid >> 6] = mx;
    __syncthreads();
    if (tid == 0) pmax[clip] = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
}

// out[clip][band][pos[f]] = max(t(power) - t(pmax), -80)  (convert.h:7-16), kept frames only
__global__ __launch_bounds__(256) void mel_db_kernel(const float *__restrict__ power, const int *__restrict__ pos,
                                                     const float *__restrict__ pmax, int n_frames, float *__restrict__ out)
{
    const int clip = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)kMelBands * n_frames) return;
    const int band = (int)(i / n_frames), f = (int)(i - (int64_t)band * n_frames);
    const int c = pos[(int64_t)clip * n_frames + f];
    if (c < 0) return;
    const float ref = db_term(pmax[clip]);
    const float l = db_term(power[(int64_t)clip * kMelBands * n_frames + i]) - ref;
    out[((int64_t)clip * kMelBands + band) * n_frames + c] = l < -80.0f ? -80.0f : l;
}

int mel_frames(int64_t n) { return n <= 0 ? 0 : (int)((n + kMelHalf + kMelHop - 1) / kMelHop); }

// workspace for n_clips clips: block sums, positions, counts are separate small buffers (see api.hip);
// this is the size of the split spectra and the band powers
size_t mel_work_bytes(int64_t n, int n_clips)
{
    const int nf = mel_frames(n), fp = (nf + 1) / 2 * 2;
    return ((size_t)n_clips * fp * 2 * kMelHpad + (size_t)n_clips * kMelBands * nf) * sizeof(float);
}

void launch_mel(const RowsArgs &rows, const float *d_win, const float *d_cpack, const int16_t *d_pcm, int64_t n, int n_clips,
                int64_t *d_blk, int *d_pos, int *d_count, float *d_pmax, float *d_work, float *d_out, hipStream_t s)
{
    const int nf = mel_frames(n), fp = (nf + 1) / 2 * 2, n_blk = (int)((n + kMelHop - 1) / kMelHop);
    float *y = d_work, *power = d_work + (size_t)n_clips * fp * 2 * kMelHpad;
    hipLaunchKernelGGL(mel_blocksum_kernel, dim3((n_blk + 3) / 4, n_clips), dim3(256), 0, s, d_pcm, n, n_blk, d_blk);
    hipLaunchKernelGGL(mel_keep_kernel, dim3(n_clips), dim3(256), 0, s, d_blk, n_blk, nf, d_pos, d_count);
    static PerDeviceOnce attr_set;
    if (attr_set.need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(stft_rows_kernel<Groups4410>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set.mark();
    }
    hipLaunchKernelGGL(stft_rows_kernel<Groups4410>, dim3(fp / 2, n_clips), dim3(kMelThreads), (size_t)kMelFrame * sizeof(cf), s,
                       rows, d_pcm, n, nf, fp, d_win, y);
    hipLaunchKernelGGL(mel_bands_kernel, dim3((nf + 127) / 128, n_clips), dim3(256), 0, s, y, d_cpack, nf, fp, power);
    hipLaunchKernelGGL(mel_max_kernel, dim3(n_clips), dim3(256), 0, s, power, d_pos, nf, d_pmax);
    hipLaunchKernelGGL(mel_db_kernel, dim3((unsigned)(((int64_t)kMelBands * nf + 255) / 256), n_clips), dim3(256), 0, s, power,
                       d_pos, d_pmax, nf, d_out);
}

} // namespace hpfw
