// k_hashprint_cfg.hip -- HashprintHandle<N, SpectrogramHandler, FramesContext, T> (reference
// include/hpfw/core/hashprint_handle.h:50-64) for template arguments other than the live-id default
// <uint64_t, CQT<>, 20, 80> that k_hashprint.hip is specialised for -- first of all the combiner's
// HashPrint<uint16_t, MelSpectrogram<>, 32, 50> (include/hpfw/audioproblems/combiner/combiner.h:12):
// 33 spectrogram rows, 32 context columns (frames of 1056 values), 16 filters, lag 50, 16-bit hashprints.
//
//   calc_frames (:79-93)  X[row * context + t, n] = S[row, n + t]       never materialised (implicit im2col)
//   filters * frames      P[r, n] = fma chain over k = row * context + t ascending           f32 MFMA
//   calc_fingerprint (:115-125) + bool_col_to_num (:137-142)  bit (bits - 1 - r) = (P[r,i] - P[r,i+T] >= 0)
//
// Spectrograms may have a different number of valid columns per clip (the Mel front end drops silent frames,
// mel.h:94-96): cols[clip] of the `stride` columns of every row are valid.
// One wave = 32 frames x RT row tiles of 32 filter rows of v_mfma_f32_32x32x2_f32 (16 filters leave half a tile
// idle: 0.1 GFLOP per 30 s clip is not worth a second instruction shape with its own accumulation order to pin).
#include "kernels.h"

namespace hpfw {

extern __shared__ __align__(16) unsigned char smem_raw[];

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kCfgTileN = 128; // frames per workgroup: 4 waves x 32

template <int RT>
__global__ __launch_bounds__(256) void project_cfg_kernel(CfgArgs a, const float *__restrict__ s, const int *__restrict__ cols,
                                                          int64_t stride, float *__restrict__ proj, int64_t proj_stride)
{
    float *slab = reinterpret_cast<float *>(smem_raw); // [rows][slab_w]
    const int clip = blockIdx.y, n0 = blockIdx.x * kCfgTileN;
    const int c = cols ? cols[clip] : (int)stride;      // valid columns of this clip
    const int nf = c - a.context + 1;
    if (n0 >= nf) return;
    const int slab_w = kCfgTileN + a.context - 1;
    const float *S = s + (int64_t)clip * a.rows * stride;
    for (int i = threadIdx.x; i < a.rows * slab_w; i += 256) {
        const int row = i / slab_w, col = i - row * slab_w;
        slab[i] = (n0 + col < c) ? S[(int64_t)row * stride + n0 + col] : 0.0f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, part = lane >> 5;
    const int col0 = wave * 32 + j; // this lane's frame within the tile
    f32x16 acc[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) acc[t] = f32x16{0};
    // k = row * context + t; this lane supplies B[k = 2 kp + part][frame]
    int row = 0, t = part;
    while (t >= a.context) { // context == 1
        t -= a.context;
        ++row;
    }
    const float *__restrict__ fp = a.fpack + lane;
    const int ksteps = (a.rows * a.context + 1) / 2;
    for (int kp = 0; kp < ksteps; ++kp) {
        const float b = row < a.rows ? slab[row * slab_w + col0 + t] : 0.0f;
#pragma unroll
        for (int tile = 0; tile < RT; ++tile)
            acc[tile] = __builtin_amdgcn_mfma_f32_32x32x2f32(fp[((int64_t)kp * RT + tile) * 64], b, acc[tile], 0, 0, 0);
        t += 2;
        while (t >= a.context) {
            t -= a.context;
            ++row;
        }
    }
    // D layout: column = lane & 31; register reg of a tile = filter row (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    const int n = n0 + col0;
    if (n >= nf) return;
    float *P = proj + (int64_t)clip * a.filters * proj_stride + n;
#pragma unroll
    for (int tile = 0; tile < RT; ++tile)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int r = tile * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * part;
            if (r < a.filters) P[(int64_t)r * proj_stride] = acc[tile][reg];
        }
}

// thread = one hashprint; WORD = uint16_t / uint32_t / uint64_t
template <class WORD>
__global__ __launch_bounds__(256) void pack_cfg_kernel(CfgArgs a, const float *__restrict__ proj, const int *__restrict__ cols,
                                                       int64_t stride, int64_t proj_stride, WORD *__restrict__ hp,
                                                       int64_t hp_stride)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int clip = blockIdx.y;
    const int c = cols ? cols[clip] : (int)stride;
    const int nhp = c - a.context + 1 - a.lag;
    if (i >= nhp) return;
    const float *P = proj + (int64_t)clip * a.filters * proj_stride + i;
    WORD v = 0;
    for (int r = 0; r < a.filters; ++r) {
        const float d = P[(int64_t)r * proj_stride] - P[(int64_t)r * proj_stride + a.lag];
        v |= (WORD)((WORD)(d >= 0.0f) << (a.filters - 1 - r));
    }
    hp[(int64_t)clip * hp_stride + i] = v;
}

// operand image of v_mfma_f32_32x32x2_f32: [k-pair][row tile][lane], lane l supplies A[row = 32 tile + (l & 31)]
// [k = 2 kp + (l >> 5)]; zero beyond the filters and beyond the frame
void pack_cfg_filters(int rows, int context, int filters, const float *f_colmajor, float *fpack)
{
    const int k_total = rows * context, ksteps = (k_total + 1) / 2, rt = (filters + 31) / 32;
    for (int kp = 0; kp < ksteps; ++kp)
        for (int tile = 0; tile < rt; ++tile)
            for (int l = 0; l < 64; ++l) {
                const int r = tile * 32 + (l & 31), k = 2 * kp + (l >> 5);
                fpack[((size_t)kp * rt + tile) * 64 + l] = (r < filters && k < k_total) ? f_colmajor[(size_t)r + (size_t)filters * k] : 0.0f;
            }
}

size_t cfg_fpack_floats(int rows, int context, int filters)
{
    return (size_t)((rows * context + 1) / 2) * ((filters + 31) / 32) * 64;
}

size_t project_cfg_lds_bytes(const CfgArgs &a) { return (size_t)a.rows * (kCfgTileN + a.context - 1) * sizeof(float); }

void launch_project_cfg(const CfgArgs &a, const float *d_s, const int *d_cols, int n_clips, int64_t stride, float *d_proj,
                        int64_t proj_stride, hipStream_t s)
{
    const int nf_max = (int)stride - a.context + 1;
    if (nf_max <= 0 || n_clips <= 0) return;
    dim3 grid((nf_max + kCfgTileN - 1) / kCfgTileN, n_clips);
    const size_t lds = project_cfg_lds_bytes(a);
    static PerDeviceOnce attr_set;
    if (attr_set.need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(project_cfg_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(project_cfg_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set.mark();
    }
    if (a.filters <= 32)
        hipLaunchKernelGGL(project_cfg_kernel<1>, grid, dim3(256), lds, s, a, d_s, d_cols, stride, d_proj, proj_stride);
    else
        hipLaunchKernelGGL(project_cfg_kernel<2>, grid, dim3(256), lds, s, a, d_s, d_cols, stride, d_proj, proj_stride);
}

void launch_pack_cfg(const CfgArgs &a, const float *d_proj, const int *d_cols, int n_clips, int64_t stride, int64_t proj_stride,
                     void *d_hp, int64_t hp_stride, hipStream_t s)
{
    const int nhp_max = (int)stride - a.context + 1 - a.lag;
    if (nhp_max <= 0 || n_clips <= 0) return;
    dim3 grid((nhp_max + 255) / 256, n_clips);
    if (a.filters == 16)
        hipLaunchKernelGGL(pack_cfg_kernel<uint16_t>, grid, dim3(256), 0, s, a, d_proj, d_cols, stride, proj_stride, (uint16_t *)d_hp, hp_stride);
    else if (a.filters == 32)
        hipLaunchKernelGGL(pack_cfg_kernel<uint32_t>, grid, dim3(256), 0, s, a, d_proj, d_cols, stride, proj_stride, (uint32_t *)d_hp, hp_stride);
    else
        hipLaunchKernelGGL(pack_cfg_kernel<uint64_t>, grid, dim3(256), 0, s, a, d_proj, d_cols, stride, proj_stride, (uint64_t *)d_hp, hp_stride);
}

} // namespace hpfw
