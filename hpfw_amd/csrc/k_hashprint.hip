// k_hashprint.hip -- a5..a8: dB spectrogram -> projection -> delta -> 64-bit hashprints.
//
// Replaces HashprintHandle::calc_frames + `filters * frames` (reference
// include/hpfw/core/hashprint_handle.h:79-93, include/hpfw/core/parallel_collector.h:57,127: an
// Eigen/MKL sgemm [64 x 2420] . [2420 x n_frames]) and calc_fingerprint / fingerprint_to_hashprint
// (hashprint_handle.h:115-142).
//
// project_kernel: implicit im2col.  frames[b*20 + t, n] = S[b, n + t] is never materialised; the
// workgroup keeps an [11 bins][256 + 19 columns] slab of S in LDS, streams the filter operand from
// L2 into registers and feeds v_mfma_f32_32x32x2_f32.  The accumulation over k = b*20 + t is one
// chain per output in ascending k (the MFMA adds k, k+1 in order and chains across instructions),
// which is bit for bit the fmaf chain of DESIGN.md S9.
#include "kernels.h"

namespace hpfw {

extern __shared__ __align__(16) unsigned char smem_raw[];

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kPjThreads = 256;            // 4 waves; each wave owns 64 filters x 64 frames
constexpr int kPjTileN = 256;              // frames per workgroup
constexpr int kPjBinsPerChunk = 11;        // 121 = 11 * 11
constexpr int kPjCols = kPjTileN + kCtx - 1;   // 275 spectrogram columns per slab row
constexpr int kPjRow = 276;                // LDS row stride in floats
constexpr int kPjTp = kCtx / 2;            // 10 MFMA k-steps (k, k+1) per bin

// LDS holds only the [11 bins][275 columns] slab of S (12 KB, so several workgroups share a CU and
// one's staging hides behind another's MFMAs).  The filter operand is streamed straight from L2 into
// registers, one bin (10 k-steps, 80 B per lane) ahead of its use: fpack is laid out
// [bin][lane][k-step][filter tile] so that a wave reads 5 KB contiguously per bin.
// FROM_T: sdb holds the dB terms t written by the chirp-z kernel; the conversion
// S = max(t - tmax[clip], -80) (convert.h:12-15) rides on the staging loads.
template <bool FROM_T>
__global__ __launch_bounds__(kPjThreads, 3) void project_kernel(const float *__restrict__ fpack,
                                                             const float *__restrict__ sdb,
                                                             const float *__restrict__ tmax, int c, int nf,
                                                             float *__restrict__ proj)
{
    __shared__ float s_tiles[2][kPjBinsPerChunk * kPjRow]; // two slabs: chunk i + 1 is written while chunk i is read
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int kh = lane >> 5;
    const int clip = blockIdx.y;
    const int n0 = blockIdx.x * kPjTileN;
    const float *S = sdb + (int64_t)clip * kBins * c;
    const float ref = FROM_T ? tmax[clip] : 0.0f;
    f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0}; // [filter tile][frame tile]
    const int nl = wave * 64 + (lane & 31);
    const float4 *ap = reinterpret_cast<const float4 *>(fpack) + lane * (kPjTp / 2);
    float4 a_cur[kPjTp / 2], a_nxt[kPjTp / 2];
#pragma unroll
    for (int i = 0; i < kPjTp / 2; ++i) a_cur[i] = ap[i];

    // S slab staging split in two (load to registers early, write to LDS late): the loads of chunk
    // i + 1 are in flight while chunk i feeds the MFMAs, so HBM latency never stalls the matrix pipe.
    // Thread t takes column t of each of the 11 bins (row base in scalar registers, t in one vector
    // register: no per-element address arithmetic to keep alive across the chunk) and threads 0..208 one
    // element of the 19-column tail.
    constexpr int kTail = kPjCols - kPjTileN;             // 19
    constexpr int kStage = kPjBinsPerChunk + 1;           // 12 words per thread
    float stage[kStage];
    const bool main_in = n0 + tid < c;
    const int tb = tid / kTail, tc = kPjTileN + (tid - tb * kTail); // the tail element of this thread
    const bool tail_on = tid < kPjBinsPerChunk * kTail, tail_in = tail_on && n0 + tc < c;
    auto stage_load = [&](int chunk) {
        const float *Sc = S + (int64_t)chunk * kPjBinsPerChunk * c + n0;
#pragma unroll
        for (int bb = 0; bb < kPjBinsPerChunk; ++bb) stage[bb] = main_in ? Sc[(int64_t)bb * c + tid] : 0.0f;
        stage[kPjBinsPerChunk] = tail_in ? Sc[(int64_t)tb * c + tc] : 0.0f;
    };
    auto stage_store = [&](float *s_tile) {
        // the dB conversion waits until here so that the loads stay in flight behind the MFMAs
        // (columns past the clip get a meaningless value: only frames >= nf, never stored, see them)
#pragma unroll
        for (int j = 0; j < kStage; ++j) {
            float v = stage[j];
            if (FROM_T) {
                const float l = v - ref;
                v = l < -80.0f ? -80.0f : l;
            }
            if (j < kPjBinsPerChunk)
                s_tile[j * kPjRow + tid] = v;
            else if (tail_on)
                s_tile[tb * kPjRow + tc] = v;
        }
    };
    stage_load(0);
    stage_store(s_tiles[0]);
    constexpr int kChunks = kBins / kPjBinsPerChunk;
    if (kChunks > 1) stage_load(1);
    __syncthreads();
    for (int chunk = 0; chunk < kChunks; ++chunk) {
        const float *s_tile = s_tiles[chunk & 1];
        const bool active = n0 + wave * 64 < nf; // a wave whose 64 frames lie past the clip only helps staging
        if (active) {
#pragma unroll 1
            for (int b = 0; b < kPjBinsPerChunk; ++b) {
                const int bin = chunk * kPjBinsPerChunk + b;
                const int nbin = bin + 1 < kBins ? bin + 1 : bin; // the last prefetch re-reads the last bin
                // next bin's filter operand first, pinned here: left alone the scheduler sinks these loads
                // to the end of the bin and the wave then waits out their L2 latency before every copy
#pragma unroll
                for (int i = 0; i < kPjTp / 2; ++i) a_nxt[i] = ap[(size_t)nbin * 64 * (kPjTp / 2) + i];
                const float *srow = s_tile + b * kPjRow + nl + kh;
                float b0 = srow[0], b1 = srow[32];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int tp = 0; tp < kPjTp; ++tp) {
                    const float4 a4 = a_cur[tp >> 1];
                    const float a0 = (tp & 1) ? a4.z : a4.x; // filter tile 0
                    const float a1 = (tp & 1) ? a4.w : a4.y; // filter tile 1
                    const float c0 = b0, c1 = b1;
                    if (tp + 1 < kPjTp) { // the S operands of the next step are read while this one multiplies
                        b0 = srow[2 * tp + 2];
                        b1 = srow[2 * tp + 34];
                    }
                    acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, c0, acc00, 0, 0, 0);
                    acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, c1, acc01, 0, 0, 0);
                    acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, c0, acc10, 0, 0, 0);
                    acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, c1, acc11, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int i = 0; i < kPjTp / 2; ++i) a_cur[i] = a_nxt[i];
            }
        }
        // chunk + 1 goes into the other slab (last read during chunk - 1, which every wave left before the
        // barrier below ran for chunk - 1), then chunk + 2 starts loading: one barrier per chunk
        if (chunk + 1 < kChunks) stage_store(s_tiles[(chunk + 1) & 1]);
        if (chunk + 2 < kChunks) stage_load(chunk + 2);
        __syncthreads();
    }
    // D layout of the 32x32 tile: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    float *P = proj + (int64_t)clip * kFilters * nf;
    const int n = n0 + nl;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * kh;
        if (n < nf) {
            P[(int64_t)row * nf + n] = acc00[reg];
            P[(int64_t)(row + 32) * nf + n] = acc10[reg];
        }
        if (n + 32 < nf) {
            P[(int64_t)row * nf + n + 32] = acc01[reg];
            P[(int64_t)(row + 32) * nf + n + 32] = acc11[reg];
        }
    }
}

// bit (63 - r) of hp[i] = (P[r,i] - P[r,i+80] >= 0)
__global__ __launch_bounds__(256) void pack_kernel(const float *__restrict__ proj, int nf, int nhp,
                                                   uint64_t *__restrict__ hp)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int clip = blockIdx.y;
    if (i >= nhp) return;
    const float *P = proj + (int64_t)clip * kFilters * nf + i;
    uint64_t v = 0;
#pragma unroll 16
    for (int r = 0; r < kFilters; ++r) {
        const float d = P[(int64_t)r * nf] - P[(int64_t)r * nf + kLag];
        v |= (uint64_t)(d >= 0.0f) << (63 - r);
    }
    hp[(int64_t)clip * nhp + i] = v;
}

void pack_filters_for_mfma(const float *f, float *fpack)
{
    // operand image of v_mfma_f32_32x32x2_f32: lane l supplies A[row = l & 31][k = l >> 5];
    // order [bin][lane][k-step][filter tile], k = bin * 20 + 2 * step + (l >> 5)
    for (int bin = 0; bin < kBins; ++bin)
        for (int l = 0; l < 64; ++l)
            for (int tp = 0; tp < kPjTp; ++tp)
                for (int tile = 0; tile < 2; ++tile) {
                    const int r = tile * 32 + (l & 31);
                    const int k = bin * kCtx + 2 * tp + (l >> 5);
                    fpack[(((size_t)bin * 64 + l) * kPjTp + tp) * 2 + tile] = f[(size_t)r + 64 * (size_t)k];
                }
}

void launch_project(const float *d_fpack, const float *d_db, const float *d_tmax, int n_clips, int c, float *d_proj,
                    hipStream_t s)
{
    const int nf = c - (kCtx - 1);
    dim3 grid((nf + kPjTileN - 1) / kPjTileN, n_clips);
    if (d_tmax)
        hipLaunchKernelGGL(project_kernel<true>, grid, dim3(kPjThreads), 0, s, d_fpack, d_db, d_tmax, c, nf, d_proj);
    else
        hipLaunchKernelGGL(project_kernel<false>, grid, dim3(kPjThreads), 0, s, d_fpack, d_db, d_tmax, c, nf, d_proj);
}

void launch_pack(const float *d_proj, int n_clips, int nf, uint64_t *d_hp, hipStream_t s)
{
    const int nhp = nf - kLag;
    dim3 grid((nhp + 255) / 256, n_clips);
    hipLaunchKernelGGL(pack_kernel, grid, dim3(256), 0, s, d_proj, nf, nhp, d_hp);
}

} // namespace hpfw
