// k_search.hip -- a9: exhaustive sliding Hamming scan + top-k.
//
// Replaces MemoryStorage::find (reference include/hpfw/audioproblems/live-song-id/storage.h:27-64):
// for every indexed clip and every offset, sum_j popcount(q[j] ^ r[off + j]); keep the first strict
// minimum per clip; then the k clips with the smallest (distance, clip id) -- k = 1 is find()'s
// global first strict minimum (storage.h:56-60), k = 10 the notebook's top-10
// (examples/python/liveid.ipynb cell 9).  Pure integer work: v_xor_b32 x2 + v_bcnt_u32_b32 x2
// per 64-bit pair, VALU-issue bound.
//
// hamming_scan_kernel: workgroup = (clip, tile of 8 queries).  Lane = offset; the clip's
// hashprints slide through LDS in chunks of 256 offsets; each LDS word fetched is compared with
// the 8 queries of the tile, whose words are wave-uniform (scalar loads).
#include "kernels.h"

namespace hpfw {

extern __shared__ __align__(16) unsigned char smem_raw[];

constexpr int kHsThreads = 256;

__device__ __forceinline__ uint64_t wave_min_u64(uint64_t v)
{
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        const uint64_t o = (uint64_t)__shfl_xor((unsigned long long)v, s);
        v = o < v ? o : v;
    }
    return v;
}

// kHsQt = queries per workgroup (8, or fewer when there are fewer queries: a single streaming query
// would otherwise pay for seven empty slots)
template <int kHsQt>
__global__ __launch_bounds__(kHsThreads) void hamming_scan_kernel(SearchArgs a)
{
    uint64_t *r_lds = reinterpret_cast<uint64_t *>(smem_raw);                  // [256 + k_max]
    uint64_t *red = r_lds + (kHsThreads + a.k_max);                            // [4][kHsQt]
    const int tid = threadIdx.x;
    const int clip = blockIdx.x;
    const int q0 = blockIdx.y * kHsQt;
    const int64_t r0 = a.db_off[clip];
    const int n = (int)(a.db_off[clip + 1] - r0);
    if (n <= 0) return;

    int kq[kHsQt];
    const uint64_t *qp[kHsQt];
    int kt = 0, kmin_eff = n;
    bool uniform_k = true;
#pragma unroll
    for (int t = 0; t < kHsQt; ++t) {
        const int qi = q0 + t;
        int k = 0;
        const uint64_t *p = a.q;
        if (qi < a.n_q) {
            const int64_t o = a.q_off[qi];
            k = (int)(a.q_off[qi + 1] - o);
            if (k > n) k = n; // storage.h:37-39: k = min(k, n)
            p = a.q + o;
        }
        kq[t] = k;
        qp[t] = p;
        kt = k > kt ? k : kt;
        if (k > 0 && k < kmin_eff) kmin_eff = k;
    }
#pragma unroll
    for (int t = 0; t < kHsQt; ++t) uniform_k = uniform_k && (kq[t] == kt);
    if (kt == 0) return;
    const int n_off = n - kmin_eff + 1; // offsets any query of the tile may use

    uint64_t best[kHsQt];
#pragma unroll
    for (int t = 0; t < kHsQt; ++t) best[t] = ~0ull;

    for (int o0 = 0; o0 < n_off; o0 += kHsThreads) {
        __syncthreads();
        for (int i = tid; i < kHsThreads + kt - 1; i += kHsThreads) {
            const int g = o0 + i;
            r_lds[i] = g < n ? a.db[r0 + g] : 0ull;
        }
        __syncthreads();
        uint32_t acc[kHsQt];
#pragma unroll
        for (int t = 0; t < kHsQt; ++t) acc[t] = 0;
        if (uniform_k) {
            for (int j = 0; j < kt; ++j) {
                const uint64_t rv = r_lds[tid + j];
#pragma unroll
                for (int t = 0; t < kHsQt; ++t) acc[t] += (uint32_t)__popcll(rv ^ qp[t][j]);
            }
        } else {
            for (int j = 0; j < kt; ++j) {
                const uint64_t rv = r_lds[tid + j];
#pragma unroll
                for (int t = 0; t < kHsQt; ++t)
                    if (j < kq[t]) acc[t] += (uint32_t)__popcll(rv ^ qp[t][j]);
            }
        }
        const int off = o0 + tid;
#pragma unroll
        for (int t = 0; t < kHsQt; ++t) {
            if (kq[t] > 0 && off <= n - kq[t]) {
                const uint64_t key = ((uint64_t)acc[t] << 32) | (uint32_t)off;
                best[t] = key < best[t] ? key : best[t]; // smaller distance, then smaller offset
            }
        }
    }
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int t = 0; t < kHsQt; ++t) {
        const uint64_t v = wave_min_u64(best[t]);
        if (lane == 0) red[wave * kHsQt + t] = v;
    }
    __syncthreads();
    if (tid < kHsQt) {
        uint64_t v = red[tid];
        for (int w = 1; w < kHsThreads / 64; ++w) {
            const uint64_t o = red[w * kHsQt + tid];
            v = o < v ? o : v;
        }
        const int qi = q0 + tid;
        if (qi < a.n_q) a.best[(int64_t)qi * a.n_clips + clip] = v;
    }
}

struct HitDev {
    uint32_t dist, clip;
    int32_t offset;
    uint32_t pad;
};

// one workgroup per query: k rounds of "smallest (dist, clip) key above the previous one"
__global__ __launch_bounds__(256) void topk_kernel(const uint64_t *__restrict__ best, int n_clips, int k,
                                                   uint32_t clip_base, HitDev *__restrict__ out)
{
    __shared__ uint64_t red[4];
    const int tid = threadIdx.x;
    const int q = blockIdx.x;
    const uint64_t *row = best + (int64_t)q * n_clips;
    uint64_t prev = 0;
    bool first = true;
    for (int t = 0; t < k; ++t) {
        uint64_t mine = ~0ull;
        for (int c = tid; c < n_clips; c += 256) {
            const uint64_t b = row[c];
            if (b == ~0ull) continue;
            const uint64_t key = (b & 0xffffffff00000000ull) | (uint32_t)c;
            if ((first || key > prev) && key < mine) mine = key;
        }
        mine = wave_min_u64(mine);
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = mine;
        __syncthreads();
        uint64_t sel = red[0];
        for (int w = 1; w < 4; ++w) sel = red[w] < sel ? red[w] : sel;
        if (tid == 0) {
            HitDev h;
            if (sel == ~0ull) {
                h.dist = 0xffffffffu;
                h.clip = 0xffffffffu;
                h.offset = 0;
            } else {
                const uint32_t c = (uint32_t)sel;
                h.dist = (uint32_t)(sel >> 32);
                h.clip = clip_base + c;
                h.offset = (int32_t)(uint32_t)row[c];
            }
            h.pad = 0;
            out[(int64_t)q * k + t] = h;
        }
        if (sel != ~0ull) {
            prev = sel;
            first = false;
        } else {
            prev = ~0ull; // nothing left: every later round also selects nothing
            first = false;
        }
    }
}

// Large indexes: top-k in two steps so that one query is not the work of one workgroup.  Step 1: slice
// g of query q (clips [g S, (g + 1) S)) -> its k best as (key = dist << 32 | clip, offset), ~0 where none;
// step 2: topk_merge_kernel picks the k smallest keys of the G k candidates.
constexpr int kTkSlices = 64;

__global__ __launch_bounds__(256) void topk_slice_kernel(const uint64_t *__restrict__ best, int n_clips, int k,
                                                         uint64_t *__restrict__ cand_key, int32_t *__restrict__ cand_off)
{
    __shared__ uint64_t red[4];
    const int tid = threadIdx.x, g = blockIdx.x, q = blockIdx.y;
    const int per = (n_clips + kTkSlices - 1) / kTkSlices, c0 = g * per, c1 = min(n_clips, c0 + per);
    const uint64_t *row = best + (int64_t)q * n_clips;
    uint64_t prev = 0;
    bool first = true;
    for (int t = 0; t < k; ++t) {
        uint64_t mine = ~0ull;
        for (int c = c0 + tid; c < c1; c += 256) {
            const uint64_t b = row[c];
            if (b == ~0ull) continue;
            const uint64_t key = (b & 0xffffffff00000000ull) | (uint32_t)c;
            if ((first || key > prev) && key < mine) mine = key;
        }
        mine = wave_min_u64(mine);
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = mine;
        __syncthreads();
        uint64_t sel = red[0];
        for (int w = 1; w < 4; ++w) sel = red[w] < sel ? red[w] : sel;
        if (tid == 0) {
            const int64_t o = ((int64_t)q * kTkSlices + g) * k + t;
            cand_key[o] = sel;
            cand_off[o] = sel == ~0ull ? 0 : (int32_t)(uint32_t)row[(uint32_t)sel];
        }
        prev = sel;
        first = false;
    }
}

__global__ __launch_bounds__(256) void topk_merge_kernel(const uint64_t *__restrict__ cand_key,
                                                         const int32_t *__restrict__ cand_off, int k, uint32_t clip_base,
                                                         HitDev *__restrict__ out)
{
    __shared__ uint64_t red[4];
    __shared__ int red_i[4];
    const int tid = threadIdx.x, q = blockIdx.x;
    const int n = kTkSlices * k;
    const uint64_t *keys = cand_key + (int64_t)q * n;
    uint64_t prev = 0;
    bool first = true;
    for (int t = 0; t < k; ++t) {
        uint64_t mine = ~0ull;
        int mi = 0;
        for (int i = tid; i < n; i += 256) {
            const uint64_t key = keys[i];
            if (key != ~0ull && (first || key > prev) && key < mine) {
                mine = key;
                mi = i;
            }
        }
        // the keys are distinct (they carry the clip id), so the winner's index follows its key
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) {
            const uint64_t o = (uint64_t)__shfl_xor((unsigned long long)mine, s);
            const int oi = __shfl_xor(mi, s);
            if (o < mine) {
                mine = o;
                mi = oi;
            }
        }
        __syncthreads();
        if ((tid & 63) == 0) {
            red[tid >> 6] = mine;
            red_i[tid >> 6] = mi;
        }
        __syncthreads();
        uint64_t sel = red[0];
        int si = red_i[0];
        for (int w = 1; w < 4; ++w)
            if (red[w] < sel) {
                sel = red[w];
                si = red_i[w];
            }
        if (tid == 0) {
            HitDev h;
            if (sel == ~0ull) {
                h.dist = 0xffffffffu;
                h.clip = 0xffffffffu;
                h.offset = 0;
            } else {
                h.dist = (uint32_t)(sel >> 32);
                h.clip = clip_base + (uint32_t)sel;
                h.offset = cand_off[(int64_t)q * n + si];
            }
            h.pad = 0;
            out[(int64_t)q * k + t] = h;
        }
        prev = sel;
        first = false;
    }
}

size_t topk_scratch_bytes(int n_q, int k) { return (size_t)n_q * kTkSlices * k * (sizeof(uint64_t) + sizeof(int32_t)); }

void launch_topk_two_step(const uint64_t *d_best, int n_q, int n_clips, int k, uint32_t clip_base, void *d_scratch,
                          void *d_out, hipStream_t s)
{
    uint64_t *keys = reinterpret_cast<uint64_t *>(d_scratch);
    int32_t *offs = reinterpret_cast<int32_t *>(keys + (size_t)n_q * kTkSlices * k);
    hipLaunchKernelGGL(topk_slice_kernel, dim3(kTkSlices, n_q), dim3(256), 0, s, d_best, n_clips, k, keys, offs);
    hipLaunchKernelGGL(topk_merge_kernel, dim3(n_q), dim3(256), 0, s, keys, offs, k, clip_base,
                       reinterpret_cast<HitDev *>(d_out));
}

template <int QT>
static void launch_hamming_scan_t(const SearchArgs &a, hipStream_t s)
{
    static PerDeviceOnce attr_set;
    if (attr_set.need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(hamming_scan_kernel<QT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set.mark();
    }
    const size_t lds = ((size_t)kHsThreads + a.k_max + 4 * QT) * sizeof(uint64_t);
    dim3 grid(a.n_clips, (a.n_q + QT - 1) / QT);
    hipLaunchKernelGGL(hamming_scan_kernel<QT>, grid, dim3(kHsThreads), lds, s, a);
}

void launch_hamming_scan(const SearchArgs &a, hipStream_t s)
{
    if (a.n_q == 1)
        launch_hamming_scan_t<1>(a, s);
    else if (a.n_q == 2)
        launch_hamming_scan_t<2>(a, s);
    else if (a.n_q <= 4)
        launch_hamming_scan_t<4>(a, s);
    else
        launch_hamming_scan_t<8>(a, s);
}

void launch_topk(const uint64_t *d_best, int n_q, int n_clips, int k, uint32_t clip_base, void *d_out,
                 hipStream_t s)
{
    hipLaunchKernelGGL(topk_kernel, dim3(n_q), dim3(256), 0, s, d_best, n_clips, k, clip_base,
                       reinterpret_cast<HitDev *>(d_out));
}

} // namespace hpfw
