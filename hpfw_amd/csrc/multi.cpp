// multi.cpp -- libhpfw_gpu_multi.so: the multi-GPU host path of include/hpfw_gpu_multi.h.
//
// One process, one hpfw_gpu handle per shard, shards placed on the devices of one node.  What
// LiveSongIdentification::index / search do around MemoryStorage (reference live_song_id.h:31-54,
// storage.h:21-64) becomes: build = contiguous blocks of clips per shard; find = replicated queries, one scan
// per shard (each on its own device and stream, enqueued by its own host thread), ONE ncclAllGather of the
// per-shard top-k lists over xGMI, one deterministic merge.  No torch, no Python: librccl and libamdhip64 only.
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include "../../include/hpfw_gpu_multi.h"
#include "legacy_internal.h"

extern "C" void hpfw_internal_set_error(const char *msg); // libhpfw_gpu.so: feeds hpfw_gpu_last_error()

namespace {

int fail(int code, const std::string &msg)
{
    hpfw_internal_set_error(msg.c_str());
    return code;
}

struct Shard {
    int dev_slot = 0;  // index into hpfw_gpu_group::devs
    int local = 0;     // position among the shards of that device
    hpfw_gpu *h = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    uint64_t *d_q = nullptr;
    size_t q_cap = 0;
    int64_t lo = 0, hi = 0; // global clip ids of its block
};

struct Dev {
    int device = 0;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    hpfw_hit *d_send = nullptr, *d_recv = nullptr;
    size_t send_cap = 0, recv_cap = 0;
    std::vector<int> shards;
};

} // namespace

struct hpfw_gpu_group {
    std::vector<hpfw_legacy_collector *> collectors; // one per shard, created by the first load / prepare
    std::string cache;
    std::vector<Shard> shards;
    std::vector<Dev> devs;
    int per_dev = 1; // shards on every device (uniform)
    int64_t n_clips = 0;
    std::string exchange;
};

namespace {

#define HIP_OK(expr, what)                                                                                         \
    do {                                                                                                           \
        hipError_t e_ = (expr);                                                                                    \
        if (e_ != hipSuccess) return fail(HPFW_E_HIP, std::string(what) + ": " + hipGetErrorString(e_));          \
    } while (0)
#define NCCL_OK(expr, what)                                                                                        \
    do {                                                                                                           \
        ncclResult_t r_ = (expr);                                                                                  \
        if (r_ != ncclSuccess) return fail(HPFW_E_HIP, std::string(what) + ": " + ncclGetErrorString(r_));        \
    } while (0)

int grow(void **p, size_t *cap, size_t need)
{
    if (*cap >= need) return 0;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
    if (hipMalloc(p, need) != hipSuccess) return fail(HPFW_E_NOMEM, "hipMalloc failed");
    *cap = need;
    return 0;
}

// fn(shard index) on one host thread per shard; the first failure's code and message reach the caller's thread
template <class F>
int per_shard(hpfw_gpu_group *g, F fn)
{
    const int n = (int)g->shards.size();
    std::vector<int> rc((size_t)n, 0);
    std::vector<std::string> why((size_t)n);
    auto run = [&](int i) {
        (void)hipSetDevice(g->devs[(size_t)g->shards[(size_t)i].dev_slot].device);
        rc[(size_t)i] = fn(i);
        if (rc[(size_t)i]) why[(size_t)i] = hpfw_gpu_last_error(); // thread-local in libhpfw_gpu.so
    };
    if (n == 1) {
        run(0);
    } else {
        std::vector<std::thread> th;
        for (int i = 1; i < n; ++i) th.emplace_back(run, i);
        run(0);
        for (auto &t : th) t.join();
    }
    for (int i = 0; i < n; ++i)
        if (rc[(size_t)i]) return fail(rc[(size_t)i], "shard " + std::to_string(i) + ": " + why[(size_t)i]);
    return 0;
}

} // namespace

extern "C" {

void hpfw_gpu_shard_range(int64_t n_clips, int shard, int n_shards, int64_t *lo, int64_t *hi)
{
    const int64_t base = n_clips / n_shards, extra = n_clips % n_shards;
    const int64_t l = shard * base + std::min<int64_t>(shard, extra);
    if (lo) *lo = l;
    if (hi) *hi = l + base + (shard < extra ? 1 : 0);
}

int hpfw_gpu_group_create(const int *devices, int n_shards, hpfw_gpu_group **out)
{
    if (!out || n_shards < 1 || n_shards > 64) return fail(HPFW_E_INVALID, "bad argument");
    *out = nullptr;
    int n_dev = 0;
    HIP_OK(hipGetDeviceCount(&n_dev), "hipGetDeviceCount");
    auto *g = new hpfw_gpu_group();
    g->shards.resize((size_t)n_shards);
    for (int i = 0; i < n_shards; ++i) {
        const int d = devices ? devices[i] : i;
        if (d < 0 || d >= n_dev) {
            hpfw_gpu_group_destroy(g);
            return fail(HPFW_E_INVALID, "shard " + std::to_string(i) + ": no device " + std::to_string(d));
        }
        size_t slot = 0;
        while (slot < g->devs.size() && g->devs[slot].device != d) ++slot;
        if (slot == g->devs.size()) {
            g->devs.emplace_back();
            g->devs.back().device = d;
        }
        g->shards[(size_t)i].dev_slot = (int)slot;
        g->shards[(size_t)i].local = (int)g->devs[slot].shards.size();
        g->devs[slot].shards.push_back(i);
    }
    g->per_dev = (int)g->devs[0].shards.size();
    for (const Dev &d : g->devs)
        if ((int)d.shards.size() != g->per_dev) {
            hpfw_gpu_group_destroy(g);
            return fail(HPFW_E_INVALID, "every device must hold the same number of shards");
        }
    for (Shard &s : g->shards) {
        const int d = g->devs[(size_t)s.dev_slot].device;
        if (hpfw_gpu_create(d, &s.h) != 0) {
            hpfw_gpu_group_destroy(g);
            return HPFW_E_HIP; // message set by hpfw_gpu_create
        }
        if (hipSetDevice(d) != hipSuccess || hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&s.done, hipEventDisableTiming) != hipSuccess) {
            hpfw_gpu_group_destroy(g);
            return fail(HPFW_E_HIP, "stream/event creation failed");
        }
    }
    // one communicator per distinct device, all in this process (ncclCommInitAll); world size 1 is allowed
    std::vector<int> devlist;
    for (Dev &d : g->devs) devlist.push_back(d.device);
    std::vector<ncclComm_t> comms(devlist.size(), nullptr);
    ncclResult_t r = ncclCommInitAll(comms.data(), (int)devlist.size(), devlist.data());
    if (r != ncclSuccess) {
        hpfw_gpu_group_destroy(g);
        return fail(HPFW_E_HIP, std::string("ncclCommInitAll: ") + ncclGetErrorString(r));
    }
    for (size_t i = 0; i < g->devs.size(); ++i) {
        g->devs[i].comm = comms[i];
        if (hipSetDevice(g->devs[i].device) != hipSuccess ||
            hipStreamCreateWithFlags(&g->devs[i].stream, hipStreamNonBlocking) != hipSuccess) {
            hpfw_gpu_group_destroy(g);
            return fail(HPFW_E_HIP, "stream creation failed");
        }
    }
    g->exchange = g->per_dev == 1 ? "rccl" : "rccl+local";
    *out = g;
    return 0;
}

int hpfw_gpu_group_create_env(hpfw_gpu_group **out)
{
    std::vector<int> devs;
    if (const char *e = std::getenv("HPFW_GPU_DEVICES")) {
        for (const char *p = e; *p;) {
            char *end = nullptr;
            const long v = std::strtol(p, &end, 10);
            if (end == p) return fail(HPFW_E_INVALID, std::string("HPFW_GPU_DEVICES: cannot parse '") + e + "'");
            devs.push_back((int)v);
            p = *end == ',' ? end + 1 : end;
            if (*end && *end != ',') return fail(HPFW_E_INVALID, std::string("HPFW_GPU_DEVICES: cannot parse '") + e + "'");
        }
    }
    if (devs.empty()) {
        int n = 0;
        HIP_OK(hipGetDeviceCount(&n), "hipGetDeviceCount");
        for (int i = 0; i < n; ++i) devs.push_back(i);
    }
    return hpfw_gpu_group_create(devs.data(), (int)devs.size(), out);
}

void hpfw_gpu_group_destroy(hpfw_gpu_group *g)
{
    if (!g) return;
    for (Dev &d : g->devs) {
        (void)hipSetDevice(d.device);
        if (d.stream) (void)hipStreamSynchronize(d.stream);
        if (d.comm) (void)ncclCommDestroy(d.comm);
        if (d.d_send) (void)hipFree(d.d_send);
        if (d.d_recv) (void)hipFree(d.d_recv);
        if (d.stream) (void)hipStreamDestroy(d.stream);
    }
    for (hpfw_legacy_collector *c : g->collectors) par_collector_del(c);
    for (Shard &s : g->shards) {
        if (s.dev_slot < (int)g->devs.size()) (void)hipSetDevice(g->devs[(size_t)s.dev_slot].device);
        if (s.stream) (void)hipStreamSynchronize(s.stream);
        if (s.d_q) (void)hipFree(s.d_q);
        if (s.done) (void)hipEventDestroy(s.done);
        if (s.stream) (void)hipStreamDestroy(s.stream);
        if (s.h) hpfw_gpu_destroy(s.h);
    }
    delete g;
}

int hpfw_gpu_group_size(const hpfw_gpu_group *g) { return g ? (int)g->shards.size() : 0; }

hpfw_gpu *hpfw_gpu_group_handle(hpfw_gpu_group *g, int shard)
{
    return g && shard >= 0 && shard < (int)g->shards.size() ? g->shards[(size_t)shard].h : nullptr;
}

const char *hpfw_gpu_group_exchange(const hpfw_gpu_group *g) { return g ? g->exchange.c_str() : ""; }

int hpfw_gpu_group_set_filters(hpfw_gpu_group *g, const float *f)
{
    if (!g || !f) return fail(HPFW_E_INVALID, "null argument");
    for (Shard &s : g->shards) {
        const int rc = hpfw_gpu_set_filters(s.h, f);
        if (rc) return rc;
    }
    return 0;
}

int hpfw_gpu_group_extract_pcm16(hpfw_gpu_group *g, const int16_t *pcm, int64_t n_samples, int64_t n_clips, uint64_t *hp)
{
    if (!g || !pcm || !hp || n_clips < 0) return fail(HPFW_E_INVALID, "bad argument");
    hpfw_geometry geo;
    int rc = hpfw_gpu_geometry(g->shards[0].h, n_samples, &geo);
    if (rc) return rc;
    const int n = (int)g->shards.size();
    return per_shard(g, [&](int i) {
        int64_t lo, hi;
        hpfw_gpu_shard_range(n_clips, i, n, &lo, &hi);
        if (hi == lo) return 0;
        return hpfw_gpu_extract_pcm16_host(g->shards[(size_t)i].h, pcm + lo * n_samples, n_samples, hi - lo, hp + lo * geo.n_hp);
    });
}

int hpfw_gpu_group_index_build(hpfw_gpu_group *g, const uint64_t *hp, const int64_t *offsets, int64_t n_clips)
{
    if (!g || !offsets || n_clips < 0 || (n_clips > 0 && !hp)) return fail(HPFW_E_INVALID, "bad argument");
    if (n_clips > 0xfffffff0ll) return fail(HPFW_E_INVALID, "too many clips");
    const int n = (int)g->shards.size();
    const int rc = per_shard(g, [&](int i) {
        Shard &s = g->shards[(size_t)i];
        hpfw_gpu_shard_range(n_clips, i, n, &s.lo, &s.hi);
        int r = hpfw_gpu_index_clear(s.h);
        if (!r) r = hpfw_gpu_index_set_clip_base(s.h, (uint32_t)s.lo);
        if (!r && s.hi > s.lo) r = hpfw_gpu_index_add(s.h, hp, offsets + s.lo, s.hi - s.lo);
        return r;
    });
    if (!rc) g->n_clips = n_clips;
    return rc;
}

int64_t hpfw_gpu_group_index_size(const hpfw_gpu_group *g) { return g ? g->n_clips : 0; }

int hpfw_gpu_group_search_topk(hpfw_gpu_group *g, const uint64_t *q_hp, const int64_t *q_off, int64_t n_q, int k, hpfw_hit *out)
{
    if (!g || !q_off || !out || n_q < 0) return fail(HPFW_E_INVALID, "bad argument");
    if (k < 1 || k > 64) return fail(HPFW_E_INVALID, "k must be in 1..64");
    if (n_q == 0) return 0;
    const int64_t total = q_off[n_q] - q_off[0];
    if (total > 0 && !q_hp) return fail(HPFW_E_INVALID, "null queries");
    std::vector<int64_t> rel((size_t)n_q + 1);
    for (int64_t i = 0; i <= n_q; ++i) rel[(size_t)i] = q_off[i] - q_off[0];
    const size_t list = (size_t)n_q * k; // hits per shard
    const int n = (int)g->shards.size(), m = (int)g->devs.size();
    for (Dev &d : g->devs) {
        HIP_OK(hipSetDevice(d.device), "hipSetDevice");
        int rc = grow((void **)&d.d_send, &d.send_cap, list * g->per_dev * sizeof(hpfw_hit));
        if (!rc) rc = grow((void **)&d.d_recv, &d.recv_cap, list * n * sizeof(hpfw_hit));
        if (rc) return rc;
    }
    // 1. every shard: replicated queries in, scan of its block, its own top-k (global clip ids) into its slot of
    //    the device's send buffer
    int rc = per_shard(g, [&](int i) {
        Shard &s = g->shards[(size_t)i];
        Dev &d = g->devs[(size_t)s.dev_slot];
        int r = grow((void **)&s.d_q, &s.q_cap, (size_t)std::max<int64_t>(total, 1) * 8);
        if (r) return r;
        if (total && hipMemcpyAsync(s.d_q, q_hp + q_off[0], (size_t)total * 8, hipMemcpyHostToDevice, s.stream) != hipSuccess)
            return fail(HPFW_E_HIP, "H2D copy of the queries failed");
        r = hpfw_gpu_search_topk_device(s.h, s.d_q, rel.data(), n_q, k, d.d_send + (size_t)s.local * list, s.stream);
        if (r) return r;
        if (hipEventRecord(s.done, s.stream) != hipSuccess) return fail(HPFW_E_HIP, "event record failed");
        return 0;
    });
    if (rc) return rc;
    // 2. the exchange step: one all-gather of per_dev x n_q x k x 16 bytes per device over xGMI
    for (Dev &d : g->devs) {
        HIP_OK(hipSetDevice(d.device), "hipSetDevice");
        for (int si : d.shards) HIP_OK(hipStreamWaitEvent(d.stream, g->shards[(size_t)si].done, 0), "hipStreamWaitEvent");
    }
    NCCL_OK(ncclGroupStart(), "ncclGroupStart");
    for (Dev &d : g->devs) {
        ncclResult_t r = ncclAllGather(d.d_send, d.d_recv, list * g->per_dev * sizeof(hpfw_hit), ncclUint8, d.comm, d.stream);
        if (r != ncclSuccess) {
            (void)ncclGroupEnd();
            return fail(HPFW_E_HIP, std::string("ncclAllGather: ") + ncclGetErrorString(r));
        }
    }
    NCCL_OK(ncclGroupEnd(), "ncclGroupEnd");
    // 3. every device now holds all n lists; the host takes device 0's copy and merges by (dist, clip)
    std::vector<hpfw_hit> all(list * n);
    HIP_OK(hipSetDevice(g->devs[0].device), "hipSetDevice");
    HIP_OK(hipMemcpyAsync(all.data(), g->devs[0].d_recv, all.size() * sizeof(hpfw_hit), hipMemcpyDeviceToHost, g->devs[0].stream),
           "D2H copy of the gathered lists");
    for (int i = 0; i < m; ++i) {
        HIP_OK(hipSetDevice(g->devs[(size_t)i].device), "hipSetDevice");
        HIP_OK(hipStreamSynchronize(g->devs[(size_t)i].stream), "all-gather");
    }
    return hpfw_gpu_merge_topk(all.data(), n, n_q, k, out);
}

int hpfw_gpu_group_cov_reset(hpfw_gpu_group *g)
{
    if (!g) return fail(HPFW_E_INVALID, "null group");
    for (Shard &s : g->shards) {
        const int rc = hpfw_gpu_cov_reset(s.h);
        if (rc) return rc;
    }
    return 0;
}

int hpfw_gpu_group_cov_accumulate_pcm16(hpfw_gpu_group *g, const int16_t *pcm, int64_t n_samples, int64_t n_clips)
{
    if (!g || !pcm || n_clips < 0) return fail(HPFW_E_INVALID, "bad argument");
    const int n = (int)g->shards.size();
    return per_shard(g, [&](int i) {
        int64_t lo, hi;
        hpfw_gpu_shard_range(n_clips, i, n, &lo, &hi);
        if (hi == lo) return 0;
        return hpfw_gpu_cov_accumulate_pcm16_host(g->shards[(size_t)i].h, pcm + lo * n_samples, n_samples, hi - lo);
    });
}

// accum_cov summed over the given handles (one per shard, shard order): afterwards every handle holds the total
// and the total file count.  One in-place ncclAllReduce of 23.4 MB when every shard has its own device.
static int sum_covariances(hpfw_gpu_group *g, const std::vector<hpfw_gpu *> &hs, int64_t *files_out)
{
    const size_t nn = (size_t)HPFW_FRAME_SIZE * HPFW_FRAME_SIZE;
    int64_t files = 0;
    for (hpfw_gpu *h : hs) files += hpfw_gpu_cov_files(h);
    *files_out = files;
    if (files == 0) return fail(HPFW_E_INVALID, "no covariance accumulated");
    int rc;
    if (g->per_dev == 1) {
        // accum_cov is a plain sum over files (parallel_collector.h:93-97)
        std::vector<float *> d_cov(hs.size(), nullptr);
        for (size_t i = 0; i < hs.size(); ++i)
            if ((rc = hpfw_gpu_cov_device(hs[i], &d_cov[i]))) return rc;
        for (Dev &d : g->devs) {
            HIP_OK(hipSetDevice(d.device), "hipSetDevice");
            HIP_OK(hipDeviceSynchronize(), "covariance kernels"); // the accumulation ran on the handles' own streams
        }
        NCCL_OK(ncclGroupStart(), "ncclGroupStart");
        for (Dev &d : g->devs) {
            float *p = d_cov[(size_t)d.shards[0]];
            ncclResult_t r = ncclAllReduce(p, p, nn, ncclFloat, ncclSum, d.comm, d.stream);
            if (r != ncclSuccess) {
                (void)ncclGroupEnd();
                return fail(HPFW_E_HIP, std::string("ncclAllReduce: ") + ncclGetErrorString(r));
            }
        }
        NCCL_OK(ncclGroupEnd(), "ncclGroupEnd");
        for (Dev &d : g->devs) {
            HIP_OK(hipSetDevice(d.device), "hipSetDevice");
            HIP_OK(hipStreamSynchronize(d.stream), "all-reduce");
        }
    } else {
        // shards that share a device: their matrices are summed on the host (RCCL has one rank per device)
        std::vector<float> sum(nn, 0.0f), one(nn);
        for (hpfw_gpu *h : hs) {
            if ((rc = hpfw_gpu_cov_get(h, one.data(), nullptr))) return rc;
            for (size_t i = 0; i < nn; ++i) sum[i] += one[i];
        }
        for (hpfw_gpu *h : hs)
            if ((rc = hpfw_gpu_cov_set(h, sum.data(), files))) return rc;
    }
    for (hpfw_gpu *h : hs)
        if ((rc = hpfw_gpu_cov_set_files(h, files))) return rc;
    return 0;
}

int hpfw_gpu_group_learn_filters(hpfw_gpu_group *g, float *filters_out)
{
    if (!g) return fail(HPFW_E_INVALID, "null group");
    std::vector<hpfw_gpu *> hs;
    for (Shard &s : g->shards) hs.push_back(s.h);
    int64_t files = 0;
    int rc = sum_covariances(g, hs, &files);
    if (rc) return rc;
    std::vector<float> f((size_t)HPFW_FILTERS * HPFW_FRAME_SIZE);
    HIP_OK(hipSetDevice(g->devs[(size_t)g->shards[0].dev_slot].device), "hipSetDevice");
    if ((rc = hpfw_gpu_learn_filters(g->shards[0].h, f.data()))) return rc;
    for (size_t i = 1; i < g->shards.size(); ++i)
        if ((rc = hpfw_gpu_set_filters(g->shards[i].h, f.data()))) return rc;
    // the reference keeps accumulating across calls (parallel_collector.h:93-97): the total stays on shard 0 only, so
    // that an accumulate + learn that follows adds every earlier file once and not once per shard
    for (size_t i = 1; i < g->shards.size(); ++i)
        if ((rc = hpfw_gpu_cov_reset(g->shards[i].h))) return rc;
    if (filters_out) std::memcpy(filters_out, f.data(), f.size() * 4);
    return 0;
}

// ---- ParallelCollector over the shards ----------------------------------------------------------------------
static int ensure_collectors(hpfw_gpu_group *g, const char *cache)
{
    if (cache && *cache) g->cache = cache;
    if (!g->collectors.empty()) return 0;
    for (size_t i = 0; i < g->shards.size(); ++i) {
        const int dev = g->devs[(size_t)g->shards[i].dev_slot].device;
        hpfw_legacy_collector *c = hpfw_internal_collector_on_device(dev, g->cache.c_str());
        if (!c) {
            for (hpfw_legacy_collector *d : g->collectors) par_collector_del(d);
            g->collectors.clear();
            return HPFW_E_HIP; // message set by hpfw_gpu_create
        }
        // accum_cov.cereal carries the covariance of earlier runs (the reference keeps accumulating, cache.h:34-36,
        // live_song_id.h:23-29): it enters the sum once, through shard 0
        if (i > 0) (void)hpfw_gpu_cov_reset(hpfw_internal_collector_gpu(c));
        g->collectors.push_back(c);
    }
    return 0;
}

int hpfw_gpu_group_load(hpfw_gpu_group *g, const char *cache)
{
    if (!g) return fail(HPFW_E_INVALID, "null group");
    if (!g->collectors.empty()) { // load again: every shard re-reads the cache
        if (cache && *cache) g->cache = cache;
        for (size_t i = 0; i < g->collectors.size(); ++i) {
            par_collector_load(g->collectors[i], g->cache.c_str());
            if (i > 0) (void)hpfw_gpu_cov_reset(hpfw_internal_collector_gpu(g->collectors[i]));
        }
        return 0;
    }
    return ensure_collectors(g, cache);
}

int hpfw_gpu_group_save(hpfw_gpu_group *g, const char *cache)
{
    if (!g) return fail(HPFW_E_INVALID, "null group");
    if (cache && *cache) g->cache = cache;
    if (g->collectors.empty()) return 0; // nothing loaded, nothing learned
    par_collector_save(g->collectors[0], g->cache.c_str());
    return 0;
}

FilenameHashprintPair *hpfw_gpu_group_prepare(hpfw_gpu_group *g, const char **filenames, int n, int *got)
{
    if (got) *got = 0;
    if (!g || !filenames || n < 0 || !got) {
        (void)fail(HPFW_E_INVALID, "bad argument");
        return nullptr;
    }
    if (ensure_collectors(g, nullptr)) return nullptr;
    const int ns = (int)g->shards.size();
    const bool learn = !std::getenv("HPFW_PREPARE_KEEP_FILTERS");
    std::vector<hpfw_prepare_job *> jobs((size_t)ns, nullptr);
    std::vector<int64_t> lo((size_t)ns), hi((size_t)ns);
    for (int i = 0; i < ns; ++i) hpfw_gpu_shard_range(n, i, ns, &lo[(size_t)i], &hi[(size_t)i]);
    // 1. preprocess (parallel_collector.h:82-105) on every shard's block of files
    int rc = per_shard(g, [&](int i) {
        jobs[(size_t)i] = hpfw_internal_prepare_accumulate(g->collectors[(size_t)i], filenames + lo[(size_t)i],
                                                           (int)(hi[(size_t)i] - lo[(size_t)i]), learn ? 1 : 0);
        return jobs[(size_t)i] ? 0 : (int)HPFW_E_NOMEM;
    });
    // 2. one all-reduce of accum_cov, the eigen-solve on shard 0 (:111), the filters to every shard, the cache saved (:61-66)
    bool ok = rc == 0;
    if (ok && learn) {
        std::vector<hpfw_gpu *> hs;
        for (hpfw_legacy_collector *c : g->collectors) hs.push_back(hpfw_internal_collector_gpu(c));
        int64_t files = 0, used = 0;
        for (hpfw_prepare_job *j : jobs) used += hpfw_internal_prepare_used(j);
        std::vector<float> f((size_t)HPFW_FILTERS * HPFW_FRAME_SIZE);
        ok = used > 0 && sum_covariances(g, hs, &files) == 0 && hipSetDevice(hpfw_gpu_device(hs[0])) == hipSuccess &&
             hpfw_gpu_learn_filters(hs[0], f.data()) == 0;
        for (size_t i = 0; ok && i < g->collectors.size(); ++i) ok = hpfw_internal_collector_set_filters(g->collectors[i], f.data()) == 0;
        // the total stays on shard 0 only, so that the next prepare() adds every file once
        for (size_t i = 1; i < hs.size(); ++i) (void)hpfw_gpu_cov_reset(hs[i]);
        if (ok) par_collector_save(g->collectors[0], g->cache.c_str());
        if (used == 0) ok = true; // nothing readable: an empty result, as the single collector returns
    }
    // 3. collect_fingerprints (:115-137): every shard hashes its own files; shard 0 adds the older tracks of the cache
    std::vector<FilenameHashprintPair *> part((size_t)ns, nullptr);
    std::vector<int> part_n((size_t)ns, 0);
    (void)per_shard(g, [&](int i) {
        if (jobs[(size_t)i])
            part[(size_t)i] = hpfw_internal_prepare_finish(g->collectors[(size_t)i], jobs[(size_t)i], filenames + lo[(size_t)i],
                                                           (int)(hi[(size_t)i] - lo[(size_t)i]), ok ? 1 : 0, 0, &part_n[(size_t)i]);
        return 0;
    });
    FilenameHashprintPair *cached = nullptr;
    int cached_n = 0;
    if (ok && !std::getenv("HPFW_NO_SPECTRO_CACHE")) {
        // older tracks: an accumulate over zero files followed by a finish that walks the cache, told about all n names
        hpfw_prepare_job *walk = hpfw_internal_prepare_accumulate(g->collectors[0], filenames, 0, 0);
        if (walk) cached = hpfw_internal_prepare_finish_cached(g->collectors[0], walk, filenames, n, &cached_n);
    }
    if (!ok) {
        for (int i = 0; i < ns; ++i)
            if (part[(size_t)i]) prepare_result_free(part[(size_t)i], part_n[(size_t)i]);
        (void)fail(HPFW_E_INVALID, "prepare: the filters could not be learned");
        return nullptr;
    }
    int total = cached_n;
    for (int i = 0; i < ns; ++i) total += part_n[(size_t)i];
    auto *res = new FilenameHashprintPair[(size_t)std::max(total, 1)];
    int w = 0;
    auto take = [&](FilenameHashprintPair *p, int cnt) {
        for (int k = 0; k < cnt; ++k) res[w++] = p[k]; // ownership of the strings and hashprints moves
        delete[] p;
    };
    for (int i = 0; i < ns; ++i)
        if (part[(size_t)i]) take(part[(size_t)i], part_n[(size_t)i]);
    if (cached) take(cached, cached_n);
    *got = w;
    return res;
}

uint64_t *hpfw_gpu_group_calc_hashprint(hpfw_gpu_group *g, const char *filename, int *size)
{
    if (size) *size = 0;
    if (!g || !filename || !size || ensure_collectors(g, nullptr)) return nullptr;
    (void)hipSetDevice(g->devs[(size_t)g->shards[0].dev_slot].device);
    return par_collector_calc_hashprint(g->collectors[0], filename, size);
}

} // extern "C"
