// api.hip -- implementation of the C-ABI declared in include/hpfw_gpu.h.
// Host orchestration only: plans, workspaces, batching over clips (the role of
// ParallelCollector::collect_fingerprints' parallel_for, reference
// include/hpfw/core/parallel_collector.h:115-137) and MemoryStorage::build/find
// (include/hpfw/audioproblems/live-song-id/storage.h:21-64).  All arithmetic is in the kernels.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <chrono>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "../../include/hpfw_gpu.h"
#include "kernels.h"
#include "plan.h"

struct hpfw_gpu;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return fail(HPFW_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));                 \
    } while (0)

thread_local size_t g_uploaded = 0; // bytes uploaded by upload() since get_plan last reset it

template <class T>
int upload(const std::vector<T> &v, const T **out, std::vector<void *> &owned)
{
    void *d = nullptr;
    g_uploaded += v.size() * sizeof(T);
    if (v.empty()) {
        *out = nullptr;
        return 0;
    }
    HIP_TRY(hipMalloc(&d, v.size() * sizeof(T)));
    owned.push_back(d);
    HIP_TRY(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = reinterpret_cast<const T *>(d);
    return 0;
}

hpfw::RadixList to_radix(const std::vector<int> &r)
{
    hpfw::RadixList rl;
    std::memset(&rl, 0, sizeof(rl));
    rl.n = (int)r.size();
    for (size_t i = 0; i < r.size(); ++i) rl.r[i] = r[i];
    return rl;
}

// Device memory of the per-length tables.  A plan takes its memory from the handle's pool: small tables are carved out of
// 4 MB chunks, large ones are blocks of their own, and an evicted plan's chunks and blocks go back to the pool for the next
// length (a corpus of tracks brings a new length with every file: ~40 hipMalloc and, at eviction, as many hipFree per plan
// cost more than generating the tables).
void *pool_take(hpfw_gpu *h, size_t bytes);
void pool_give(hpfw_gpu *h, void *p, size_t bytes);
void pool_release(hpfw_gpu *h);

struct DevPlan {
    hpfw_gpu *owner = nullptr;
    std::vector<std::pair<void *, size_t>> blocks; // what this plan holds of the pool
    char *cur = nullptr;                           // the open chunk
    size_t left = 0;
    // small tables are written to a host image of the open chunk and go over in one copy per run of them (plan_flush):
    // some thirty-five synchronous copies of a few KB each cost more than the tables' bytes
    char *chunk_base = nullptr;
    std::vector<char> stage;
    std::vector<std::pair<size_t, size_t>> staged; // (offset, bytes) written to the image, in order
    hpfw::HostPlan hp;
    hpfw::ColsQArgs cols;
    hpfw::Rows2Out rows_out;
    hpfw::RowsArgs rows;
    hpfw::BzArgs bz; // clip lengths with a prime factor above 7 (hp.bluestein)
    hpfw::CqPlanDev cq;
    std::vector<hpfw::CqClassDev> cls;
    size_t bytes = 0;      // device memory of the tables
    uint64_t last_use = 0; // for the least-recently-used eviction in get_plan
    ~DevPlan()
    {
        for (auto &b : blocks) pool_give(owner, b.first, b.second);
    }
};

constexpr size_t kPlanChunk = (size_t)4 << 20;

// HPFW_PLAN_TIMING=1: where the first use of a clip length spends the host's time (printed when the handle goes)
struct PlanTiming {
    double host_wait = 0, host_build = 0, upload = 0, device_tables = 0, alloc = 0, evict = 0, total = 0;
    long plans = 0, copies = 0;
    size_t copied = 0;
};
thread_local PlanTiming *g_plan_timing = nullptr;
inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
struct PlanTimer {
    double *acc, t0;
    explicit PlanTimer(double PlanTiming::*m) : acc(g_plan_timing ? &(g_plan_timing->*m) : nullptr), t0(acc ? now_s() : 0.0) {}
    ~PlanTimer()
    {
        if (acc) *acc += now_s() - t0;
    }
};
hipError_t plan_h2d(hpfw_gpu *h, void *dst, const void *src, size_t bytes); // (below: needs the handle)

int plan_flush(DevPlan *dp)
{
    size_t i = 0;
    while (i < dp->staged.size()) {
        size_t off = dp->staged[i].first, end = off + dp->staged[i].second;
        // (a run = tables in adjacent 256-byte slots; a table the device fills itself, in between, ends the run)
        for (++i; i < dp->staged.size() && dp->staged[i].first == (end + 255) / 256 * 256; ++i) end = dp->staged[i].first + dp->staged[i].second;
        HIP_TRY(plan_h2d(dp->owner, dp->chunk_base + off, dp->stage.data() + off, end - off));
    }
    dp->staged.clear();
    return 0;
}

int plan_alloc(DevPlan *dp, size_t bytes, void **out)
{
    PlanTimer t(&PlanTiming::alloc);
    bytes = (bytes + 255) / 256 * 256;
    if (bytes >= kPlanChunk / 4) { // a block of its own, in 64 KB steps (equal sizes recur: lengths near each other share n1 and n2)
        const size_t size = (bytes + 65535) / 65536 * 65536;
        void *p = pool_take(dp->owner, size);
        if (!p) return fail(HPFW_E_HIP, "out of device memory for the tables of a clip length");
        dp->blocks.emplace_back(p, size);
        *out = p;
        return 0;
    }
    if (dp->left < bytes) {
        int rc = plan_flush(dp);
        if (rc) return rc;
        void *p = pool_take(dp->owner, kPlanChunk);
        if (!p) return fail(HPFW_E_HIP, "out of device memory for the tables of a clip length");
        dp->blocks.emplace_back(p, kPlanChunk);
        dp->cur = dp->chunk_base = static_cast<char *>(p);
        dp->left = kPlanChunk;
        dp->stage.resize(kPlanChunk);
    }
    *out = dp->cur;
    dp->cur += bytes;
    dp->left -= bytes;
    return 0;
}

template <class T>
int upload(const std::vector<T> &v, const T **out, DevPlan *dp)
{
    g_uploaded += v.size() * sizeof(T);
    if (v.empty()) {
        *out = nullptr;
        return 0;
    }
    void *d = nullptr;
    const size_t bytes = v.size() * sizeof(T);
    int rc = plan_alloc(dp, bytes, &d);
    if (rc) return rc;
    const bool in_chunk = dp->chunk_base && static_cast<char *>(d) >= dp->chunk_base && static_cast<char *>(d) < dp->chunk_base + kPlanChunk &&
                          bytes < kPlanChunk / 4;
    if (in_chunk) {
        const size_t off = (size_t)(static_cast<char *>(d) - dp->chunk_base);
        std::memcpy(dp->stage.data() + off, v.data(), bytes);
        dp->staged.emplace_back(off, bytes);
    } else {
        HIP_TRY(plan_h2d(dp->owner, d, v.data(), bytes));
    }
    *out = reinterpret_cast<const T *>(d);
    return 0;
}

enum KernelKind { K_ROWS = 0, K_COLS, K_CQ, K_DB, K_PROJECT, K_PACK, K_SCAN, K_TOPK, K_PAIRS, K_FWD, K_COUNT };
const char *const kKernelNames[K_COUNT] = {"fwd_rows", "fwd_cols", "cq_chirpz", "db",
                                           "project_mfma", "delta_pack", "hamming_scan", "topk", "pcm_pairs", "fwd_span"};

struct TimedLaunch {
    int kind;
    hipEvent_t a, b;
};

} // namespace

struct hpfw_gpu {
    int device = 0;
    bool has_filters = false;
    float *d_fpack = nullptr;
    void *d_fq_image = nullptr; // the filters' fixed-point digits (k_project_q.hip)
    int projection = 1;         // 1: fixed point (S9q), 0: the f32 fma chain (S9); hpfw_gpu_set_projection
    std::map<int64_t, std::unique_ptr<DevPlan>> plans; // one per clip length, least recently used evicted
    // host halves of plans prepared ahead by other threads (hpfw_gpu_prepare_length): a null entry is being built
    std::mutex host_mtx;
    std::condition_variable host_cv;
    std::map<int64_t, std::unique_ptr<hpfw::HostPlan>> host_ready;
    std::set<int64_t> host_seen; // lengths being prepared, prepared, or resident on the device (not prepared again while they are)
    size_t plan_bytes = 0;
    uint64_t plan_clock = 0;
    // plans last used at or before this value of plan_clock are known to be idle: the caller has waited for all the work it
    // queued on this handle since (hpfw_internal_note_idle: the file collectors, once per window of files) -- such plans
    // are evicted without the device-wide wait that an eviction otherwise needs
    uint64_t idle_clock = 0;
    std::multimap<size_t, void *> dev_pool; // free chunks and blocks of evicted plans, by size
    size_t dev_pool_bytes = 0;
    unsigned conventions = 0; // hpfw_gpu_set_conventions: essentia conventions that cannot be checked offline
    // clips per pass: 2.5 GB of workspace at 30 s; every launch but the forward transform's chunks fills the 256 CUs many
    // times over, and what one stage leaves for the next (forward bins, dB terms: 2.3 MB per clip) is still in the caches
    // when it is read (1000 clips: 10.25 ms in one pass, 10.0 in four)
    int batch = 256;
    // extraction workspace
    size_t ws_bytes[7] = {0, 0, 0, 0, 0, 0, 0};
    // yp, x, mag, proj, wave maxima [clip][121][16], pairs, second planar buffer of the chirp-z forward transform
    void *ws[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    // index
    uint64_t *d_db = nullptr;
    size_t db_cap = 0;
    std::vector<int64_t> db_off{0};
    int64_t *d_db_off = nullptr;
    size_t db_off_cap = 0;
    bool db_off_dirty = true;
    uint32_t clip_base = 0;
    // search scratch
    uint64_t *d_best = nullptr;
    size_t best_cap = 0;
    int64_t *d_q_off = nullptr;
    size_t q_off_cap = 0;
    // filter learning: accum_cov of ParallelCollector (parallel_collector.h:76), upper tiles only
    float *d_cov = nullptr;
    float *d_cov_ws = nullptr; // scratch of the covariance kernels
    void *d_cqwork = nullptr;  // chirp-z bands too long for the LDS (k_cq_big.hip)
    // the size classes of the chirp-z stage run side by side (run_front): their workgroups differ in LDS footprint and
    // one class alone leaves part of every CU's LDS and issue slots unused
    static constexpr int kCqSide = 4;
    hipStream_t cq_side[kCqSide] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t cq_fork = nullptr, cq_join[kCqSide] = {nullptr, nullptr, nullptr, nullptr};
    int cq_concurrent = 1; // HPFW_CQ_SERIAL=1 in the environment at creation: one class after the other on the caller's stream
    // the forward transform of a large batch in chunks of fwd_chunk clips taken in turn by fwd_streams streams (the
    // caller's and side streams): column stage and row stage of a chunk back to back, so that the column stage's output
    // (5.3 MB per clip) is read back out of the Infinity Cache instead of HBM.  HPFW_FWD_CHUNK (0: one launch per stage
    // for the whole batch), HPFW_FWD_STREAMS (1..5) in the environment at creation
    int fwd_chunk = 16, fwd_streams = 2;
    int cols_variant = 0; // HPFW_COLS_VARIANT (tests, diagnosis): kernels.h ColsQArgs::variant
    std::unique_ptr<PlanTiming> plan_timing; // HPFW_PLAN_TIMING
    // The tables of a new clip length go to the device on a stream of the handle's own, from a pinned ring, without a host
    // wait (the stream that first uses them waits for plan_ev): synchronous copies on the default stream waited behind
    // whatever shared its hardware queue -- after bench.py's host-buffer section that was the collector's extraction stream,
    // and a corpus of distinct lengths lost a fifth of its rate (tools/ffi_interaction.sh)
    hipStream_t plan_stream = nullptr;
    char *pin_ring = nullptr;
    size_t pin_cap = 0, pin_off = 0;
    // HPFW_BACK_OVERLAP=1: the back end (hashprints from dB terms: the int8 matrix pipe) of one pass beside the front end
    // (transforms: vector ALU, LDS) of the next, on a stream of its own.  Off by default: the kernels slow each other by
    // what the overlap would gain (9.95-10.03 against 9.70-9.92 ms per 1000 clips on one box)
    int back_overlap = 0;
    hipStream_t back_side = nullptr;
    hipEvent_t back_fork = nullptr, back_join = nullptr;
    int bz_chunk = 32;  // the same for the chirp-z forward transform's three kernels (HPFW_BZ_CHUNK; 38.6 -> 39.6 k clips/s at 30 s)
    void *d_topk_scratch = nullptr;
    size_t topk_scratch_cap = 0;
    // Mel front-end: tables (owned by mel_owned), workspaces
    bool mel_ready = false;
    hpfw::HostPlan mel_plan;
    hpfw::RowsArgs mel_rows;
    const float *d_mel_win = nullptr, *d_mel_cpack = nullptr;
    std::vector<void *> mel_owned;
    void *d_mel_work = nullptr, *d_mel_small = nullptr;
    size_t mel_work_cap = 0, mel_small_cap = 0;
    size_t cqwork_cap = 0;
    size_t cov_ws_cap = 0;
    void *d_qa = nullptr;   // queries expanded to fp4 for the matrix-core scan
    size_t qa_cap = 0;
    int *d_gk = nullptr;    // longest query of each group of 32
    size_t gk_cap = 0;
    // staging of the host-buffer entry points: kept between calls (a one-file call is otherwise mostly
    // allocation and stream set-up)
    void *stage_pcm[2] = {nullptr, nullptr};
    size_t stage_pcm_cap[2] = {0, 0};
    void *stage_hp = nullptr;
    size_t stage_hp_cap = 0;
    hipStream_t stage_copy = nullptr, stage_comp = nullptr;
    hipEvent_t stage_copied[2] = {nullptr, nullptr}, stage_consumed[2] = {nullptr, nullptr};
    float *d_clipmax = nullptr; // per-clip maximum magnitude (reference level of the dB conversion)
    size_t clipmax_cap = 0;
    int *d_cov_tiles = nullptr;
    int64_t cov_files = 0;
    // HashprintHandle configurations other than the default (hpfw_gpu_cfg_*): filter operand images by config
    std::map<std::vector<int>, float *> cfg_fpack;
    float *d_cfg_proj = nullptr;
    size_t cfg_proj_cap = 0;
    struct CfgCov {
        float *d_accum = nullptr;
        int *d_tiles = nullptr;
        int64_t clips = 0;
    };
    std::map<std::vector<int>, CfgCov> cfg_cov; // accum_cov of other configurations, by (rows, context)
    float *d_cfg_cov_ws = nullptr;
    size_t cfg_cov_ws_cap = 0;
    // ordering of consecutive entry points that were handed different streams (the workspaces are shared)
    hipEvent_t order_ev = nullptr;
    hipStream_t order_stream = nullptr;
    bool order_valid = false;
    // the tables a new length generates on the device (default stream) are awaited by the stream that first uses them
    // (Ordered), not by the host: a caller on a stream of its own keeps preparing lengths while earlier files run
    hipEvent_t plan_ev = nullptr;
    bool plan_ev_pending = false;
    // timing
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    unsigned timing_mask = 0;
    std::vector<TimedLaunch> timed;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    float k_ms[K_COUNT] = {0};
    int k_launches[K_COUNT] = {0};
};

namespace {

constexpr size_t kPinRing = (size_t)48 << 20;

hipError_t plan_h2d(hpfw_gpu *h, void *dst, const void *src, size_t bytes)
{
    PlanTimer t(&PlanTiming::upload);
    if (g_plan_timing) {
        ++g_plan_timing->copies;
        g_plan_timing->copied += bytes;
    }
    if (!h->pin_ring) {
        if (hipHostMalloc(reinterpret_cast<void **>(&h->pin_ring), kPinRing, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            h->pin_ring = nullptr;
        } else {
            h->pin_cap = kPinRing;
        }
    }
    if (!h->pin_ring || bytes > h->pin_cap / 2) { // (a table beyond the ring: the runtime stages it; rare -- clips of many minutes)
        hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, h->plan_stream);
        return e != hipSuccess ? e : hipStreamSynchronize(h->plan_stream);
    }
    if (h->pin_off + bytes > h->pin_cap) { // wrap: what was copied out of the ring before has to be gone
        hipError_t e = hipStreamSynchronize(h->plan_stream);
        if (e != hipSuccess) return e;
        h->pin_off = 0;
    }
    std::memcpy(h->pin_ring + h->pin_off, src, bytes);
    hipError_t e = hipMemcpyAsync(dst, h->pin_ring + h->pin_off, bytes, hipMemcpyHostToDevice, h->plan_stream);
    h->pin_off += (bytes + 255) / 256 * 256;
    return e;
}

int ensure(void **p, size_t *cap, size_t need, hpfw_gpu *pool_owner = nullptr)
{
    if (*cap >= need) return 0;
    if (*p) HIP_TRY(hipFree(*p));
    *p = nullptr;
    *cap = 0;
    if (hipMalloc(p, need) != hipSuccess && pool_owner) { // the handle's pool of table blocks may be holding what is missing
        (void)hipGetLastError();
        pool_release(pool_owner);
    } else if (*p) {
        *cap = need;
        return 0;
    }
    HIP_TRY(hipMalloc(p, need));
    *cap = need;
    return 0;
}


// The pool: blocks of evicted plans and the temporaries of table generation, kept (up to 4 GiB, on top of the plan cache's
// HPFW_PLAN_CACHE_GB) so that a corpus of distinct lengths does not pay a hipMalloc / hipFree -- a device-wide
// synchronisation -- per file.  A request takes the smallest block that holds it with at most a quarter to spare (block
// sizes of distinct lengths rarely recur exactly); whoever fails to allocate -- the pool itself, the workspaces -- gives the
// whole pool back first.
void pool_release(hpfw_gpu *h)
{
    for (auto &kv : h->dev_pool) (void)hipFree(kv.second);
    h->dev_pool.clear();
    h->dev_pool_bytes = 0;
}

void *pool_take(hpfw_gpu *h, size_t bytes)
{
    auto it = h->dev_pool.lower_bound(bytes);
    if (it != h->dev_pool.end() && it->first <= bytes + bytes / 4) {
        void *p = it->second;
        h->dev_pool_bytes -= it->first;      // (it comes back under the size asked for here: the label only ever shrinks)
        h->dev_pool.erase(it);
        return p;
    }
    void *d = nullptr;
    if (hipMalloc(&d, bytes) != hipSuccess) {
        (void)hipGetLastError();
        pool_release(h);
        if (hipMalloc(&d, bytes) != hipSuccess) return nullptr;
    }
    return d;
}

void pool_give(hpfw_gpu *h, void *p, size_t bytes)
{
    if (h && h->dev_pool_bytes + bytes <= ((size_t)4 << 30)) {
        h->dev_pool.emplace(bytes, p);
        h->dev_pool_bytes += bytes;
    } else {
        (void)hipFree(p);
    }
}


// Every entry point of a handle works in the handle's shared workspaces (ws[], d_clipmax, d_best, d_qa, the
// index itself ...) and only enqueues on the caller's stream.  Two calls on different streams (a torch
// side stream and the null stream, or the private non-blocking streams of the *_host entry points) would
// otherwise overlap on those buffers: each call first makes its stream wait for the event the previous
// call recorded, and records its own when it has enqueued its work.
struct Ordered {
    hpfw_gpu *h;
    hipStream_t s;
    Ordered(hpfw_gpu *h_, hipStream_t s_) : h(h_), s(s_)
    {
        if (h->order_valid && h->order_stream != s) (void)hipStreamWaitEvent(s, h->order_ev, 0);
        if (h->plan_ev_pending) {
            (void)hipStreamWaitEvent(s, h->plan_ev, 0);
            h->plan_ev_pending = false; // (later calls on other streams are ordered after this one)
        }
    }
    ~Ordered()
    {
        if (hipEventRecord(h->order_ev, s) == hipSuccess) {
            h->order_valid = true;
            h->order_stream = s;
        }
    }
};

struct Timed {
    hpfw_gpu *h;
    int kind;
    hipStream_t s;
    bool on;
    hipEvent_t a = nullptr, b = nullptr;
    Timed(hpfw_gpu *h_, int kind_, hipStream_t s_) : h(h_), kind(kind_), s(s_), on((h_->timing_mask >> kind_) & 1u)
    {
        if (!on) return;
        if (h->ev_pool.empty()) {
            (void)hipEventCreate(&a);
            (void)hipEventCreate(&b);
        } else {
            a = h->ev_pool.back().first;
            b = h->ev_pool.back().second;
            h->ev_pool.pop_back();
        }
        (void)hipEventRecord(a, s);
    }
    ~Timed()
    {
        if (!on) return;
        (void)hipEventRecord(b, s);
        h->timed.push_back({kind, a, b});
    }
};

int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(HPFW_E_HIP, std::string("launch of ") + what + ": " + hipGetErrorString(e));
    return 0;
}

int get_plan(hpfw_gpu *h, int64_t n, DevPlan **out)
{
    auto it = h->plans.find(n);
    if (it != h->plans.end()) {
        it->second->last_use = ++h->plan_clock;
        *out = it->second.get();
        return 0;
    }
    g_uploaded = 0;
    g_plan_timing = h->plan_timing.get();
    PlanTimer whole(&PlanTiming::total);
    if (g_plan_timing) ++g_plan_timing->plans;
    auto dp = std::make_unique<DevPlan>();
    dp->owner = h;
    std::string why;
    bool have_host = false;
    {
        PlanTimer t(&PlanTiming::host_wait);
        // the host half may have been prepared (or be in preparation) by a reader thread: take it, or wait for it
        std::unique_lock<std::mutex> lock(h->host_mtx);
        auto ready = h->host_ready.find(n);
        if (ready != h->host_ready.end()) {
            h->host_cv.wait(lock, [&] { return h->host_ready.find(n)->second != nullptr; });
            ready = h->host_ready.find(n);
            have_host = ready->second->n == n;     // (a failed preparation leaves an empty plan: rebuilt below for its message)
            if (have_host) dp->hp = std::move(*ready->second);
            h->host_ready.erase(ready);
        }
        h->host_seen.insert(n);
    }
    // HPFW_FORCE_BLUESTEIN=1 (tests): the chirp-z forward transform for 7-smooth lengths too
    if (!have_host) {
        PlanTimer t(&PlanTiming::host_build);
        if (!hpfw::build_plan(n, dp->hp, why, false, std::getenv("HPFW_FORCE_BLUESTEIN") != nullptr, h->conventions, false))
            return fail(HPFW_E_UNSUPPORTED, "clip length " + std::to_string(n) + ": " + why);
    }
    const hpfw::HostPlan &p = dp->hp;
    using hpfw::cf;
    int rc;
    static_assert(sizeof(hpfw::HostCf) == sizeof(cf), "complex layout");
    hpfw::RowsArgs &ra = dp->rows;
    std::memset(&ra, 0, sizeof(ra));
    ra.n1 = p.n1;
    ra.n2 = p.n2;
    ra.h = p.h;
    ra.hpad = (p.h + 31) / 32 * 32;
    ra.pair_stride = 1;
    ra.groups.n = (int)p.groups.size();
    for (size_t g = 0; g < p.groups.size(); ++g) {
        ra.groups.r1[g] = p.groups[g].first;
        ra.groups.r2[g] = p.groups[g].second;
        ra.groups.tw_off[g] = p.rows_gtw_off[g];
    }
    if ((rc = upload(p.rows_gtw, reinterpret_cast<const hpfw::HostCf **>(&ra.gtw), dp.get()))) return rc;
    if ((rc = upload(p.pos_n2, &ra.pos_n2, dp.get()))) return rc;
    if ((rc = upload(p.kb_last, &ra.kb_last, dp.get()))) return rc;
    if (hpfw::fwd_rows_lds_bytes(ra) > 160 * 1024) return fail(HPFW_E_UNSUPPORTED, "n2 exceeds the LDS");
    hpfw::BzArgs &bz = dp->bz;
    std::memset(&bz, 0, sizeof(bz));
    if (p.bluestein) {
        PlanTimer t_dev(&PlanTiming::device_tables);
        bz.n1 = p.n1;
        bz.n2 = p.n2;
        bz.n2pad = (p.n2 + 31) / 32 * 32;
        bz.kmin = p.kmin;
        bz.kmax = p.kmax;
        bz.a = p.n1 / 16;
        bz.n_tiles1 = (bz.a + 15) / 16;
        bz.k1lo = p.kmin / p.n2;
        bz.k1n = (p.kmax - 1) / p.n2 - bz.k1lo + 1;
        {
            const int need = (2 * bz.k1n + 31) / 32; // row tiles of 32 that hold the consumed rows
            bz.nt2 = need < 3 ? need : 3;
            bz.n_tiles2 = (need + bz.nt2 - 1) / bz.nt2 * bz.nt2;
        }
        // coefficient images of the column transforms (the first one's two stages; the second one's rows that hold
        // consumed bins), packed on the device
        // temporaries of this block (T_n1, the generation scratch) come from the handle's pool and go back to it: hipFree
        // waits for every stream of the device, which would stop a caller that extracts on a stream of its own from
        // preparing the next length while the previous file's kernels run
        const cf *d_tw_n1 = nullptr;
        struct PoolTmp {
            hpfw_gpu *h;
            std::vector<std::pair<void *, size_t>> v;
            ~PoolTmp() { for (auto &b : v) pool_give(h, b.first, b.second); }
            void *take(size_t bytes)
            {
                const size_t size = (bytes + 65535) / 65536 * 65536;
                void *q = pool_take(h, size);
                if (q) v.emplace_back(q, size);
                return q;
            }
        } tmp{h, {}};
        {
            void *d = tmp.take(p.tw_n1.size() * sizeof(cf));
            if (!d) return fail(HPFW_E_HIP, "out of device memory for the tables of a clip length");
            HIP_TRY(plan_h2d(h, d, p.tw_n1.data(), p.tw_n1.size() * sizeof(cf)));
            d_tw_n1 = static_cast<const cf *>(d);
        }
        {
            const size_t bytes[3] = {(size_t)bz.a * bz.n_tiles1 * 64 * sizeof(float), (size_t)bz.a * 16 * 64 * sizeof(float),
                                     (size_t)p.n1 * bz.n_tiles2 * 64 * sizeof(float)};
            const float **slot[3] = {&bz.apack1, &bz.apack3, &bz.apack2};
            for (int i = 0; i < 3; ++i) {
                void *d = nullptr;
                if ((rc = plan_alloc(dp.get(), bytes[i], &d))) return rc;
                g_uploaded += bytes[i];
                *slot[i] = static_cast<const float *>(d);
            }
            if ((rc = plan_flush(dp.get()))) return rc; // the row transform's tables are used by the kernels below
            hpfw::launch_bz_pack_stages(bz, d_tw_n1, const_cast<float *>(bz.apack1), const_cast<float *>(bz.apack3), h->plan_stream);
            hpfw::launch_bz_pack_coefficients(p.n1, bz.k1lo, bz.k1n, d_tw_n1, bz.n_tiles2, const_cast<float *>(bz.apack2), h->plan_stream);
        }
        // chirp, T_L, w[k] / L and Bhat are generated on the device (k_bluestein.hip): a corpus of
        // real recordings brings a new length with every file
        const size_t big_l = (size_t)p.n1 * p.n2, plane = hpfw::bz_plane_bytes(bz, 1);
        {
            const size_t bytes[4] = {big_l * sizeof(cf), big_l * sizeof(cf), big_l * sizeof(cf), (size_t)(p.kmax - p.kmin) * sizeof(cf)};
            const void **slot[4] = {reinterpret_cast<const void **>(&bz.wp), reinterpret_cast<const void **>(&bz.tl),
                                    reinterpret_cast<const void **>(&bz.bhat), reinterpret_cast<const void **>(&bz.wk)};
            for (int i = 0; i < 4; ++i) {
                void *d = nullptr;
                if ((rc = plan_alloc(dp.get(), bytes[i], &d))) return rc;
                g_uploaded += bytes[i];
                *slot[i] = d;
            }
        }
        void *scratch = tmp.take(big_l * 8 + plane); // back to the pool with T_n1 when this block ends, after the synchronisation below
        if (!scratch) return fail(HPFW_E_HIP, "out of device memory for the tables of a clip length");
        hpfw::launch_bz_make_tables(ra, bz, n, static_cast<float *>(scratch), static_cast<float *>(scratch) + 2 * big_l, h->plan_stream);
        // (no host wait: the temporaries go back to the pool, whose next user is ordered after these kernels on the handle's
        // table stream like every table generation and upload; the stream that extracts waits for the event recorded below)
        const hipError_t launched = hipGetLastError();
        if (launched != hipSuccess) return fail(HPFW_E_HIP, std::string("chirp-z tables: ") + hipGetErrorString(launched));
    }
    // S6 (7-smooth lengths): the column stage's twiddle digits, digit-offset correction and inter-stage twiddles
    hpfw::ColsQArgs &ca = dp->cols;
    std::memset(&ca, 0, sizeof(ca));
    dp->rows_out = hpfw::Rows2Out{p.n1, p.hq, p.q2lo, p.q2w, nullptr, nullptr, (p.n2 + 3) / 4, 5 /* kZBlock / 4 = 2^5 pieces */,
                                  (long long)2 * p.hq * hpfw::kZBlock, hpfw::kZBlock, hpfw::z_floats_per_clip(p.hq, p.n2)};
    static_assert(hpfw::kZBlock == 128, "Rows2Out::zshift4 above");
    if (!p.bluestein) {
        ca.n1 = p.n1;
        ca.n2 = p.n2;
        ca.hq = p.hq;
        ca.mt = p.cols_mt;
        ca.ks = p.cols_ks;
        ca.zclip = hpfw::z_floats_per_clip(p.hq, p.n2);
        const int8_t *img = nullptr;
        if ((rc = upload(p.cols_image, &img, dp.get()))) return rc;
        ca.image = img;
        if ((rc = upload(p.cols_corr, &ca.corr, dp.get()))) return rc;
        if ((rc = upload(p.ts_seed, reinterpret_cast<const hpfw::HostCf **>(&dp->rows_out.seed), dp.get()))) return rc;
        if ((rc = upload(p.ts_step, reinterpret_cast<const hpfw::HostCf **>(&dp->rows_out.step), dp.get()))) return rc;
    }
    hpfw::CqPlanDev &c = dp->cq;
    c.kmin = p.kmin;
    c.nk = p.kmax - p.kmin;
    c.c = p.c;
    if (p.bluestein) { // natural order from kmin on
        c.xn1 = 1;
        c.xw = 0;
        c.xq0 = p.kmin;
        c.xclip = c.nk;
        c.xmagic = 0;
    } else {           // x[k mod n1][k / n1 - q2lo], rows of q2w
        c.xn1 = p.n1;
        c.xw = p.q2w;
        c.xq0 = p.q2lo;
        c.xclip = (int64_t)p.n1 * p.q2w;
        c.xmagic = ((1ull << 40) + (unsigned long long)p.n1 - 1) / (unsigned long long)p.n1;
    }
    std::vector<int> start(p.start, p.start + 121), lg(p.lg, p.lg + 121);
    if ((rc = upload(start, &c.start, dp.get()))) return rc;
    if ((rc = upload(lg, &c.lg, dp.get()))) return rc;
    if ((rc = upload(p.g_off, &c.g_off, dp.get()))) return rc;
    // the window table itself is generated on the device, behind the uploads (below): S5, k_cq_tables.hip
    {
        void *d = nullptr;
        if ((rc = plan_alloc(dp.get(), (size_t)std::max<int64_t>(p.g_total, 1) * sizeof(cf), &d))) return rc;
        g_uploaded += (size_t)p.g_total * sizeof(cf);
        c.g = static_cast<const cf *>(d);
    }
    int g2_max_entries = 0;
    c.g2 = nullptr;
    c.g2_off = nullptr;
    c.q2a = c.nq2 = nullptr;
    c.nq2_magic = nullptr;
    c.rows_min = 4; // (measured on one box: 0..8 alike, 16 and above slower; element by element throughout +0.4 ms per 1000 clips)
    if (!p.bluestein) { // the windows once more, in the order the rows layout of the forward bins is read (kernels.h XsBandRows)
        std::vector<int> q2a(121), nq2(121);
        std::vector<unsigned> magic(121);
        std::vector<int64_t> g2_off(121);
        int64_t total = 0;
        for (int j = 0; j < 121; ++j) {
            q2a[j] = p.start[j] / p.n1;
            nq2[j] = (p.start[j] + p.lg[j] - 1) / p.n1 - q2a[j] + 1;
            magic[j] = nq2[j] >= 2 ? (unsigned)(((1ull << 32) + (unsigned)nq2[j] - 1) / (unsigned)nq2[j]) : 0u;
            g2_off[j] = total;
            total += (int64_t)p.n1 * nq2[j];
        }
        for (int j = 0; j < 121; ++j) g2_max_entries = std::max(g2_max_entries, p.n1 * nq2[j]);
        {
            void *d = nullptr;
            if ((rc = plan_alloc(dp.get(), (size_t)std::max<int64_t>(total, 1) * sizeof(cf), &d))) return rc;
            g_uploaded += (size_t)total * sizeof(cf);
            c.g2 = static_cast<const cf *>(d);
        }
        if ((rc = upload(g2_off, &c.g2_off, dp.get()))) return rc;
        if ((rc = upload(q2a, &c.q2a, dp.get()))) return rc;
        if ((rc = upload(nq2, &c.nq2, dp.get()))) return rc;
        if ((rc = upload(magic, &c.nq2_magic, dp.get()))) return rc;
    }
    for (const hpfw::BluesteinClass &bc : p.classes) {
        hpfw::CqClassDev cd;
        cd.p = bc.p;
        cd.n_bands = (int)bc.bands.size();
        cd.radix = to_radix(bc.radix);
        if ((rc = upload(bc.tw, reinterpret_cast<const hpfw::HostCf **>(&cd.tw), dp.get()))) return rc;
        if ((rc = upload(bc.gtw, reinterpret_cast<const hpfw::HostCf **>(&cd.gtw.tab), dp.get()))) return rc;
        for (int g = 0; g < 4; ++g) cd.gtw.off[g] = bc.goff[g];
        cd.gtw.mid_off = bc.mid_off;
        if ((rc = upload(bc.vrev, reinterpret_cast<const hpfw::HostCf **>(&cd.vrev), dp.get()))) return rc;
        if ((rc = upload(bc.bands, &cd.band, dp.get()))) return rc;
        cd.len0 = bc.len0;
        cd.outer = bc.outer;
        dp->cls.push_back(cd);
    }
    if ((size_t)p.n2 * sizeof(cf) > 150 * 1024) return fail(HPFW_E_UNSUPPORTED, "n2 exceeds the LDS");
    // the tables live on the device now: drop the host copies (only the sizes are read from here on)
    {
        hpfw::HostPlan &hp = dp->hp;
        std::vector<hpfw::HostCf>().swap(hp.rows_gtw);
        std::vector<hpfw::HostCf>().swap(hp.tw_n2);
        std::vector<hpfw::HostCf>().swap(hp.tw_n1);
        std::vector<hpfw::HostCf>().swap(hp.ts_seed);
        std::vector<int8_t>().swap(hp.cols_image);
        std::vector<hpfw::HostCf>().swap(hp.g);
        for (hpfw::BluesteinClass &bc : hp.classes) {
            std::vector<hpfw::HostCf>().swap(bc.tw);
            std::vector<hpfw::HostCf>().swap(bc.gtw);
            std::vector<hpfw::HostCf>().swap(bc.vrev);
        }
    }
    dp->bytes = g_uploaded;
    dp->last_use = ++h->plan_clock;
    // a corpus of files of many different lengths would otherwise keep one set of tables per length
    // (14 MB for 30 s clips, growing with the length): bound the cache (HPFW_PLAN_CACHE_GB, default 16) by
    // evicting the least recently used plans; work already queued may still read their tables, hence the sync
    size_t budget = (size_t)16 << 30;
    if (const char *e = std::getenv("HPFW_PLAN_CACHE_GB")) budget = (size_t)(std::max(0.0, std::atof(e)) * 1073741824.0);
    if (h->plan_bytes + dp->bytes > budget && !h->plans.empty()) {
        PlanTimer t(&PlanTiming::evict);
        // Plans known to be idle go one at a time, as many as the new one needs: their blocks pass through the pool to the
        // next length's tables (sizes of neighbouring lengths recur), no hipMalloc, no hipFree.  A plan that queued work may
        // still read costs a device-wide wait first: then room for a quarter of the budget is made at once, so that a corpus
        // of distinct lengths larger than the cache does not pay that wait with every file.
        size_t goal = budget;
        while (h->plan_bytes + dp->bytes > goal && !h->plans.empty()) {
            auto lru = h->plans.begin();
            for (auto q = h->plans.begin(); q != h->plans.end(); ++q)
                if (q->second->last_use < lru->second->last_use) lru = q;
            if (lru->second->last_use > h->idle_clock) {
                (void)hipDeviceSynchronize();
                h->idle_clock = h->plan_clock;
                goal = budget - budget / 4;
            }
            h->plan_bytes -= lru->second->bytes;
            {
                // an evicted length may be prepared ahead again by the reader threads the next time a file brings it
                std::scoped_lock lock(h->host_mtx);
                h->host_seen.erase(lru->first);
            }
            h->plans.erase(lru);
        }
    }
    if ((rc = plan_flush(dp.get()))) return rc;
    std::vector<char>().swap(dp->stage);
    // the constant-Q windows, generated behind the uploads of the band tables they read (S5, k_cq_tables.hip)
    {
        PlanTimer t(&PlanTiming::device_tables);
        hpfw::CqWindowBands wb;
        int lg_max = 0;
        for (int j = 0; j < 121; ++j) {
            wb.scale[j] = hpfw::cq_window_scale(h->conventions, p.big_m, p.psize[j]);
            wb.hann_den[j] = (int)hpfw::cq_hann_den(h->conventions, p.lg[j]);
            lg_max = std::max(lg_max, p.lg[j]);
        }
        hpfw::launch_cq_windows(c, wb, p.big_m, lg_max, const_cast<cf *>(c.g), h->plan_stream);
        if (c.g2) hpfw::launch_cq_windows_rows(c, p.n1, g2_max_entries, const_cast<cf *>(c.g2), h->plan_stream);
        const hipError_t launched = hipGetLastError();
        if (launched != hipSuccess) return fail(HPFW_E_HIP, std::string("constant-Q window tables: ") + hipGetErrorString(launched));
    }
    // every table of the length is on its way on the table stream: whoever uses them first waits for this (Ordered)
    HIP_TRY(hipEventRecord(h->plan_ev, h->plan_stream));
    h->plan_ev_pending = true;
    h->plan_bytes += dp->bytes;
    *out = dp.get();
    h->plans[n] = std::move(dp);
    return 0;
}

// nb: clips per front-end pass (large intermediates); ns: clips per back-end pass (S and P only)
// clips per front-end pass: the handle's batch, the number of clips, and what ~24 GB of workspace hold
int pass_clips(hpfw_gpu *h, const DevPlan *dp, int64_t n_clips)
{
    const hpfw::HostPlan &p = dp->hp;
    size_t per_clip = (size_t)121 * p.c * 4 + (size_t)64 * std::max(p.n_frames, 1) * 4;
    if (p.bluestein)
        per_clip += 2 * hpfw::bz_plane_bytes(dp->bz, 1) + (size_t)(p.kmax - p.kmin) * 8;
    else
        per_clip += (size_t)hpfw::z_floats_per_clip(p.hq, p.n2) * 4 + (size_t)p.n1 * p.q2w * 8;
    size_t work = 0;
    for (const hpfw::CqClassDev &cd : dp->cls) work = std::max(work, hpfw::cq_big_work_bytes(cd, 1));
    per_clip += work;
    const int64_t fit = std::max<int64_t>(1, (int64_t)(((size_t)24 << 30) / per_clip));
    // passes of equal size (1000 clips at a batch of 256: four passes of 250, not three and a ragged one)
    const int64_t cap = std::min<int64_t>(h->batch, fit), n = std::max<int64_t>(n_clips, 1);
    const int64_t passes = (n + cap - 1) / cap;
    return (int)((n + passes - 1) / passes);
}

int ensure_ws(hpfw_gpu *h, const DevPlan *dp, int nb, int ns)
{
    const hpfw::HostPlan &p = dp->hp;
    size_t work = 0;
    for (const hpfw::CqClassDev &cd : dp->cls) work = std::max(work, hpfw::cq_big_work_bytes(cd, nb));
    if (work) {
        int rc = ensure(&h->d_cqwork, &h->cqwork_cap, work);
        if (rc) return rc;
    }
    const size_t planar = p.bluestein ? hpfw::bz_plane_bytes(dp->bz, nb) : 0;
    // ws[0]: the column stage's output z [hq][n2] (chirp-z path: a planar buffer); ws[1]: the forward bins (XsView layout)
    const size_t need[7] = {p.bluestein ? planar : (size_t)nb * hpfw::z_floats_per_clip(p.hq, p.n2) * 4,
                            p.bluestein ? (size_t)nb * (p.kmax - p.kmin) * 8 : (size_t)nb * p.n1 * p.q2w * 8,
                            (size_t)ns * 121 * p.c * 4, (size_t)ns * 64 * (size_t)std::max(p.n_frames, 1) * 4, // (P: the f32-chain projection only)
                            (size_t)ns * 121 * hpfw::kCqMaxWaves * 4,
                            0, planar};
    for (int i = 0; i < 7; ++i) {
        int rc = ensure(&h->ws[i], &h->ws_bytes[i], need[i], h);
        if (rc) return rc;
    }
    return ensure((void **)&h->d_clipmax, &h->clipmax_cap, (size_t)ns * 4, h);
}

// the side streams and their events (chirp-z classes side by side, chunks of the forward transform in turn)
int ensure_side_streams(hpfw_gpu *h)
{
    if (h->cq_fork) return HPFW_OK;
    bool ok = hipEventCreateWithFlags(&h->cq_fork, hipEventDisableTiming) == hipSuccess;
    for (int k = 0; k < hpfw_gpu::kCqSide && ok; ++k)
        ok = hipStreamCreateWithFlags(&h->cq_side[k], hipStreamNonBlocking) == hipSuccess &&
             hipEventCreateWithFlags(&h->cq_join[k], hipEventDisableTiming) == hipSuccess;
    return ok ? HPFW_OK : fail(HPFW_E_HIP, "side streams");
}

#if defined(HPFW_ROWS_SNAP) || defined(HPFW_ROWS_STAMPS)
hpfw::cf *g_rows_snap = nullptr; // diagnosis builds: fft_rows.h HPFW_SNAP / HPFW_STAMP
#endif

// a1 + the forward transform for nb clips: PCM -> bins [kmin, kmax) in x
int run_forward(hpfw_gpu *h, DevPlan *dp, const int16_t *d_pcm, int nb, hpfw::cf *x, hipStream_t s)
{
    const hpfw::HostPlan &p = dp->hp;
    float *yp = (float *)h->ws[0];
    int rc;
    if (p.bluestein) { // S15: the clip length has a prime factor above 7
        float *other = (float *)h->ws[6];
        if (h->bz_chunk > 0 && nb >= 6 * h->bz_chunk) {
            // in chunks taken in turn by the streams, as below: 24 MB per clip between the three kernels
            const int lanes = h->fwd_streams;
            if ((rc = ensure_side_streams(h))) return rc;
            if (lanes > 1) {
                HIP_TRY(hipEventRecord(h->cq_fork, s));
                for (int k = 0; k + 1 < lanes; ++k) HIP_TRY(hipStreamWaitEvent(h->cq_side[k], h->cq_fork, 0));
            }
            const int64_t region = (int64_t)(hpfw::bz_plane_bytes(dp->bz, h->bz_chunk) / sizeof(float));
            int i = 0;
            for (int c0 = 0; c0 < nb; c0 += h->bz_chunk, ++i) {
                const int nc = std::min(nb - c0, h->bz_chunk);
                const int lane = i % lanes;
                hipStream_t st = lane ? h->cq_side[lane - 1] : s;
                float *ya = yp + lane * region, *yb = other + lane * region;
                {
                    Timed t(h, K_COLS, st);
                    hpfw::launch_bz_cols_first(dp->bz, d_pcm + (int64_t)c0 * p.n, p.n, nc, ya, st);
                }
                {
                    Timed t(h, K_ROWS, st);
                    hpfw::launch_bz_rows_both(dp->rows, dp->bz, ya, nc, yb, st);
                }
                {
                    Timed t(h, K_COLS, st);
                    hpfw::launch_bz_cols_last(dp->bz, yb, nc, x + (int64_t)c0 * dp->cq.xclip, st);
                }
            }
            for (int k = 0; k + 1 < lanes; ++k) {
                HIP_TRY(hipEventRecord(h->cq_join[k], h->cq_side[k]));
                HIP_TRY(hipStreamWaitEvent(s, h->cq_join[k], 0));
            }
            return check_launch("bz_chunks");
        }
        {
            Timed t(h, K_COLS, s);
            hpfw::launch_bz_cols_first(dp->bz, d_pcm, p.n, nb, yp, s); // pcm as it lies -> G' [q1][k2']
        }
        if ((rc = check_launch("bz_cols"))) return rc;
        {
            Timed t(h, K_ROWS, s);
            hpfw::launch_bz_rows_both(dp->rows, dp->bz, yp, nb, other, s); // G' -> A -> C -> H' [q1][m2]
        }
        if ((rc = check_launch("bz_rows"))) return rc;
        {
            Timed t(h, K_COLS, s);
            hpfw::launch_bz_cols_last(dp->bz, other, nb, x, s);
        }
        return check_launch("bz_cols");
    }
    Timed span(h, K_FWD, s);                     // the whole forward transform as one span (its chunks overlap)
    hpfw::ColsQArgs cols = dp->cols;
    cols.variant = h->cols_variant;
    if (h->fwd_chunk > 0 && nb >= 6 * h->fwd_chunk) {
        const int lanes = h->fwd_streams;
        if ((rc = ensure_side_streams(h))) return rc;
        if (lanes > 1) {
            HIP_TRY(hipEventRecord(h->cq_fork, s));
            for (int k = 0; k + 1 < lanes; ++k) HIP_TRY(hipStreamWaitEvent(h->cq_side[k], h->cq_fork, 0));
        }
        // a stream's chunks follow each other in order, so every stream has one region of z of its own
        const int64_t region = (int64_t)h->fwd_chunk * dp->rows_out.zclip;
        int i = 0;
        for (int c0 = 0; c0 < nb; c0 += h->fwd_chunk, ++i) {
            const int nc = std::min(nb - c0, h->fwd_chunk);
            const int lane = i % lanes;
            hipStream_t st = lane ? h->cq_side[lane - 1] : s;
            float *zr = (float *)h->ws[0] + lane * region;
            {
                Timed t(h, K_COLS, st);
                hpfw::launch_fwd_cols_q(cols, d_pcm + (int64_t)c0 * p.n, p.n, nc, zr, st);
            }
            {
                Timed t(h, K_ROWS, st);
                hpfw::launch_fwd_rows2(dp->rows, dp->rows_out, zr, nc, x + (int64_t)c0 * dp->rows_out.n1 * dp->rows_out.q2w, st);
            }
        }
        for (int k = 0; k + 1 < lanes; ++k) {
            HIP_TRY(hipEventRecord(h->cq_join[k], h->cq_side[k]));
            HIP_TRY(hipStreamWaitEvent(s, h->cq_join[k], 0));
        }
        return check_launch("fwd_chunks");
    }
    {
        Timed t(h, K_COLS, s);
        hpfw::launch_fwd_cols_q(cols, d_pcm, p.n, nb, (float *)h->ws[0], s); // pcm as it lies -> z [hq][Re, Im][n2]
    }
    if ((rc = check_launch("fwd_cols"))) return rc;
    {
        Timed t(h, K_ROWS, s);
#if defined(HPFW_ROWS_SNAP)
        dp->rows.snap = g_rows_snap;
#endif
#if defined(HPFW_ROWS_STAMPS)
        dp->rows.stamps = reinterpret_cast<long long *>(g_rows_snap);
#endif
        hpfw::launch_fwd_rows2(dp->rows, dp->rows_out, (const float *)h->ws[0], nb, x, s); // -> x [n1][q2w]
    }
    return check_launch("fwd_rows");
}

// front end for nb clips: PCM -> dB terms t (and their per-clip maximum in d_clipmax) at clip slot
// `slot` of the S workspace; finish_db: also turn them into the dB spectrogram S = max(t - t_max, -80)
// in place (the projection does that itself while staging, the covariance wants S)
int run_front(hpfw_gpu *h, DevPlan *dp, const int16_t *d_pcm, int nb, int slot, bool finish_db, hipStream_t s)
{
    using hpfw::cf;
    const hpfw::HostPlan &p = dp->hp;
    cf *x = (cf *)h->ws[1];
    float *mag = (float *)h->ws[2] + (size_t)slot * 121 * p.c;
    float *mm = (float *)h->ws[4] + (size_t)slot * 121 * hpfw::kCqMaxWaves; // this pass's wave maxima
    int rc;
    if ((rc = run_forward(h, dp, d_pcm, nb, x, s))) return rc;
    {
        // fork: classes that run in LDS alone go to the side streams in turn (the caller's stream takes one too), largest
        // first -- dp->cls is in ascending order of size; classes with passes through the shared global workspace stay
        // on the caller's stream.  join: the caller's stream waits for every side stream used.
        Timed t(h, K_CQ, s);
        int n_lds = 0;
        for (const hpfw::CqClassDev &cd : dp->cls) n_lds += cd.outer ? 0 : 1;
        // (a handful of clips: the five launches are tens of microseconds each, and forking costs the host a dozen calls)
        const bool fork = h->cq_concurrent && n_lds > 1 && nb >= 4;
        if (fork && (rc = ensure_side_streams(h))) return rc;
        if (fork) {
            HIP_TRY(hipEventRecord(h->cq_fork, s));
            for (int k = 0; k < hpfw_gpu::kCqSide; ++k) HIP_TRY(hipStreamWaitEvent(h->cq_side[k], h->cq_fork, 0));
        }
        unsigned used = 0;
        int turn = 0;
        for (size_t ci = dp->cls.size(); ci-- > 0;) {
            const hpfw::CqClassDev &cd = dp->cls[ci];
            if (cd.outer) {
                hpfw::launch_cq_big_class(dp->cq, cd, x, nb, (cf *)h->d_cqwork, mag, mm, true, s);
                continue;
            }
            const int lane = fork ? turn++ % (hpfw_gpu::kCqSide + 1) : 0; // 0: the caller's stream
            if (lane) used |= 1u << (lane - 1);
            hpfw::launch_cq_class(dp->cq, cd, x, nb, mag, mm, true, lane ? h->cq_side[lane - 1] : s);
        }
        for (int k = 0; k < hpfw_gpu::kCqSide; ++k)
            if (used >> k & 1u) {
                HIP_TRY(hipEventRecord(h->cq_join[k], h->cq_side[k]));
                HIP_TRY(hipStreamWaitEvent(s, h->cq_join[k], 0));
            }
    }
    if ((rc = check_launch("cq_chirpz"))) return rc;
    {
        Timed t(h, K_DB, s);
        hpfw::launch_clipmax(mm, h->d_clipmax + slot, nb, s);
        if (finish_db) hpfw::launch_db_finish(mag, h->d_clipmax + slot, nb, (int64_t)121 * p.c, s);
    }
    return check_launch("db");
}

// back end for ns clips: dB spectrograms of the S workspace -> hashprints
int run_back(hpfw_gpu *h, DevPlan *dp, int ns, uint64_t *d_hp, hipStream_t s)
{
    const hpfw::HostPlan &p = dp->hp;
    float *sdb = (float *)h->ws[2];
    float *proj = (float *)h->ws[3];
    int rc;
    if (h->projection) { // S9q: reference level, clip, exact integer sums on the int8 matrix pipe, sign and pack in ONE kernel
        {
            Timed t(h, K_PROJECT, s);
            hpfw::launch_hashprints_q(h->d_fq_image, sdb, h->d_clipmax, ns, p.c, d_hp, nullptr, s);
        }
        return check_launch("project");
    }
    {
        Timed t(h, K_PROJECT, s);
        hpfw::launch_project(h->d_fpack, sdb, h->d_clipmax, ns, p.c, proj, s);
    }
    if ((rc = check_launch("project"))) return rc;
    {
        Timed t(h, K_PACK, s);
        hpfw::launch_pack(proj, ns, p.n_frames, d_hp, s);
    }
    return check_launch("delta_pack");
}

constexpr int kBackBatch = 1024; // clips per projection launch: ~10^4 workgroups, a small launch tail

} // namespace

extern "C" {

const char *hpfw_gpu_last_error(void) { return g_err.c_str(); }
// used by legacy.cpp so that the file entry points report through the same thread-local message
void hpfw_internal_set_error(const char *msg) { g_err = msg ? msg : ""; }
// (legacy.cpp, after it has waited for the stream that carried everything it queued on the handle)
void hpfw_internal_note_idle(hpfw_gpu *h)
{
    if (h) h->idle_clock = h->plan_clock;
}
const char *hpfw_gpu_version(void) { return "hpfw-gpu 0.1 (gfx950)"; }

int hpfw_gpu_create(int device, hpfw_gpu **out)
{
    if (!out) return fail(HPFW_E_INVALID, "out is NULL");
    *out = nullptr;
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return fail(HPFW_E_INVALID, "no such device");
    HIP_TRY(hipSetDevice(device));
    auto *h = new hpfw_gpu();
    h->device = device;
    if (std::getenv("HPFW_CQ_SERIAL")) h->cq_concurrent = 0;
    if (const char *e = std::getenv("HPFW_FWD_CHUNK")) h->fwd_chunk = std::max(0, atoi(e));
    if (const char *e = std::getenv("HPFW_BZ_CHUNK")) h->bz_chunk = std::max(0, atoi(e));
    if (const char *e = std::getenv("HPFW_COLS_VARIANT")) h->cols_variant = atoi(e);
    if (std::getenv("HPFW_PLAN_TIMING")) h->plan_timing = std::make_unique<PlanTiming>();
    if (const char *e = std::getenv("HPFW_BACK_OVERLAP")) h->back_overlap = atoi(e);
    if (const char *e = std::getenv("HPFW_FWD_STREAMS")) h->fwd_streams = std::min(hpfw_gpu::kCqSide + 1, std::max(1, atoi(e)));
    if (const char *e = std::getenv("HPFW_PROJECTION")) // "f32": handles start with the f32 fma chain (hpfw_gpu_set_projection(h, 0))
        h->projection = std::strcmp(e, "f32") == 0 ? 0 : 1;
    if (hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess ||
        hipEventCreateWithFlags(&h->order_ev, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->plan_ev, hipEventDisableTiming) != hipSuccess ||
        hipStreamCreateWithFlags(&h->plan_stream, hipStreamNonBlocking) != hipSuccess) {
        delete h;
        return fail(HPFW_E_HIP, "hipEventCreate failed");
    }
    *out = h;
    return 0;
}

int hpfw_gpu_device(const hpfw_gpu *h) { return h ? h->device : -1; }

void hpfw_gpu_destroy(hpfw_gpu *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    if (h->plan_timing && h->plan_timing->plans) {
        const PlanTiming &t = *h->plan_timing;
        std::fprintf(stderr, "hpfw plan timing: %ld lengths, %.1f ms in get_plan = %.3f ms each: wait for the host half %.3f, host build %.3f, "
                             "uploads %.3f (%ld copies, %.2f MB per length), device tables (incl. their uploads) %.3f, device memory %.3f, "
                             "evictions %.3f\n",
                     t.plans, t.total * 1e3, t.total * 1e3 / t.plans, t.host_wait * 1e3 / t.plans, t.host_build * 1e3 / t.plans,
                     t.upload * 1e3 / t.plans, t.copies, t.copied / 1e6 / t.plans, t.device_tables * 1e3 / t.plans, t.alloc * 1e3 / t.plans,
                     t.evict * 1e3 / t.plans);
    }
    h->plans.clear();
    h->plan_bytes = 0;
    for (auto &kv : h->dev_pool) (void)hipFree(kv.second);
    h->dev_pool.clear();
    for (void *p : h->ws)
        if (p) (void)hipFree(p);
    if (h->d_fpack) (void)hipFree(h->d_fpack);
    if (h->d_fq_image) (void)hipFree(h->d_fq_image);
    if (h->d_cov) (void)hipFree(h->d_cov);
    if (h->d_cov_ws) (void)hipFree(h->d_cov_ws);
    if (h->d_cqwork) (void)hipFree(h->d_cqwork);
    if (h->d_topk_scratch) (void)hipFree(h->d_topk_scratch);
    for (void *q : h->mel_owned) (void)hipFree(q);
    if (h->d_mel_work) (void)hipFree(h->d_mel_work);
    if (h->d_mel_small) (void)hipFree(h->d_mel_small);
    if (h->d_qa) (void)hipFree(h->d_qa);
    if (h->d_gk) (void)hipFree(h->d_gk);
    if (h->d_clipmax) (void)hipFree(h->d_clipmax);
    for (int b = 0; b < 2; ++b) {
        if (h->stage_pcm[b]) (void)hipFree(h->stage_pcm[b]);
        if (h->stage_copied[b]) (void)hipEventDestroy(h->stage_copied[b]);
        if (h->stage_consumed[b]) (void)hipEventDestroy(h->stage_consumed[b]);
    }
    if (h->stage_hp) (void)hipFree(h->stage_hp);
    for (int k = 0; k < hpfw_gpu::kCqSide; ++k) {
        if (h->cq_side[k]) (void)hipStreamDestroy(h->cq_side[k]);
        if (h->cq_join[k]) (void)hipEventDestroy(h->cq_join[k]);
    }
    if (h->cq_fork) (void)hipEventDestroy(h->cq_fork);
    if (h->back_side) (void)hipStreamDestroy(h->back_side);
    if (h->back_fork) (void)hipEventDestroy(h->back_fork);
    if (h->back_join) (void)hipEventDestroy(h->back_join);
    if (h->stage_copy) (void)hipStreamDestroy(h->stage_copy);
    if (h->stage_comp) (void)hipStreamDestroy(h->stage_comp);
    if (h->d_cov_tiles) (void)hipFree(h->d_cov_tiles);
    for (auto &kv : h->cfg_fpack) (void)hipFree(kv.second);
    if (h->d_cfg_proj) (void)hipFree(h->d_cfg_proj);
    for (auto &kv : h->cfg_cov) {
        if (kv.second.d_accum) (void)hipFree(kv.second.d_accum);
        if (kv.second.d_tiles) (void)hipFree(kv.second.d_tiles);
    }
    if (h->d_cfg_cov_ws) (void)hipFree(h->d_cfg_cov_ws);
    if (h->d_db) (void)hipFree(h->d_db);
    if (h->d_db_off) (void)hipFree(h->d_db_off);
    if (h->d_best) (void)hipFree(h->d_best);
    if (h->d_q_off) (void)hipFree(h->d_q_off);
    for (auto &t : h->timed) {
        (void)hipEventDestroy(t.a);
        (void)hipEventDestroy(t.b);
    }
    for (auto &e : h->ev_pool) {
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
    }
    if (h->plan_ev) (void)hipEventDestroy(h->plan_ev);
    if (h->plan_stream) (void)hipStreamDestroy(h->plan_stream);
    if (h->pin_ring) (void)hipHostFree(h->pin_ring);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->order_ev) (void)hipEventDestroy(h->order_ev);
    delete h;
}

int hpfw_gpu_set_filters(hpfw_gpu *h, const float *f)
{
    if (!h || !f) return fail(HPFW_E_INVALID, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    std::vector<float> packed((size_t)hpfw::kFilters * hpfw::kFrame);
    hpfw::pack_filters_for_mfma(f, packed.data());
    if (!h->d_fpack) HIP_TRY(hipMalloc((void **)&h->d_fpack, packed.size() * 4));
    HIP_TRY(hipMemcpy(h->d_fpack, packed.data(), packed.size() * 4, hipMemcpyHostToDevice));
    std::vector<int8_t> image;
    hpfw::pack_filters_q(f, image);
    if (!h->d_fq_image) HIP_TRY(hipMalloc(&h->d_fq_image, image.size()));
    HIP_TRY(hipMemcpy(h->d_fq_image, image.data(), image.size(), hipMemcpyHostToDevice));
    h->has_filters = true;
    return 0;
}

int hpfw_gpu_set_projection(hpfw_gpu *h, int mode)
{
    if (!h || (mode != 0 && mode != 1)) return fail(HPFW_E_INVALID, "projection mode must be 0 (f32 chain) or 1 (fixed point)");
    h->projection = mode;
    return 0;
}

int hpfw_gpu_get_projection(hpfw_gpu *h) { return h ? h->projection : -1; }

// dB spectrograms [n_clips][121][c] (device) -> hashprints [n_clips][c - 99] with the handle's projection
int hpfw_gpu_hashprints_from_db(hpfw_gpu *h, const float *d_db, int64_t n_clips, int64_t c, uint64_t *d_hp, void *stream)
{
    if (!h || !d_db || !d_hp) return fail(HPFW_E_INVALID, "null argument");
    if (!h->has_filters) return fail(HPFW_E_NOFILTERS, "no filters: call hpfw_gpu_set_filters or hpfw_gpu_learn_filters first");
    const int64_t nf = c - (hpfw::kCtx - 1), nhp = nf - hpfw::kLag;
    if (nhp <= 0) return 0;
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    Ordered ordered(h, s);
    int rc;
    const int nbmax = 256;
    if (!h->projection && (rc = ensure(&h->ws[3], &h->ws_bytes[3], (size_t)nbmax * 64 * (size_t)nf * 4))) return rc;
    for (int64_t c0 = 0; c0 < n_clips; c0 += nbmax) {
        const int nb = (int)std::min<int64_t>(nbmax, n_clips - c0);
        if (h->projection) {
            hpfw::launch_hashprints_q(h->d_fq_image, d_db + c0 * 121 * c, nullptr, nb, (int)c, d_hp + c0 * nhp, nullptr, s);
        } else {
            hpfw::launch_project(h->d_fpack, d_db + c0 * 121 * c, nullptr, nb, (int)c, (float *)h->ws[3], s);
            hpfw::launch_pack((const float *)h->ws[3], nb, (int)nf, d_hp + c0 * nhp, s);
        }
        if ((rc = check_launch("project"))) return rc;
    }
    return 0;
}

int hpfw_gpu_stage_delta_q(hpfw_gpu *h, const float *d_db, int64_t n_clips, int64_t c, int64_t *d_delta, uint64_t *d_hp, void *stream)
{
    if (!h || !d_db || !d_delta) return fail(HPFW_E_INVALID, "null argument");
    if (!h->has_filters) return fail(HPFW_E_NOFILTERS, "no filters: call hpfw_gpu_set_filters or hpfw_gpu_learn_filters first");
    const int64_t nhp = c - (hpfw::kCtx - 1) - hpfw::kLag;
    if (nhp <= 0 || n_clips <= 0) return 0;
    if (n_clips > 65535) return fail(HPFW_E_INVALID, "at most 65535 clips per call");
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    Ordered ordered(h, s);
    int rc;
    uint64_t *hp = d_hp;
    if (!hp) { // the kernel always writes its hashprints
        if ((rc = ensure(&h->ws[3], &h->ws_bytes[3], (size_t)n_clips * (size_t)nhp * 8))) return rc;
        hp = (uint64_t *)h->ws[3];
    }
    hpfw::launch_hashprints_q(h->d_fq_image, d_db, nullptr, (int)n_clips, (int)c, hp, (long long *)d_delta, s);
    return check_launch("project");
}

int hpfw_gpu_geometry(hpfw_gpu *h, int64_t n_samples, hpfw_geometry *out)
{
    if (!h || !out) return fail(HPFW_E_INVALID, "null argument");
    // The sizes alone: from the cached plan or from the host half a reader thread prepared, else by the geometry part of
    // the plan (microseconds).  No device table is built for the question -- a caller that asks for the geometry of every
    // file of a window before extracting the first would otherwise build all their tables with the GPU idle.
    auto fill = [&](const hpfw::HostPlan &p) { *out = {p.n, p.n1, p.n2, p.kmin, p.kmax, p.m, p.c, p.n_frames, p.n_hp}; };
    auto it = h->plans.find(n_samples);
    if (it != h->plans.end()) {
        fill(it->second->hp);
        return 0;
    }
    {
        std::scoped_lock lock(h->host_mtx);
        auto ready = h->host_ready.find(n_samples);
        if (ready != h->host_ready.end() && ready->second && ready->second->n == n_samples) {
            fill(*ready->second);
            return 0;
        }
    }
    hpfw::HostPlan hp;
    std::string why;
    if (!hpfw::build_plan(n_samples, hp, why, true, std::getenv("HPFW_FORCE_BLUESTEIN") != nullptr, h->conventions))
        return fail(HPFW_E_UNSUPPORTED, "clip length " + std::to_string(n_samples) + ": " + why);
    // what get_plan would refuse later is refused here (callers size their buffers from this answer)
    if ((size_t)hp.n2 * sizeof(hpfw::cf) > 150 * 1024)
        return fail(HPFW_E_UNSUPPORTED, "clip length " + std::to_string(n_samples) + ": n2 exceeds the LDS");
    fill(hp);
    return 0;
}

int hpfw_gpu_set_conventions(hpfw_gpu *h, unsigned flags)
{
    if (!h || flags > hpfw::kConvAll) return fail(HPFW_E_INVALID, "unknown convention flag");
    HIP_TRY(hipSetDevice(h->device));
    if (flags != h->conventions) { // the tables of every cached length were built under the old conventions
        HIP_TRY(hipDeviceSynchronize());
        h->plans.clear();
        h->plan_bytes = 0;
        std::unique_lock<std::mutex> lock(h->host_mtx);
        h->host_cv.wait(lock, [&] { // (preparations in flight finish first: their threads write into the map)
            for (auto &kv : h->host_ready)
                if (!kv.second) return false;
            return true;
        });
        h->host_ready.clear();
        h->host_seen.clear();
    }
    h->conventions = flags;
    return 0;
}

int hpfw_gpu_set_batch(hpfw_gpu *h, int clips)
{
    if (!h || clips < 0 || clips > 4096) return fail(HPFW_E_INVALID, "batch out of range");
    h->batch = clips == 0 ? 256 : clips;
    return 0;
}

int hpfw_gpu_extract_pcm16(hpfw_gpu *h, const int16_t *d_pcm, int64_t n_samples, int64_t n_clips,
                           uint64_t *d_hp, void *stream)
{
    if (!h || !d_pcm || !d_hp || n_clips < 0) return fail(HPFW_E_INVALID, "bad argument");
    if (!h->has_filters) return fail(HPFW_E_NOFILTERS, "no filters loaded: call hpfw_gpu_set_filters first");
    HIP_TRY(hipSetDevice(h->device));
    DevPlan *dp;
    int rc = get_plan(h, n_samples, &dp);
    if (rc) return rc;
    if (dp->hp.n_hp <= 0) return fail(HPFW_E_UNSUPPORTED, "clip too short to yield a hashprint");
    hipStream_t s = (hipStream_t)stream;
    Ordered ordered(h, s);
    const int nbmax = pass_clips(h, dp, n_clips);
    const int nsmax = (int)std::min<int64_t>(std::max(kBackBatch, nbmax), std::max<int64_t>(n_clips, 1));
    if ((rc = ensure_ws(h, dp, nbmax, nsmax))) return rc;
    // several passes and the fixed-point back end: the hashprints of pass i are computed on a stream of their own while the
    // front end of pass i + 1 runs on the caller's (different pipes: matrix against vector ALU and LDS)
    const bool overlap = h->back_overlap && h->projection && n_clips > nbmax;
    if (overlap && !h->back_side) {
        if (hipStreamCreateWithFlags(&h->back_side, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&h->back_fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&h->back_join, hipEventDisableTiming) != hipSuccess)
            return fail(HPFW_E_HIP, "back-end stream");
    }
    for (int64_t s0 = 0; s0 < n_clips; s0 += nsmax) {
        const int ns = (int)std::min<int64_t>(nsmax, n_clips - s0);
        for (int c0 = 0; c0 < ns; c0 += nbmax) {
            const int nb = std::min(nbmax, ns - c0);
            rc = run_front(h, dp, d_pcm + (s0 + c0) * n_samples, nb, c0, false, s);
            if (rc) return rc;
            if (overlap) {
                HIP_TRY(hipEventRecord(h->back_fork, s));
                HIP_TRY(hipStreamWaitEvent(h->back_side, h->back_fork, 0));
                Timed t(h, K_PROJECT, h->back_side);
                hpfw::launch_hashprints_q(h->d_fq_image, (const float *)h->ws[2] + (size_t)c0 * 121 * dp->hp.c, h->d_clipmax + c0, nb, dp->hp.c,
                                          d_hp + (s0 + c0) * dp->hp.n_hp, nullptr, h->back_side);
                if ((rc = check_launch("project"))) return rc;
            }
        }
        if (overlap) { // the caller's stream (and the next batch's passes, which write the same slots) behind the last back end
            HIP_TRY(hipEventRecord(h->back_join, h->back_side));
            HIP_TRY(hipStreamWaitEvent(s, h->back_join, 0));
        } else {
            rc = run_back(h, dp, ns, d_hp + s0 * dp->hp.n_hp, s);
            if (rc) return rc;
        }
    }
    return 0;
}

int hpfw_gpu_extract_pcm16_host(hpfw_gpu *h, const int16_t *pcm, int64_t n_samples, int64_t n_clips,
                                uint64_t *hp)
{
    if (!h || !pcm || !hp || n_clips < 0) return fail(HPFW_E_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(h->device));
    hpfw_geometry g;
    int rc = hpfw_gpu_geometry(h, n_samples, &g);
    if (rc) return rc;
    if (n_clips == 0) return 0;
    // uploads in chunks on a copy stream, two device buffers deep, so that the PCIe transfer of chunk
    // i + 1 runs under the kernels of chunk i (from pinned host memory; a pageable source is staged
    // by the runtime and overlaps only partly)
    const int64_t chunk = std::min<int64_t>(n_clips, std::max<int64_t>(1, (192ll << 20) / (n_samples * 2)));
    if ((rc = ensure(&h->stage_hp, &h->stage_hp_cap, (size_t)n_clips * std::max<int64_t>(g.n_hp, 1) * 8))) return rc;
    for (int b = 0; b < 2; ++b) {
        if (b == 1 && chunk >= n_clips) break; // one chunk: one buffer
        if ((rc = ensure(&h->stage_pcm[b], &h->stage_pcm_cap[b], (size_t)chunk * n_samples * 2))) return rc;
    }
    if (!h->stage_copy) {
        HIP_TRY(hipStreamCreateWithFlags(&h->stage_copy, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&h->stage_comp, hipStreamNonBlocking));
        for (int b = 0; b < 2; ++b) {
            HIP_TRY(hipEventCreateWithFlags(&h->stage_copied[b], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&h->stage_consumed[b], hipEventDisableTiming));
        }
    }
    hipStream_t s_copy = h->stage_copy, s_comp = h->stage_comp;
    uint64_t *d_hp = (uint64_t *)h->stage_hp;
    int64_t ci = 0;
    for (int64_t c0 = 0; !rc && c0 < n_clips; c0 += chunk, ++ci) {
        const int b = (int)(ci & 1);
        const int64_t cnt = std::min(chunk, n_clips - c0);
        int16_t *d_pcm = (int16_t *)h->stage_pcm[b];
        if (ci >= 2 && hipStreamWaitEvent(s_copy, h->stage_consumed[b], 0) != hipSuccess) rc = fail(HPFW_E_HIP, "event wait failed");
        if (!rc && hipMemcpyAsync(d_pcm, pcm + c0 * n_samples, (size_t)cnt * n_samples * 2, hipMemcpyHostToDevice, s_copy) !=
                       hipSuccess)
            rc = fail(HPFW_E_HIP, "H2D copy failed");
        if (!rc && (hipEventRecord(h->stage_copied[b], s_copy) != hipSuccess ||
                    hipStreamWaitEvent(s_comp, h->stage_copied[b], 0) != hipSuccess))
            rc = fail(HPFW_E_HIP, "event record failed");
        if (!rc) rc = hpfw_gpu_extract_pcm16(h, d_pcm, n_samples, cnt, d_hp + c0 * g.n_hp, s_comp);
        if (!rc && hipEventRecord(h->stage_consumed[b], s_comp) != hipSuccess) rc = fail(HPFW_E_HIP, "event record failed");
    }
    if (hipStreamSynchronize(s_copy) != hipSuccess && !rc) rc = fail(HPFW_E_HIP, "H2D copy failed");
    if (!rc && hipMemcpyAsync(hp, d_hp, (size_t)n_clips * g.n_hp * 8, hipMemcpyDeviceToHost, s_comp) != hipSuccess)
        rc = fail(HPFW_E_HIP, "D2H copy failed");
    if (hipStreamSynchronize(s_comp) != hipSuccess && !rc) rc = fail(HPFW_E_HIP, "kernel execution failed");
    return rc;
}

// Host half of the tables of a clip length, built on the CALLING thread and kept for the next entry point that meets
// the length (which then only generates / uploads the device tables).  Thread-safe against every other call on the
// handle: the collectors' reader threads call it for each file they have decoded.
int hpfw_gpu_prepare_length(hpfw_gpu *h, int64_t n_samples)
{
    if (!h) return fail(HPFW_E_INVALID, "null handle");
    {
        std::scoped_lock lock(h->host_mtx);
        if (!h->host_seen.insert(n_samples).second) return 0; // known already
        h->host_ready[n_samples] = nullptr;                   // in preparation
    }
    auto hp = std::make_unique<hpfw::HostPlan>();
    std::string why;
    bool ok;
    {
        hpfw::PlanSerial serial;
        ok = hpfw::build_plan(n_samples, *hp, why, false, std::getenv("HPFW_FORCE_BLUESTEIN") != nullptr, h->conventions, false);
    }
    if (!ok) *hp = hpfw::HostPlan(); // n = 0: get_plan builds it again and reports why
    {
        std::scoped_lock lock(h->host_mtx);
        h->host_ready[n_samples] = std::move(hp);
        if (!ok) h->host_seen.erase(n_samples); // an unsupported length says so every time it is asked for
    }
    h->host_cv.notify_all();
    return ok ? 0 : fail(HPFW_E_UNSUPPORTED, "clip length " + std::to_string(n_samples) + ": " + why);
}

#if defined(HPFW_ROWS_SNAP) || defined(HPFW_ROWS_STAMPS)
int hpfw_gpu_debug_set_rows_snap(void *d_snap)
{
    g_rows_snap = static_cast<hpfw::cf *>(d_snap);
    return 0;
}
#endif

int hpfw_gpu_debug_workspace(hpfw_gpu *h, int which, void **d_ptr, size_t *bytes)
{
    if (!h || which < 0 || which >= 7 || !d_ptr || !bytes) return fail(HPFW_E_INVALID, "bad argument");
    *d_ptr = h->ws[which];
    *bytes = h->ws_bytes[which];
    return 0;
}

// ---- diagnostic: the device-generated tables of the chirp-z forward transform ------------------
int hpfw_gpu_chirpz_table(hpfw_gpu *h, int64_t n_samples, int which, float *out, int64_t capacity, int64_t *count)
{
    if (!h || !count) return fail(HPFW_E_INVALID, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    DevPlan *dp;
    int rc = get_plan(h, n_samples, &dp);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(h->plan_stream)); // the tables are generated on the handle's table stream
    if (which == 4) { // the constant-Q stage's windows (every length): bands concatenated
        *count = 2 * dp->hp.g_total;
        if (!out) return 0;
        if (capacity < *count) return fail(HPFW_E_INVALID, "table buffer too small");
        HIP_TRY(hipMemcpy(out, dp->cq.g, (size_t)*count * sizeof(float), hipMemcpyDeviceToHost));
        return 0;
    }
    if (!dp->hp.bluestein) return fail(HPFW_E_INVALID, "clip length takes the mixed-radix transform: no chirp-z tables");
    const hpfw::BzArgs &bz = dp->bz;
    const void *tab[4] = {bz.wp, bz.tl, bz.bhat, bz.wk};
    if (which < 0 || which > 3) return fail(HPFW_E_INVALID, "table index out of range");
    *count = 2 * (which == 3 ? (int64_t)(bz.kmax - bz.kmin) : (int64_t)bz.n1 * bz.n2);
    if (!out) return 0;
    if (capacity < *count) return fail(HPFW_E_INVALID, "table buffer too small");
    if (which == 0) { // the chirp lies in two planes on the device
        std::vector<float> planar((size_t)*count);
        HIP_TRY(hipMemcpy(planar.data(), tab[0], planar.size() * sizeof(float), hipMemcpyDeviceToHost));
        const size_t big_l = planar.size() / 2;
        for (size_t j = 0; j < big_l; ++j) {
            out[2 * j] = planar[j];
            out[2 * j + 1] = planar[big_l + j];
        }
        return 0;
    }
    HIP_TRY(hipMemcpy(out, tab[which], (size_t)*count * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

// ---- stages ----------------------------------------------------------------------------------
int hpfw_gpu_stage_spectrum(hpfw_gpu *h, const int16_t *d_pcm, int64_t n_samples, int64_t n_clips,
                            float *d_x, void *stream)
{
    if (!h || !d_pcm || !d_x) return fail(HPFW_E_INVALID, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    DevPlan *dp;
    int rc = get_plan(h, n_samples, &dp);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    Ordered ordered(h, s);
    const int nbmax = pass_clips(h, dp, n_clips);
    if ((rc = ensure_ws(h, dp, nbmax, nbmax))) return rc;
    const int64_t nk = dp->hp.kmax - dp->hp.kmin;
    for (int64_t c0 = 0; c0 < n_clips; c0 += nbmax) {
        const int nb = (int)std::min<int64_t>(nbmax, n_clips - c0);
        if ((rc = run_forward(h, dp, d_pcm + c0 * n_samples, nb, (hpfw::cf *)h->ws[1], s))) return rc;
        hpfw::launch_gather_bins(dp->cq, (const hpfw::cf *)h->ws[1], nb, (hpfw::cf *)d_x + c0 * nk, s); // natural order [kmin, kmax)
        if ((rc = check_launch("gather_bins"))) return rc;
    }
    return 0;
}

int hpfw_gpu_stage_cqmag(hpfw_gpu *h, const float *d_x, int64_t n_samples, int64_t n_clips, float *d_mag,
                         void *stream)
{
    if (!h || !d_x || !d_mag) return fail(HPFW_E_INVALID, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    DevPlan *dp;
    int rc = get_plan(h, n_samples, &dp);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    Ordered ordered(h, s);
    const int nbmax = pass_clips(h, dp, n_clips);
    if ((rc = ensure_ws(h, dp, nbmax, nbmax))) return rc;
    const int64_t nk = dp->hp.kmax - dp->hp.kmin;
    hpfw::CqPlanDev cq = dp->cq; // the caller's bins lie in natural order
    cq.xn1 = 1;
    cq.xw = 0;
    cq.xq0 = dp->hp.kmin;
    cq.xclip = nk;
    for (int64_t c0 = 0; c0 < n_clips; c0 += nbmax) {
        const int nb = (int)std::min<int64_t>(nbmax, n_clips - c0);
        for (const hpfw::CqClassDev &cd : dp->cls) {
            if (cd.outer)
                hpfw::launch_cq_big_class(cq, cd, (const hpfw::cf *)d_x + c0 * nk, nb, (hpfw::cf *)h->d_cqwork,
                                          d_mag + c0 * 121 * dp->hp.c, (float *)h->ws[4], false, s);
            else
                hpfw::launch_cq_class(cq, cd, (const hpfw::cf *)d_x + c0 * nk, nb,
                                      d_mag + c0 * 121 * dp->hp.c, (float *)h->ws[4], false, s);
        }
        if ((rc = check_launch("cq_chirpz"))) return rc;
    }
    return 0;
}

int hpfw_gpu_stage_db(hpfw_gpu *h, const float *d_mag, int64_t n_clips, int64_t c, float *d_db, void *stream)
{
    if (!h || !d_mag || !d_db || c <= 0) return fail(HPFW_E_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    Ordered ordered(h, s);
    int rc;
    const int64_t per = 121 * c;
    const int nbmax = 1024;
    if ((rc = ensure(&h->ws[4], &h->ws_bytes[4], (size_t)nbmax * 121 * hpfw::kCqMaxWaves * 4))) return rc;
    if ((rc = ensure((void **)&h->d_clipmax, &h->clipmax_cap, (size_t)nbmax * 4))) return rc;
    for (int64_t c0 = 0; c0 < n_clips; c0 += nbmax) {
        const int nb = (int)std::min<int64_t>(nbmax, n_clips - c0);
        hpfw::launch_magmax(d_mag + c0 * per, nb, (int)c, (float *)h->ws[4], s);
        hpfw::launch_db(d_mag + c0 * per, (const float *)h->ws[4], h->d_clipmax, nb, per, d_db + c0 * per, s);
        if ((rc = check_launch("db"))) return rc;
    }
    return 0;
}

int hpfw_gpu_stage_project(hpfw_gpu *h, const float *d_db, int64_t n_clips, int64_t c, float *d_proj,
                           void *stream)
{
    if (!h || !d_db || !d_proj || c < hpfw::kCtx) return fail(HPFW_E_INVALID, "bad argument");
    if (!h->has_filters) return fail(HPFW_E_NOFILTERS, "no filters loaded: call hpfw_gpu_set_filters first");
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    Ordered ordered(h, s);
    const int64_t nf = c - (hpfw::kCtx - 1);
    for (int64_t c0 = 0; c0 < n_clips; c0 += 16384) {
        const int nb = (int)std::min<int64_t>(16384, n_clips - c0);
        Timed t(h, K_PROJECT, s);
        hpfw::launch_project(h->d_fpack, d_db + c0 * 121 * c, nullptr, nb, (int)c, d_proj + c0 * 64 * nf, s);
    }
    return check_launch("project");
}

int hpfw_gpu_stage_pack(hpfw_gpu *h, const float *d_proj, int64_t n_clips, int64_t n_frames, uint64_t *d_hp,
                        void *stream)
{
    if (!h || !d_proj || !d_hp || n_frames <= hpfw::kLag) return fail(HPFW_E_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    Ordered ordered(h, s);
    for (int64_t c0 = 0; c0 < n_clips; c0 += 16384) {
        const int nb = (int)std::min<int64_t>(16384, n_clips - c0);
        hpfw::launch_pack(d_proj + c0 * 64 * n_frames, nb, (int)n_frames, d_hp + c0 * (n_frames - hpfw::kLag), s);
    }
    return check_launch("delta_pack");
}

// Hashprints of one cached dB spectrogram (collect_fingerprints over cache.get_spectros(),
// parallel_collector.h:114-137; file layout utils.h:77-106: Eigen column-major [rows = 121][cols]).
// The caller passes the matrix as stored; it is transposed to the bin-major device layout here.
int hpfw_gpu_extract_db_host(hpfw_gpu *h, const float *s_colmajor, int32_t rows, int32_t cols, uint64_t *hp,
                             int64_t hp_cap, int64_t *n_hp)
{
    if (!h || !s_colmajor || !n_hp) return fail(HPFW_E_INVALID, "null argument");
    if (rows != hpfw::kBins) return fail(HPFW_E_INVALID, "a spectrogram has 121 rows");
    if (!h->has_filters) return fail(HPFW_E_NOFILTERS, "no filters set");
    HIP_TRY(hipSetDevice(h->device));
    const int64_t nf = (int64_t)cols - (hpfw::kCtx - 1), nh = nf - hpfw::kLag;
    *n_hp = nh > 0 ? nh : 0;
    if (nh <= 0) return 0; // too short: no hashprints (hashprint_handle.h:118: empty fingerprint)
    if (!hp || hp_cap < nh) return fail(HPFW_E_INVALID, "hashprint buffer too small");
    std::vector<float> binmajor((size_t)rows * cols);
    for (int32_t c = 0; c < cols; ++c)
        for (int32_t b = 0; b < rows; ++b) binmajor[(size_t)b * cols + c] = s_colmajor[(size_t)c * rows + b];
    float *d_s = nullptr;
    uint64_t *d_h = nullptr;
    int rc = 0;
    if (hipMalloc((void **)&d_s, binmajor.size() * 4) != hipSuccess || hipMalloc((void **)&d_h, (size_t)nh * 8) != hipSuccess)
        rc = fail(HPFW_E_HIP, "out of device memory");
    if (!rc && hipMemcpy(d_s, binmajor.data(), binmajor.size() * 4, hipMemcpyHostToDevice) != hipSuccess)
        rc = fail(HPFW_E_HIP, "H2D copy failed");
    if (!rc) rc = hpfw_gpu_hashprints_from_db(h, d_s, 1, cols, d_h, nullptr);
    if (!rc && hipMemcpy(hp, d_h, (size_t)nh * 8, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(HPFW_E_HIP, "D2H copy failed");
    if (d_s) (void)hipFree(d_s);
    if (d_h) (void)hipFree(d_h);
    return rc;
}

// ---- Mel front-end (f3): MelSpectrogram<>::spectrogram (mel.h:34-104) ---------------------------
int64_t hpfw_gpu_mel_frames(int64_t n_samples) { return hpfw::mel_frames(n_samples); }

static int mel_prepare(hpfw_gpu *h)
{
    if (h->mel_ready) return 0;
    std::string why;
    if (!hpfw::build_frame_transform(hpfw::kMelFrame, h->mel_plan, why)) return fail(HPFW_E_UNSUPPORTED, why.c_str());
    const hpfw::HostPlan &p = h->mel_plan;
    hpfw::RowsArgs &ra = h->mel_rows;
    std::memset(&ra, 0, sizeof(ra));
    ra.n1 = 2;
    ra.n2 = p.n2;
    ra.h = p.h;
    ra.hpad = 2208;
    ra.pair_stride = 1;
    ra.groups.n = (int)p.groups.size();
    for (size_t g = 0; g < p.groups.size(); ++g) {
        ra.groups.r1[g] = p.groups[g].first;
        ra.groups.r2[g] = p.groups[g].second;
        ra.groups.tw_off[g] = p.rows_gtw_off[g];
    }
    int rc;
    if ((rc = upload(p.rows_gtw, reinterpret_cast<const hpfw::HostCf **>(&ra.gtw), h->mel_owned))) return rc;
    if ((rc = upload(p.tw_big, reinterpret_cast<const hpfw::HostCf **>(&ra.tw_big), h->mel_owned))) return rc;
    if ((rc = upload(p.pos_n2, &ra.pos_n2, h->mel_owned))) return rc;
    if ((rc = upload(p.kb_last, &ra.kb_last, h->mel_owned))) return rc;
    std::vector<float> win, cpack;
    hpfw::mel_tables(win, cpack);
    if ((rc = upload(win, &h->d_mel_win, h->mel_owned))) return rc;
    if ((rc = upload(cpack, &h->d_mel_cpack, h->mel_owned))) return rc;
    h->mel_ready = true;
    return 0;
}

int hpfw_gpu_mel_spectrogram_pcm16(hpfw_gpu *h, const int16_t *d_pcm, int64_t n_samples, int64_t n_clips, float *d_out,
                                   int32_t *d_cols, void *stream)
{
    if (!h || !d_pcm || !d_out || !d_cols || n_clips < 0 || n_samples < 1) return fail(HPFW_E_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(h->device));
    int rc = mel_prepare(h);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    Ordered ordered(h, s);
    const int nf = hpfw::mel_frames(n_samples), n_blk = (int)((n_samples + hpfw::kMelHop - 1) / hpfw::kMelHop);
    // clips per pass: the split spectra take 2 * 2208 floats per frame
    const int64_t per_clip = (int64_t)hpfw::mel_work_bytes(n_samples, 1);
    const int nbmax = (int)std::max<int64_t>(1, std::min<int64_t>(std::max<int64_t>(n_clips, 1), ((int64_t)8 << 30) / per_clip));
    if ((rc = ensure(&h->d_mel_work, &h->mel_work_cap, hpfw::mel_work_bytes(n_samples, nbmax)))) return rc;
    if ((rc = ensure(&h->d_mel_small, &h->mel_small_cap, (size_t)nbmax * ((size_t)n_blk * 8 + (size_t)nf * 4 + 8)))) return rc;
    for (int64_t c0 = 0; c0 < n_clips; c0 += nbmax) {
        const int nb = (int)std::min<int64_t>(nbmax, n_clips - c0);
        int64_t *blk = (int64_t *)h->d_mel_small;
        int *pos = (int *)(blk + (size_t)nbmax * n_blk);
        float *pmax = (float *)(pos + (size_t)nbmax * nf);
        hpfw::launch_mel(h->mel_rows, h->d_mel_win, h->d_mel_cpack, d_pcm + c0 * n_samples, n_samples, nb, blk, pos,
                         d_cols + c0, pmax, (float *)h->d_mel_work, d_out + c0 * hpfw::kMelBands * nf, s);
        if ((rc = check_launch("mel"))) return rc;
    }
    return 0;
}

int hpfw_gpu_mel_spectrogram_pcm16_host(hpfw_gpu *h, const int16_t *pcm, int64_t n_samples, int64_t n_clips, float *out,
                                        int32_t *cols)
{
    if (!h || !pcm || !out || !cols || n_clips < 0 || n_samples < 1) return fail(HPFW_E_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(h->device));
    if (n_clips == 0) return 0;
    const size_t per = (size_t)hpfw::kMelBands * hpfw::mel_frames(n_samples);
    int16_t *d_pcm = nullptr;
    float *d_out = nullptr;
    int32_t *d_cols = nullptr;
    int rc = 0;
    if (hipMalloc((void **)&d_pcm, (size_t)n_clips * n_samples * 2) != hipSuccess ||
        hipMalloc((void **)&d_out, (size_t)n_clips * per * 4) != hipSuccess || hipMalloc((void **)&d_cols, (size_t)n_clips * 4) != hipSuccess)
        rc = fail(HPFW_E_NOMEM, "hipMalloc failed");
    if (!rc && (hipMemcpy(d_pcm, pcm, (size_t)n_clips * n_samples * 2, hipMemcpyHostToDevice) != hipSuccess ||
                hipMemset(d_out, 0, (size_t)n_clips * per * 4) != hipSuccess))
        rc = fail(HPFW_E_HIP, "H2D copy failed");
    if (!rc) rc = hpfw_gpu_mel_spectrogram_pcm16(h, d_pcm, n_samples, n_clips, d_out, d_cols, nullptr);
    if (!rc && (hipDeviceSynchronize() != hipSuccess || hipMemcpy(out, d_out, (size_t)n_clips * per * 4, hipMemcpyDeviceToHost) != hipSuccess ||
                hipMemcpy(cols, d_cols, (size_t)n_clips * 4, hipMemcpyDeviceToHost) != hipSuccess))
        rc = fail(HPFW_E_HIP, "kernel execution or D2H copy failed");
    if (d_pcm) (void)hipFree(d_pcm);
    if (d_out) (void)hipFree(d_out);
    if (d_cols) (void)hipFree(d_cols);
    return rc;
}

// ---- HashprintHandle with other template arguments (hashprint_handle.h:50-64) -------------------------
static int cfg_check(const hpfw_handle_config *c)
{
    if (!c) return fail(HPFW_E_INVALID, "null config");
    if (c->rows < 1 || c->rows > 512 || c->context < 1 || c->context > 256 || c->lag < 1 ||
        (c->bits != 16 && c->bits != 32 && c->bits != 64))
        return fail(HPFW_E_INVALID, "config: rows 1..512, context 1..256, lag >= 1, bits 16, 32 or 64");
    hpfw::CfgArgs a{c->rows, c->context, c->lag, c->bits, nullptr};
    if (hpfw::project_cfg_lds_bytes(a) > 160 * 1024) return fail(HPFW_E_UNSUPPORTED, "config: rows x context exceeds the LDS slab");
    return 0;
}

int hpfw_gpu_cfg_set_filters(hpfw_gpu *h, const hpfw_handle_config *c, const float *f)
{
    if (!h || !f) return fail(HPFW_E_INVALID, "null argument");
    int rc = cfg_check(c);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(h->device));
    std::vector<float> packed(hpfw::cfg_fpack_floats(c->rows, c->context, c->bits));
    hpfw::pack_cfg_filters(c->rows, c->context, c->bits, f, packed.data());
    float *&d = h->cfg_fpack[{c->rows, c->context, c->bits}];
    if (!d) HIP_TRY(hipMalloc((void **)&d, packed.size() * 4));
    Ordered ordered(h, nullptr);
    HIP_TRY(hipMemcpy(d, packed.data(), packed.size() * 4, hipMemcpyHostToDevice));
    return 0;
}

int hpfw_gpu_cfg_hashprints(hpfw_gpu *h, const hpfw_handle_config *c, const float *d_s, const int32_t *d_cols, int64_t n_clips,
                            int64_t stride, void *d_hp, int64_t hp_stride, float *d_proj, void *stream)
{
    if (!h || !d_s || !d_hp || n_clips < 0 || stride < 1) return fail(HPFW_E_INVALID, "bad argument");
    int rc = cfg_check(c);
    if (rc) return rc;
    auto it = h->cfg_fpack.find({c->rows, c->context, c->bits});
    if (it == h->cfg_fpack.end()) return fail(HPFW_E_NOFILTERS, "no filters for this configuration: call hpfw_gpu_cfg_set_filters first");
    const int64_t nf = stride - c->context + 1, nhp = nf - c->lag;
    if (nhp > hp_stride) return fail(HPFW_E_INVALID, "hp_stride smaller than stride - context + 1 - lag");
    if (n_clips == 0 || nhp <= 0) return 0;
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    Ordered ordered(h, s);
    const hpfw::CfgArgs a{c->rows, c->context, c->lag, c->bits, it->second};
    // clips per pass: the projection scratch stays below 1 GiB
    const int64_t per = (int64_t)c->bits * nf * 4;
    const int64_t chunk = d_proj ? n_clips : std::max<int64_t>(1, std::min<int64_t>(n_clips, ((int64_t)1 << 30) / per));
    if (!d_proj && (rc = ensure((void **)&h->d_cfg_proj, &h->cfg_proj_cap, (size_t)chunk * per))) return rc;
    const size_t word = (size_t)c->bits / 8;
    for (int64_t c0 = 0; c0 < n_clips; c0 += chunk) {
        const int nb = (int)std::min<int64_t>(chunk, n_clips - c0);
        float *pj = d_proj ? d_proj + c0 * c->bits * nf : h->d_cfg_proj;
        const int *cols = d_cols ? d_cols + c0 : nullptr;
        {
            Timed t(h, K_PROJECT, s);
            hpfw::launch_project_cfg(a, d_s + c0 * c->rows * stride, cols, nb, stride, pj, nf, s);
        }
        if ((rc = check_launch("project_cfg"))) return rc;
        {
            Timed t(h, K_PACK, s);
            hpfw::launch_pack_cfg(a, pj, cols, nb, stride, nf, (char *)d_hp + (size_t)c0 * hp_stride * word, hp_stride, s);
        }
        if ((rc = check_launch("pack_cfg"))) return rc;
    }
    return 0;
}

static int cfg_cov_slot(hpfw_gpu *h, const hpfw_handle_config *c, hpfw_gpu::CfgCov **out)
{
    const int kt = c->rows * c->context;
    hpfw_gpu::CfgCov &cc = h->cfg_cov[{c->rows, c->context}];
    if (!cc.d_accum) {
        HIP_TRY(hipMalloc((void **)&cc.d_accum, (size_t)kt * kt * 4));
        HIP_TRY(hipMemset(cc.d_accum, 0, (size_t)kt * kt * 4));
        std::vector<int> xy((size_t)2 * hpfw::cov_cfg_tile_count(kt));
        hpfw::cov_cfg_tile_list(kt, xy.data());
        HIP_TRY(hipMalloc((void **)&cc.d_tiles, xy.size() * 4));
        HIP_TRY(hipMemcpy(cc.d_tiles, xy.data(), xy.size() * 4, hipMemcpyHostToDevice));
        cc.clips = 0;
    }
    *out = &cc;
    return 0;
}

int hpfw_gpu_cfg_cov_reset(hpfw_gpu *h, const hpfw_handle_config *c)
{
    if (!h) return fail(HPFW_E_INVALID, "null handle");
    int rc = cfg_check(c);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(h->device));
    auto it = h->cfg_cov.find({c->rows, c->context});
    if (it == h->cfg_cov.end()) return 0;
    Ordered ordered(h, nullptr);
    const int kt = c->rows * c->context;
    HIP_TRY(hipMemset(it->second.d_accum, 0, (size_t)kt * kt * 4));
    it->second.clips = 0;
    return 0;
}

int hpfw_gpu_cfg_cov_accumulate(hpfw_gpu *h, const hpfw_handle_config *c, const float *d_s, const int32_t *d_cols, int64_t n_clips,
                                int64_t stride, void *stream)
{
    if (!h || !d_s || n_clips < 0 || stride < 1) return fail(HPFW_E_INVALID, "bad argument");
    int rc = cfg_check(c);
    if (rc) return rc;
    const hpfw::CfgArgs a{c->rows, c->context, c->lag, c->bits, nullptr};
    if (!hpfw::cov_cfg_supported(a)) return fail(HPFW_E_UNSUPPORTED, "covariance: context below 9 is not supported");
    HIP_TRY(hipSetDevice(h->device));
    hpfw_gpu::CfgCov *cc;
    if ((rc = cfg_cov_slot(h, c, &cc))) return rc;
    hipStream_t s = (hipStream_t)stream;
    Ordered ordered(h, s);
    const int64_t chunk = 512; // clips per pass: bounds the partial tiles and the per-clip sums
    if ((rc = ensure((void **)&h->d_cfg_cov_ws, &h->cfg_cov_ws_cap,
                     hpfw::cov_cfg_workspace_bytes(a, (int)std::min(chunk, std::max<int64_t>(n_clips, 1))))))
        return rc;
    for (int64_t c0 = 0; c0 < n_clips; c0 += chunk) {
        const int nb = (int)std::min(chunk, n_clips - c0);
        hpfw::launch_cov_cfg(a, d_s + c0 * c->rows * stride, d_cols ? d_cols + c0 : nullptr, nb, stride, cc->d_tiles,
                             h->d_cfg_cov_ws, cc->d_accum, s);
        if ((rc = check_launch("cov_cfg"))) return rc;
    }
    cc->clips += n_clips;
    return 0;
}

int hpfw_gpu_cfg_cov_get(hpfw_gpu *h, const hpfw_handle_config *c, float *cov, int64_t *n_clips)
{
    if (!h || !cov) return fail(HPFW_E_INVALID, "null argument");
    int rc = cfg_check(c);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(h->device));
    const size_t nn = (size_t)c->rows * c->context * c->rows * c->context;
    auto it = h->cfg_cov.find({c->rows, c->context});
    if (it == h->cfg_cov.end()) {
        std::memset(cov, 0, nn * 4);
        if (n_clips) *n_clips = 0;
        return 0;
    }
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(cov, it->second.d_accum, nn * 4, hipMemcpyDeviceToHost));
    if (n_clips) *n_clips = it->second.clips;
    return 0;
}

int hpfw_gpu_cfg_learn_filters(hpfw_gpu *h, const hpfw_handle_config *c, float *filters_out)
{
    if (!h) return fail(HPFW_E_INVALID, "null handle");
    int rc = cfg_check(c);
    if (rc) return rc;
    const int kt = c->rows * c->context;
    std::vector<float> cov((size_t)kt * kt);
    int64_t clips = 0;
    if ((rc = hpfw_gpu_cfg_cov_get(h, c, cov.data(), &clips))) return rc;
    if (clips == 0) return fail(HPFW_E_INVALID, "no covariance accumulated for this configuration");
    std::vector<float> rows((size_t)c->bits * kt);
    if (hpfw::top_eigenvectors(cov.data(), kt, c->bits, rows.data(), nullptr) != 0) return fail(HPFW_E_INVALID, "eigen-solve failed");
    std::vector<float> colmajor((size_t)c->bits * kt);
    for (int r = 0; r < c->bits; ++r)
        for (int k = 0; k < kt; ++k) colmajor[(size_t)r + (size_t)c->bits * k] = rows[(size_t)r * kt + k];
    if ((rc = hpfw_gpu_cfg_set_filters(h, c, colmajor.data()))) return rc;
    if (filters_out) std::memcpy(filters_out, colmajor.data(), colmajor.size() * 4);
    return 0;
}

int hpfw_gpu_mel_hashprints_pcm16_host(hpfw_gpu *h, const int16_t *pcm, int64_t n_samples, int64_t n_clips, uint16_t *hp,
                                       int64_t hp_stride, int32_t *n_hp)
{
    if (!h || !pcm || !hp || !n_hp || n_clips < 0 || n_samples < 1) return fail(HPFW_E_INVALID, "bad argument");
    const hpfw_handle_config cfg = HPFW_CONFIG_COMBINER;
    const int64_t frames = hpfw::mel_frames(n_samples), nhp_max = frames - cfg.context + 1 - cfg.lag;
    if (nhp_max > hp_stride) return fail(HPFW_E_INVALID, "hp_stride smaller than hpfw_gpu_mel_frames(n_samples) - 81");
    HIP_TRY(hipSetDevice(h->device));
    if (n_clips == 0) return 0;
    if (h->cfg_fpack.find({cfg.rows, cfg.context, cfg.bits}) == h->cfg_fpack.end())
        return fail(HPFW_E_NOFILTERS, "no filters for the combiner configuration: call hpfw_gpu_cfg_set_filters first");
    int16_t *d_pcm = nullptr;
    float *d_s = nullptr;
    int32_t *d_cols = nullptr;
    uint16_t *d_hp = nullptr;
    const size_t per = (size_t)hpfw::kMelBands * frames, hp_words = (size_t)n_clips * std::max<int64_t>(hp_stride, 1);
    int rc = 0;
    if (hipMalloc((void **)&d_pcm, (size_t)n_clips * n_samples * 2) != hipSuccess || hipMalloc((void **)&d_s, (size_t)n_clips * per * 4) != hipSuccess ||
        hipMalloc((void **)&d_cols, (size_t)n_clips * 4) != hipSuccess || hipMalloc((void **)&d_hp, hp_words * 2) != hipSuccess)
        rc = fail(HPFW_E_NOMEM, "hipMalloc failed");
    if (!rc && (hipMemcpy(d_pcm, pcm, (size_t)n_clips * n_samples * 2, hipMemcpyHostToDevice) != hipSuccess ||
                hipMemset(d_s, 0, (size_t)n_clips * per * 4) != hipSuccess || hipMemset(d_hp, 0, hp_words * 2) != hipSuccess))
        rc = fail(HPFW_E_HIP, "H2D copy failed");
    if (!rc) rc = hpfw_gpu_mel_spectrogram_pcm16(h, d_pcm, n_samples, n_clips, d_s, d_cols, nullptr);
    if (!rc) rc = hpfw_gpu_cfg_hashprints(h, &cfg, d_s, d_cols, n_clips, frames, d_hp, hp_stride, nullptr, nullptr);
    std::vector<int32_t> cols((size_t)n_clips);
    if (!rc && (hipDeviceSynchronize() != hipSuccess || hipMemcpy(hp, d_hp, hp_words * 2, hipMemcpyDeviceToHost) != hipSuccess ||
                hipMemcpy(cols.data(), d_cols, (size_t)n_clips * 4, hipMemcpyDeviceToHost) != hipSuccess))
        rc = fail(HPFW_E_HIP, "kernel execution or D2H copy failed");
    for (int64_t i = 0; !rc && i < n_clips; ++i) n_hp[i] = std::max<int32_t>(cols[(size_t)i] - cfg.context + 1 - cfg.lag, 0);
    if (d_pcm) (void)hipFree(d_pcm);
    if (d_s) (void)hipFree(d_s);
    if (d_cols) (void)hipFree(d_cols);
    if (d_hp) (void)hipFree(d_hp);
    return rc;
}

// ---- filter learning: preprocess() of the reference (parallel_collector.h:82-112) ---------------
static int cov_prepare(hpfw_gpu *h, hipStream_t s)
{
    if (!h->d_cov) {
        HIP_TRY(hipMalloc((void **)&h->d_cov, (size_t)hpfw::kFrame * hpfw::kFrame * 4));
        HIP_TRY(hipMemsetAsync(h->d_cov, 0, (size_t)hpfw::kFrame * hpfw::kFrame * 4, s));
        h->cov_files = 0;
    }
    if (!h->d_cov_tiles) {
        std::vector<int> xy((size_t)2 * hpfw::cov_tile_count());
        hpfw::cov_tile_list(xy.data());
        HIP_TRY(hipMalloc((void **)&h->d_cov_tiles, xy.size() * 4));
        HIP_TRY(hipMemcpy(h->d_cov_tiles, xy.data(), xy.size() * 4, hipMemcpyHostToDevice));
    }
    return 0;
}

int hpfw_gpu_cov_reset(hpfw_gpu *h)
{
    if (!h) return fail(HPFW_E_INVALID, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    if (h->d_cov) HIP_TRY(hipMemset(h->d_cov, 0, (size_t)hpfw::kFrame * hpfw::kFrame * 4));
    h->cov_files = 0;
    return 0;
}

int hpfw_gpu_cov_accumulate_db(hpfw_gpu *h, const float *d_db, int64_t n_clips, int64_t c, void *stream)
{
    if (!h || !d_db || n_clips < 0 || c < hpfw::kCtx + 1) return fail(HPFW_E_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    Ordered ordered(h, s);
    int rc = cov_prepare(h, s);
    if (rc) return rc;
    const int64_t chunk = 128; // clips per pass: bounds the workspace (Z, correction vectors, partial sums)
    if ((rc = ensure((void **)&h->d_cov_ws, &h->cov_ws_cap,
                     hpfw::cov_workspace_bytes((int)std::min(chunk, std::max<int64_t>(n_clips, 1)), (int)c))))
        return rc;
    for (int64_t c0 = 0; c0 < n_clips; c0 += chunk) {
        const int nb = (int)std::min(chunk, n_clips - c0);
        hpfw::launch_cov(d_db + c0 * 121 * c, nb, (int)c, h->d_cov_tiles, h->d_cov_ws, h->d_cov, s);
        if ((rc = check_launch("covariance"))) return rc;
    }
    h->cov_files += n_clips;
    return 0;
}

int hpfw_gpu_stage_spectrogram(hpfw_gpu *h, const int16_t *d_pcm, int64_t n_samples, int64_t n_clips, float *d_db,
                               void *stream)
{
    if (!h || !d_pcm || !d_db || n_clips < 0) return fail(HPFW_E_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(h->device));
    DevPlan *dp;
    int rc = get_plan(h, n_samples, &dp);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    Ordered ordered(h, s);
    const int nbmax = pass_clips(h, dp, n_clips);
    if ((rc = ensure_ws(h, dp, nbmax, nbmax))) return rc;
    const size_t per = (size_t)121 * dp->hp.c;
    for (int64_t c0 = 0; c0 < n_clips; c0 += nbmax) {
        const int nb = (int)std::min<int64_t>(nbmax, n_clips - c0);
        if ((rc = run_front(h, dp, d_pcm + c0 * n_samples, nb, 0, true, s))) return rc;
        HIP_TRY(hipMemcpyAsync(d_db + c0 * per, h->ws[2], (size_t)nb * per * 4, hipMemcpyDeviceToDevice, s));
    }
    return 0;
}

int hpfw_gpu_cov_accumulate_pcm16(hpfw_gpu *h, const int16_t *d_pcm, int64_t n_samples, int64_t n_clips,
                                  void *stream)
{
    if (!h || !d_pcm || n_clips < 0) return fail(HPFW_E_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(h->device));
    DevPlan *dp;
    int rc = get_plan(h, n_samples, &dp);
    if (rc) return rc;
    if (dp->hp.n_frames < 2) return fail(HPFW_E_UNSUPPORTED, "clip too short for a covariance");
    hipStream_t s = (hipStream_t)stream;
    Ordered ordered(h, s);
    const int nbmax = pass_clips(h, dp, n_clips);
    if ((rc = ensure_ws(h, dp, nbmax, nbmax))) return rc;
    for (int64_t c0 = 0; c0 < n_clips; c0 += nbmax) {
        const int nb = (int)std::min<int64_t>(nbmax, n_clips - c0);
        if ((rc = run_front(h, dp, d_pcm + c0 * n_samples, nb, 0, true, s))) return rc;
        if ((rc = hpfw_gpu_cov_accumulate_db(h, (const float *)h->ws[2], nb, dp->hp.c, stream))) return rc;
    }
    return 0;
}

int hpfw_gpu_cov_accumulate_pcm16_host(hpfw_gpu *h, const int16_t *pcm, int64_t n_samples, int64_t n_clips)
{
    if (!h || !pcm || n_clips < 0) return fail(HPFW_E_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(h->device));
    if (n_clips == 0) return 0;
    int16_t *d_pcm = nullptr;
    HIP_TRY(hipMalloc((void **)&d_pcm, (size_t)n_clips * n_samples * 2));
    int rc = 0;
    if (hipMemcpy(d_pcm, pcm, (size_t)n_clips * n_samples * 2, hipMemcpyHostToDevice) != hipSuccess)
        rc = fail(HPFW_E_HIP, "H2D copy failed");
    if (!rc) rc = hpfw_gpu_cov_accumulate_pcm16(h, d_pcm, n_samples, n_clips, nullptr);
    if (!rc && hipDeviceSynchronize() != hipSuccess) rc = fail(HPFW_E_HIP, "kernel execution failed");
    (void)hipFree(d_pcm);
    return rc;
}

// full symmetric matrix, row-major 2420 x 2420 (= the column-major Eigen matrix of accum_cov.cereal)
int hpfw_gpu_cov_get(hpfw_gpu *h, float *cov, int64_t *n_files)
{
    if (!h || !cov) return fail(HPFW_E_INVALID, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    const size_t nn = (size_t)hpfw::kFrame * hpfw::kFrame;
    if (!h->d_cov) {
        std::memset(cov, 0, nn * 4);
    } else {
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(cov, h->d_cov, nn * 4, hipMemcpyDeviceToHost));
        // the device holds tiles on or above the diagonal (128-wide); mirror them
        for (int r = 0; r < hpfw::kFrame; ++r)
            for (int c2 = 0; c2 < r; ++c2)
                if (c2 / 128 < r / 128) cov[(size_t)r * hpfw::kFrame + c2] = cov[(size_t)c2 * hpfw::kFrame + r];
    }
    if (n_files) *n_files = h->cov_files;
    return 0;
}

int hpfw_gpu_cov_set(hpfw_gpu *h, const float *cov, int64_t n_files)
{
    if (!h || !cov || n_files < 0) return fail(HPFW_E_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(h->device));
    int rc = cov_prepare(h, nullptr);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(h->d_cov, cov, (size_t)hpfw::kFrame * hpfw::kFrame * 4, hipMemcpyHostToDevice));
    h->cov_files = n_files;
    return 0;
}

int hpfw_gpu_cov_device(hpfw_gpu *h, float **d_cov)
{
    if (!h || !d_cov) return fail(HPFW_E_INVALID, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    int rc = cov_prepare(h, nullptr);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(nullptr));
    *d_cov = h->d_cov;
    return 0;
}

int64_t hpfw_gpu_cov_files(hpfw_gpu *h) { return h ? h->cov_files : 0; }

int hpfw_gpu_cov_set_files(hpfw_gpu *h, int64_t n_files)
{
    if (!h || n_files < 0) return fail(HPFW_E_INVALID, "bad argument");
    h->cov_files = n_files;
    return 0;
}

// calc_filters (hashprint_handle.h:105-112): eigenvectors of the accumulated covariance by descending
// eigenvalue, the first 64 as rows; they become the handle's filters.  filters_out (optional) receives
// them in the reference's column-major layout.
int hpfw_gpu_learn_filters(hpfw_gpu *h, float *filters_out)
{
    if (!h) return fail(HPFW_E_INVALID, "null handle");
    if (!h->d_cov || h->cov_files == 0) return fail(HPFW_E_INVALID, "no covariance accumulated");
    std::vector<float> cov((size_t)hpfw::kFrame * hpfw::kFrame);
    int rc = hpfw_gpu_cov_get(h, cov.data(), nullptr);
    if (rc) return rc;
    std::vector<float> rows((size_t)hpfw::kFilters * hpfw::kFrame);
    if (hpfw::top_eigenvectors(cov.data(), hpfw::kFrame, hpfw::kFilters, rows.data(), nullptr) != 0)
        return fail(HPFW_E_INVALID, "eigen-solve failed");
    std::vector<float> colmajor((size_t)hpfw::kFilters * hpfw::kFrame);
    for (int r = 0; r < hpfw::kFilters; ++r)
        for (int k = 0; k < hpfw::kFrame; ++k) colmajor[(size_t)r + 64 * (size_t)k] = rows[(size_t)r * hpfw::kFrame + k];
    if ((rc = hpfw_gpu_set_filters(h, colmajor.data()))) return rc;
    if (filters_out) std::memcpy(filters_out, colmajor.data(), colmajor.size() * 4);
    return 0;
}

// ---- index + search ----------------------------------------------------------------------------
int hpfw_gpu_index_clear(hpfw_gpu *h)
{
    if (!h) return fail(HPFW_E_INVALID, "null handle");
    h->db_off.assign(1, 0);
    h->db_off_dirty = true;
    return 0;
}

static int index_add_impl(hpfw_gpu *h, const uint64_t *hp, const int64_t *offsets, int64_t n_clips, bool dev,
                          hipStream_t s)
{
    if (!h || !hp || !offsets || n_clips < 0) return fail(HPFW_E_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(h->device));
    for (int64_t i = 0; i < n_clips; ++i)
        if (offsets[i + 1] < offsets[i]) return fail(HPFW_E_INVALID, "offsets must be non-decreasing");
    const int64_t add = offsets[n_clips] - offsets[0];
    const int64_t have = h->db_off.back();
    Ordered ordered(h, s);
    if ((size_t)(have + add) > h->db_cap) {
        size_t ncap = std::max<size_t>((size_t)(have + add), h->db_cap * 2);
        ncap = std::max<size_t>(ncap, 1 << 16);
        uint64_t *nd = nullptr;
        HIP_TRY(hipMalloc((void **)&nd, ncap * 8));
        // earlier appends may still be in flight on a non-blocking stream the null-stream copy below would
        // not wait for, and scans may still be reading the old buffer: growing is rare (capacity doubles)
        HIP_TRY(hipDeviceSynchronize());
        if (have) HIP_TRY(hipMemcpy(nd, h->d_db, (size_t)have * 8, hipMemcpyDeviceToDevice));
        if (h->d_db) HIP_TRY(hipFree(h->d_db));
        h->d_db = nd;
        h->db_cap = ncap;
    }
    if (add) {
        if (dev)
            HIP_TRY(hipMemcpyAsync(h->d_db + have, hp + offsets[0], (size_t)add * 8, hipMemcpyDeviceToDevice, s));
        else
            HIP_TRY(hipMemcpy(h->d_db + have, hp + offsets[0], (size_t)add * 8, hipMemcpyHostToDevice));
    }
    for (int64_t i = 0; i < n_clips; ++i) h->db_off.push_back(have + (offsets[i + 1] - offsets[0]));
    h->db_off_dirty = true;
    return 0;
}

int hpfw_gpu_index_add(hpfw_gpu *h, const uint64_t *hp, const int64_t *offsets, int64_t n_clips)
{
    return index_add_impl(h, hp, offsets, n_clips, false, nullptr);
}

int hpfw_gpu_index_add_device(hpfw_gpu *h, const uint64_t *d_hp, const int64_t *offsets, int64_t n_clips,
                              void *stream)
{
    return index_add_impl(h, d_hp, offsets, n_clips, true, (hipStream_t)stream);
}

int64_t hpfw_gpu_index_size(hpfw_gpu *h) { return h ? (int64_t)h->db_off.size() - 1 : 0; }

// the index back on the host (MemoryStorage::save, storage.h:67-75, dumps the whole db):
// offsets [n_clips + 1] always; hp [offsets[n_clips]] when hp != NULL and hp_cap is large enough
int hpfw_gpu_index_get(hpfw_gpu *h, int64_t *offsets, uint64_t *hp, int64_t hp_cap)
{
    if (!h || !offsets) return fail(HPFW_E_INVALID, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    std::memcpy(offsets, h->db_off.data(), h->db_off.size() * sizeof(int64_t));
    if (!hp) return 0;
    const int64_t total = h->db_off.back();
    if (hp_cap < total) return fail(HPFW_E_INVALID, "hashprint buffer too small for the index");
    if (total) {
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(hp, h->d_db, (size_t)total * 8, hipMemcpyDeviceToHost));
    }
    return 0;
}

int hpfw_gpu_index_set_clip_base(hpfw_gpu *h, uint32_t base)
{
    if (!h) return fail(HPFW_E_INVALID, "null handle");
    h->clip_base = base;
    return 0;
}

int hpfw_gpu_search_topk_device(hpfw_gpu *h, const uint64_t *d_q_hp, const int64_t *q_off, int64_t n_q, int k,
                                hpfw_hit *d_out, void *stream)
{
    if (!h || !q_off || !d_out || n_q < 0 || k < 1 || k > 64) return fail(HPFW_E_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    Ordered ordered(h, s);
    if (n_q == 0) return 0;
    if (!d_q_hp) return fail(HPFW_E_INVALID, "null queries");
    const int64_t n_clips = (int64_t)h->db_off.size() - 1;
    int rc;
    if (n_clips == 0) { // nothing indexed: every slot is "none"
        hpfw::launch_topk(nullptr, (int)n_q, 0, k, h->clip_base, d_out, s);
        return check_launch("topk");
    }
    if (h->db_off_dirty) {
        if ((rc = ensure((void **)&h->d_db_off, &h->db_off_cap, h->db_off.size() * 8))) return rc;
        HIP_TRY(hipMemcpyAsync(h->d_db_off, h->db_off.data(), h->db_off.size() * 8, hipMemcpyHostToDevice, s));
        HIP_TRY(hipStreamSynchronize(s));
        h->db_off_dirty = false;
    }
    int64_t k_max = 0;
    for (int64_t i = 0; i < n_q; ++i) {
        if (q_off[i + 1] < q_off[i]) return fail(HPFW_E_INVALID, "q_off must be non-decreasing");
        k_max = std::max(k_max, q_off[i + 1] - q_off[i]);
    }
    if (k_max > 16000) return fail(HPFW_E_UNSUPPORTED, "query longer than 16000 hashprints");
    if ((rc = ensure((void **)&h->d_q_off, &h->q_off_cap, (size_t)(n_q + 1) * 8))) return rc;
    HIP_TRY(hipMemcpyAsync(h->d_q_off, q_off, (size_t)(n_q + 1) * 8, hipMemcpyHostToDevice, s));
    // queries are processed in groups so the (query, clip) table stays below 1 GiB
    int64_t qgroup = std::max<int64_t>(32, ((int64_t)1 << 27) / n_clips / 32 * 32);
    qgroup = std::min<int64_t>(qgroup, (n_q + 31) / 32 * 32);
    qgroup = std::min<int64_t>(qgroup, (int64_t)65535 * 8 / 32 * 32); // the scans put groups of 8 / 32 queries along gridDim.y
    if ((rc = ensure((void **)&h->d_best, &h->best_cap, (size_t)qgroup * n_clips * 8))) return rc;
    // The scan runs on the matrix cores (k_search_mfma.hip) unless the window does not fit the LDS
    // (queries of several thousand hashprints) or HPFW_SEARCH_POPC asks for the xor/popcount kernel; fewer than
    // 8 queries go one by one through the shifted-rows variant (HPFW_SEARCH_MFMA / HPFW_SEARCH_SHIFT force
    // the grouped / the shifted-rows kernel for any number of queries).
    // A group of 32 queries is one MFMA tile: with fewer than 8 queries most of its rows would be padding
    // and the popcount kernel (one workgroup per 8 queries) does less work.
    const bool mfma = !std::getenv("HPFW_SEARCH_POPC") && hpfw::hamming_mfma_lds_bytes((int)k_max) <= 160 * 1024 &&
                      k_max > 0 && (n_q >= 8 || std::getenv("HPFW_SEARCH_MFMA"));
    const int kt_pad = hpfw::hamming_mfma_kt_pad((int)k_max);
    int64_t n_max = 0;
    for (int64_t i = 0; i < n_clips; ++i) n_max = std::max(n_max, h->db_off[i + 1] - h->db_off[i]);
    if (mfma) {
        if ((rc = ensure((void **)&h->d_qa, &h->qa_cap, (size_t)(qgroup / 32) * kt_pad * 1024))) return rc;
        if ((rc = ensure((void **)&h->d_gk, &h->gk_cap, (size_t)(qgroup / 32) * 8))) return rc;
    }
    std::vector<int> gk;
    for (int64_t g0 = 0; g0 < n_q; g0 += qgroup) {
        const int ng = (int)std::min<int64_t>(qgroup, n_q - g0);
        HIP_TRY(hipMemsetAsync(h->d_best, 0xff, (size_t)ng * n_clips * 8, s));
        hpfw::SearchArgs a;
        a.db = h->d_db;
        a.db_off = h->d_db_off;
        a.n_clips = (int)n_clips;
        a.q = d_q_hp;
        a.q_off = h->d_q_off + g0;
        a.n_q = ng;
        a.k_max = (int)k_max;
        a.best = h->d_best;
        const bool few = (!mfma || std::getenv("HPFW_SEARCH_SHIFT")) && !std::getenv("HPFW_SEARCH_POPC") && k_max > 0 && n_max > 0 &&
                         hpfw::hamming_shift_lds_bytes((int)k_max) <= 160 * 1024;
        if (few) { // a handful of queries: one launch each, the tile rows are shifts of the query
            if ((rc = ensure((void **)&h->d_qa, &h->qa_cap, hpfw::hamming_shift_image_bytes((int)k_max)))) return rc;
            Timed t(h, K_SCAN, s);
            for (int i = 0; i < ng; ++i) {
                const int kq = (int)(q_off[g0 + i + 1] - q_off[g0 + i]);
                if (kq <= 0) continue;
                hpfw::launch_hamming_shift(h->d_db, h->d_db_off, (int)n_clips, (int)std::max<int64_t>(n_max - std::min<int64_t>(kq, n_max) + 1, 1),
                                           d_q_hp + q_off[g0 + i], kq, h->d_qa, h->d_best + (size_t)i * n_clips, s);
            }
        } else if (mfma && n_max > 0) {
            gk.assign((size_t)(ng + 31) / 32 * 2, 0); // per group: longest query, shortest non-empty query
            int kmin_all = 0;
            for (int i = 0; i < ng; ++i) {
                const int kq = (int)(q_off[g0 + i + 1] - q_off[g0 + i]);
                int &mx = gk[(size_t)i / 32 * 2], &mn = gk[(size_t)i / 32 * 2 + 1];
                mx = std::max(mx, kq);
                if (kq > 0) mn = mn == 0 ? kq : std::min(mn, kq);
                if (kq > 0) kmin_all = kmin_all == 0 ? kq : std::min(kmin_all, kq);
            }
            HIP_TRY(hipMemcpyAsync(h->d_gk, gk.data(), gk.size() * 4, hipMemcpyHostToDevice, s));
            HIP_TRY(hipStreamSynchronize(s)); // gk is reused by the next group of queries
            Timed t(h, K_SCAN, s);
            hpfw::launch_expand_queries(d_q_hp, a.q_off, ng, kt_pad, h->d_qa, s);
            // offsets exist up to n_max - (shortest query): that many chunks of workgroups per clip
            hpfw::launch_hamming_mfma(a, h->d_qa, kt_pad, h->d_gk, (int)std::max<int64_t>(n_max - std::min<int64_t>(kmin_all, n_max) + 1, 1), s);
        } else {
            Timed t(h, K_SCAN, s);
            hpfw::launch_hamming_scan(a, s);
        }
        if ((rc = check_launch("hamming_scan"))) return rc;
        {
            Timed t(h, K_TOPK, s);
            if (n_clips >= 16384 && ng <= 64) { // one workgroup per query would crawl through the whole table
                if ((rc = ensure(&h->d_topk_scratch, &h->topk_scratch_cap, hpfw::topk_scratch_bytes(ng, k)))) return rc;
                hpfw::launch_topk_two_step(h->d_best, ng, (int)n_clips, k, h->clip_base, h->d_topk_scratch, d_out + g0 * k, s);
            } else {
                hpfw::launch_topk(h->d_best, ng, (int)n_clips, k, h->clip_base, d_out + g0 * k, s);
            }
        }
        if ((rc = check_launch("topk"))) return rc;
    }
    return 0;
}

int hpfw_gpu_search_topk(hpfw_gpu *h, const uint64_t *q_hp, const int64_t *q_off, int64_t n_q, int k,
                         hpfw_hit *out)
{
    if (!h || !q_off || !out || n_q < 0) return fail(HPFW_E_INVALID, "bad argument");
    if (k < 1 || k > 64) return fail(HPFW_E_INVALID, "k must be in 1..64");
    HIP_TRY(hipSetDevice(h->device));
    if (n_q == 0) return 0;
    const int64_t total = q_off[n_q] - q_off[0];
    uint64_t *d_q = nullptr;
    hpfw_hit *d_out = nullptr;
    HIP_TRY(hipMalloc((void **)&d_q, (size_t)std::max<int64_t>(total, 1) * 8));
    if (hipMalloc((void **)&d_out, (size_t)n_q * k * sizeof(hpfw_hit)) != hipSuccess) {
        (void)hipFree(d_q);
        return fail(HPFW_E_NOMEM, "hipMalloc failed");
    }
    int rc = 0;
    std::vector<int64_t> rel((size_t)n_q + 1);
    for (int64_t i = 0; i <= n_q; ++i) rel[(size_t)i] = q_off[i] - q_off[0];
    if (total && hipMemcpy(d_q, q_hp + q_off[0], (size_t)total * 8, hipMemcpyHostToDevice) != hipSuccess)
        rc = fail(HPFW_E_HIP, "H2D copy failed");
    if (!rc) rc = hpfw_gpu_search_topk_device(h, d_q, rel.data(), n_q, k, d_out, nullptr);
    if (!rc && hipDeviceSynchronize() != hipSuccess) rc = fail(HPFW_E_HIP, "kernel execution failed");
    if (!rc && hipMemcpy(out, d_out, (size_t)n_q * k * sizeof(hpfw_hit), hipMemcpyDeviceToHost) != hipSuccess)
        rc = fail(HPFW_E_HIP, "D2H copy failed");
    (void)hipFree(d_q);
    (void)hipFree(d_out);
    return rc;
}

// ---- voting search (AnnStorage semantics, exact neighbours) ------------------------------------
namespace {
constexpr int kVoteWin = 64, kVoteNn = 5;

// keys [n_win][5] of the windows of all queries, sorted per window; w_first[q] = first window of query q
int knn_windows_impl(hpfw_gpu *h, const uint64_t *q_hp, const int64_t *q_off, int64_t n_q, std::vector<uint64_t> &keys,
                     std::vector<int64_t> &w_first)
{
    HIP_TRY(hipSetDevice(h->device));
    w_first.assign((size_t)n_q + 1, 0);
    std::vector<int64_t> w_start;
    for (int64_t q = 0; q < n_q; ++q) {
        if (q_off[q + 1] < q_off[q]) return fail(HPFW_E_INVALID, "q_off must be non-decreasing");
        const int64_t k = q_off[q + 1] - q_off[q];
        for (int64_t i = 0; i + kVoteWin <= k; ++i) w_start.push_back(q_off[q] - q_off[0] + i);
        w_first[(size_t)q + 1] = (int64_t)w_start.size();
    }
    const int64_t n_win = (int64_t)w_start.size();
    keys.assign((size_t)n_win * kVoteNn, ~0ull);
    const int64_t n_clips = (int64_t)h->db_off.size() - 1;
    int64_t n_max = 0;
    for (int64_t i = 0; i < n_clips; ++i) n_max = std::max(n_max, h->db_off[i + 1] - h->db_off[i]);
    if (n_win == 0 || n_max < kVoteWin) return 0;
    // one launch: groups of 32 windows along gridDim.y (at most 65535)
    if (n_win > (int64_t)65535 * 32) return fail(HPFW_E_UNSUPPORTED, "too many query windows in one call (limit 2097120)");
    int rc;
    Ordered ordered(h, nullptr);
    if (h->db_off_dirty) {
        if ((rc = ensure((void **)&h->d_db_off, &h->db_off_cap, h->db_off.size() * 8))) return rc;
        HIP_TRY(hipMemcpy(h->d_db_off, h->db_off.data(), h->db_off.size() * 8, hipMemcpyHostToDevice));
        h->db_off_dirty = false;
    }
    const int64_t total = q_off[n_q] - q_off[0];
    const int kt_pad = hpfw::hamming_mfma_kt_pad(kVoteWin);
    uint64_t *d_q = nullptr, *d_slots = nullptr;
    int64_t *d_ws = nullptr;
    void *d_qa = nullptr;
    const size_t n_groups = (size_t)(n_win + 31) / 32;
    if (hipMalloc((void **)&d_q, (size_t)total * 8) != hipSuccess || hipMalloc((void **)&d_ws, (size_t)n_win * 8) != hipSuccess ||
        hipMalloc((void **)&d_slots, (size_t)n_win * 64) != hipSuccess || hipMalloc(&d_qa, n_groups * kt_pad * 1024) != hipSuccess)
        rc = fail(HPFW_E_NOMEM, "hipMalloc failed");
    else
        rc = 0;
    if (!rc && (hipMemcpy(d_q, q_hp + q_off[0], (size_t)total * 8, hipMemcpyHostToDevice) != hipSuccess ||
                hipMemcpy(d_ws, w_start.data(), (size_t)n_win * 8, hipMemcpyHostToDevice) != hipSuccess ||
                hipMemset(d_slots, 0xff, (size_t)n_win * 64) != hipSuccess))
        rc = fail(HPFW_E_HIP, "H2D copy failed");
    if (!rc) {
        hpfw::launch_expand_windows(d_q, d_ws, (int)n_win, kVoteWin, kt_pad, d_qa, nullptr);
        hpfw::launch_knn_windows(h->d_db, h->d_db_off, (int)n_clips, (int)(n_max - kVoteWin + 1), d_qa, kt_pad, (int)n_win,
                                 kVoteWin, kVoteNn, d_slots, nullptr);
        rc = check_launch("knn_windows");
    }
    std::vector<uint64_t> slots((size_t)n_win * 8);
    if (!rc && hipMemcpy(slots.data(), d_slots, slots.size() * 8, hipMemcpyDeviceToHost) != hipSuccess)
        rc = fail(HPFW_E_HIP, "kernel execution failed");
    if (d_q) (void)hipFree(d_q);
    if (d_ws) (void)hipFree(d_ws);
    if (d_slots) (void)hipFree(d_slots);
    if (d_qa) (void)hipFree(d_qa);
    if (rc) return rc;
    for (int64_t w = 0; w < n_win; ++w) { // the device keeps the 5 smallest keys unsorted
        uint64_t *s5 = &slots[(size_t)w * 8];
        std::sort(s5, s5 + kVoteNn);
        for (int r = 0; r < kVoteNn; ++r) keys[(size_t)w * kVoteNn + r] = s5[r];
    }
    return 0;
}
} // namespace

int hpfw_gpu_knn_windows(hpfw_gpu *h, const uint64_t *q_hp, const int64_t *q_off, int64_t n_q, uint64_t *keys,
                         int64_t keys_cap)
{
    if (!h || !q_hp || !q_off || !keys || n_q < 0) return fail(HPFW_E_INVALID, "bad argument");
    std::vector<uint64_t> k;
    std::vector<int64_t> wf;
    int rc = knn_windows_impl(h, q_hp, q_off, n_q, k, wf);
    if (rc) return rc;
    if ((int64_t)k.size() > keys_cap) return fail(HPFW_E_INVALID, "keys buffer too small");
    std::memcpy(keys, k.data(), k.size() * 8);
    return 0;
}

int hpfw_gpu_search_votes(hpfw_gpu *h, const uint64_t *q_hp, const int64_t *q_off, int64_t n_q, hpfw_vote *out)
{
    if (!h || !q_hp || !q_off || !out || n_q < 0) return fail(HPFW_E_INVALID, "bad argument");
    std::vector<uint64_t> keys;
    std::vector<int64_t> wf;
    int rc = knn_windows_impl(h, q_hp, q_off, n_q, keys, wf);
    if (rc) return rc;
    struct Bucket {
        int64_t clip, off;
        float cnt;
    };
    std::vector<Bucket> buckets;
    for (int64_t q = 0; q < n_q; ++q) {
        hpfw_vote best = {0xffffffffu, 0, 0, 0.0f, 0.0f}; // annoy_storage.h:43
        buckets.clear();
        for (int64_t w = wf[(size_t)q]; w < wf[(size_t)q + 1]; ++w) {
            const int64_t i = w - wf[(size_t)q];
            for (int r = 0; r < kVoteNn; ++r) {
                const uint64_t key = keys[(size_t)w * kVoteNn + r];
                if (key == ~0ull) continue;
                const uint64_t d = key >> 40;
                const int64_t pos = (int64_t)(key & (((uint64_t)1 << 40) - 1));
                const int64_t clip = (int64_t)(std::upper_bound(h->db_off.begin(), h->db_off.end(), pos) - h->db_off.begin()) - 1;
                const int64_t off = i - (pos - h->db_off[(size_t)clip]);
                size_t s = 0;
                while (s < buckets.size() && !(buckets[s].clip == clip && buckets[s].off == off)) ++s;
                if (s == buckets.size()) buckets.push_back({clip, off, 0.0f});
                buckets[s].cnt = (float)((double)buckets[s].cnt + 1.0 / (double)(float)(d + 1)); // :53
                if (buckets[s].cnt > best.cnt) {                                                  // :55-59
                    best.clip = h->clip_base + (uint32_t)clip;
                    best.offset = off;
                    best.cnt = buckets[s].cnt;
                }
            }
        }
        out[q] = best;
    }
    return 0;
}

int hpfw_gpu_merge_topk(const hpfw_hit *in, int n_shards, int64_t n_q, int k, hpfw_hit *out)
{
    if (!in || !out || n_shards < 1 || n_q < 0 || k < 1) return fail(HPFW_E_INVALID, "bad argument");
    std::vector<hpfw_hit> all((size_t)n_shards * k);
    for (int64_t q = 0; q < n_q; ++q) {
        for (int s = 0; s < n_shards; ++s)
            for (int t = 0; t < k; ++t) all[(size_t)s * k + t] = in[((size_t)s * n_q + q) * k + t];
        std::stable_sort(all.begin(), all.end(), [](const hpfw_hit &a, const hpfw_hit &b) {
            if (a.dist != b.dist) return a.dist < b.dist;
            return a.clip < b.clip;
        });
        for (int t = 0; t < k; ++t) out[(size_t)q * k + t] = all[(size_t)t];
    }
    return 0;
}

// ---- timing ------------------------------------------------------------------------------------
int hpfw_gpu_timer_start(hpfw_gpu *h, void *stream)
{
    if (!h) return fail(HPFW_E_INVALID, "null handle");
    HIP_TRY(hipEventRecord(h->ev0, (hipStream_t)stream));
    return 0;
}

int hpfw_gpu_timer_stop(hpfw_gpu *h, void *stream, float *ms)
{
    if (!h || !ms) return fail(HPFW_E_INVALID, "null argument");
    HIP_TRY(hipEventRecord(h->ev1, (hipStream_t)stream));
    HIP_TRY(hipEventSynchronize(h->ev1));
    HIP_TRY(hipEventElapsedTime(ms, h->ev0, h->ev1));
    return 0;
}

int hpfw_gpu_set_kernel_timing(hpfw_gpu *h, int mask)
{
    if (!h) return fail(HPFW_E_INVALID, "null handle");
    h->timing_mask = (unsigned)mask;
    for (auto &t : h->timed) h->ev_pool.push_back({t.a, t.b});
    h->timed.clear();
    std::memset(h->k_ms, 0, sizeof(h->k_ms));
    std::memset(h->k_launches, 0, sizeof(h->k_launches));
    return 0;
}

int hpfw_gpu_get_kernel_timing(hpfw_gpu *h, const char **names, float *ms, int *launches, int *n)
{
    if (!h || !names || !ms || !launches || !n) return fail(HPFW_E_INVALID, "null argument");
    for (auto &t : h->timed) {
        HIP_TRY(hipEventSynchronize(t.b));
        float e = 0.0f;
        HIP_TRY(hipEventElapsedTime(&e, t.a, t.b));
        h->k_ms[t.kind] += e;
        h->k_launches[t.kind] += 1;
        h->ev_pool.push_back({t.a, t.b});
    }
    h->timed.clear();
    const int cap = *n;
    int w = 0;
    for (int i = 0; i < K_COUNT && w < cap; ++i, ++w) {
        names[w] = kKernelNames[i];
        ms[w] = h->k_ms[i];
        launches[w] = h->k_launches[i];
    }
    *n = w;
    return 0;
}

} // extern "C"
