// legacy.cpp -- the eight extern "C" symbols of the reference's only FFI boundary
// (modules/python/parallel_collector_wrapper.hpp:21-38, implementation wrapper.cpp:5-62), kept
// source-compatible and routed to the GPU path.  Ownership as in the reference: results are
// allocated with new[] and released by the paired *_free function.  Unlike the reference
// (wrapper.cpp has no exception barrier) nothing throws through the C boundary: failures return
// NULL / *got = 0 and leave a message in hpfw_gpu_last_error().
//
// Audio input: essentia MonoLoader (reference include/hpfw/spectrum/cqt.h:45-52) is replaced by a
// RIFF/WAVE reader for PCM16 at 44.1 kHz (mono, or stereo averaged as MonoLoader's "mix" downmix);
// other containers / rates are rejected (decode and resampling are outside the accelerated path).
// Filters: read from / written to <cache>/filters.cereal in cereal's binary layout of an Eigen
// matrix (reference include/hpfw/utils.h:84-90: int32 rows, int32 cols, column-major payload).
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "../../include/hpfw_gpu.h"

extern "C" void hpfw_internal_set_error(const char *msg); // api.hip: feeds hpfw_gpu_last_error()

namespace {

bool read_wav_pcm16_mono(const std::string &path, std::vector<int16_t> &out, std::string &why)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) {
        why = "cannot open " + path;
        return false;
    }
    char hdr[12];
    f.read(hdr, 12);
    if (!f || std::memcmp(hdr, "RIFF", 4) || std::memcmp(hdr + 8, "WAVE", 4)) {
        why = path + ": not a RIFF/WAVE file";
        return false;
    }
    uint16_t fmt = 0, channels = 0, bits = 0;
    uint32_t rate = 0;
    bool have_fmt = false;
    while (f) {
        char id[4];
        uint32_t sz = 0;
        f.read(id, 4);
        f.read(reinterpret_cast<char *>(&sz), 4);
        if (!f) break;
        if (!std::memcmp(id, "fmt ", 4)) {
            if (sz < 16 || sz > 4096) break; // malformed: reported as "no data chunk" below
            std::vector<char> b(sz);
            f.read(b.data(), sz);
            std::memcpy(&fmt, b.data(), 2);
            std::memcpy(&channels, b.data() + 2, 2);
            std::memcpy(&rate, b.data() + 4, 4);
            std::memcpy(&bits, b.data() + 14, 2);
            if (fmt == 0xFFFE && sz >= 26) std::memcpy(&fmt, b.data() + 24, 2); // WAVE_FORMAT_EXTENSIBLE: the sub-format's tag
            if (sz & 1) f.seekg(1, std::ios::cur);
            have_fmt = true;
        } else if (!std::memcmp(id, "data", 4)) {
            if (!have_fmt || fmt != 1 || bits != 16 || (channels != 1 && channels != 2) || rate != 44100) {
                why = path + ": only PCM16 mono/stereo at 44100 Hz is supported";
                return false;
            }
            // a streamed file may carry 0 or 0xffffffff as the size: trust the file's length instead
            const std::streamoff here = f.tellg();
            f.seekg(0, std::ios::end);
            const std::streamoff left = f.tellg() - here;
            f.seekg(here);
            if (sz == 0 || (std::streamoff)sz > left) sz = (uint32_t)std::min<std::streamoff>(left, 0xfffffffe);
            std::vector<int16_t> raw(sz / 2);
            f.read(reinterpret_cast<char *>(raw.data()), (std::streamsize)raw.size() * 2);
            raw.resize((size_t)f.gcount() / 2);
            if (channels == 1) {
                out.swap(raw);
            } else {
                out.resize(raw.size() / 2);
                for (size_t i = 0; i < out.size(); ++i) out[i] = (int16_t)(((int)raw[2 * i] + (int)raw[2 * i + 1]) / 2);
            }
            return true;
        } else {
            f.seekg(sz + (sz & 1), std::ios::cur);
        }
    }
    why = path + ": no data chunk";
    return false;
}

bool load_filters_cereal(const std::string &path, std::vector<float> &f)
{
    std::ifstream is(path, std::ios::binary);
    if (!is) return false;
    int32_t rows = 0, cols = 0;
    is.read(reinterpret_cast<char *>(&rows), 4);
    is.read(reinterpret_cast<char *>(&cols), 4);
    if (!is || rows != HPFW_FILTERS || cols != HPFW_FRAME_SIZE) return false;
    f.resize((size_t)rows * cols);
    is.read(reinterpret_cast<char *>(f.data()), (std::streamsize)f.size() * 4);
    return (bool)is;
}

bool save_filters_cereal(const std::string &path, const std::vector<float> &f)
{
    std::ofstream os(path, std::ios::binary);
    if (!os) return false;
    const int32_t rows = HPFW_FILTERS, cols = HPFW_FRAME_SIZE;
    os.write(reinterpret_cast<const char *>(&rows), 4);
    os.write(reinterpret_cast<const char *>(&cols), 4);
    os.write(reinterpret_cast<const char *>(f.data()), (std::streamsize)f.size() * 4);
    return (bool)os;
}

} // namespace

struct hpfw_legacy_collector {
    hpfw_gpu *gpu = nullptr;
    std::string cache_dir = "cache/"; // parallel_collector.h:38
    std::vector<float> filters;
};

// the exported entry points: no exception leaves the C boundary (the reference's wrapper.cpp has no such
// barrier); a failure that surfaces as one -- out of host memory, in practice -- becomes NULL + a message
template <class F>
static auto guarded(F f) -> decltype(f())
{
    try {
        return f();
    } catch (const std::exception &e) {
        hpfw_internal_set_error((std::string("host failure: ") + e.what()).c_str());
    } catch (...) {
        hpfw_internal_set_error("host failure");
    }
    return nullptr;
}

extern "C" {

hpfw_legacy_collector *par_collector_new(void)
{
    auto *c = new hpfw_legacy_collector();
    int dev = 0;
    if (const char *e = std::getenv("HPFW_GPU_DEVICE")) dev = std::atoi(e);
    if (hpfw_gpu_create(dev, &c->gpu) != 0) {
        delete c;
        return nullptr;
    }
    return c;
}

void par_collector_del(hpfw_legacy_collector *c)
{
    if (!c) return;
    hpfw_gpu_destroy(c->gpu);
    delete c;
}

// ParallelCollector::load (parallel_collector.h:68-73); like the reference wrapper the argument
// is optional: NULL or "" means the default "cache/" directory.
void par_collector_load(hpfw_legacy_collector *c, const char *cache)
{
    if (!c) return;
    if (cache && *cache) {
        c->cache_dir = cache;
        if (c->cache_dir.back() != '/') c->cache_dir += '/';
    }
    std::vector<float> f;
    if (load_filters_cereal(c->cache_dir + "filters.cereal", f)) { // missing file: silent no-op, cache.h:77-79
        c->filters.swap(f);
        (void)hpfw_gpu_set_filters(c->gpu, c->filters.data());
    }
    // accum_cov.cereal (cache.h:34-36): int32 2420, int32 2420, 2420^2 floats (symmetric, so the
    // column-major payload is also row-major); the reference keeps accumulating across runs
    std::ifstream is(c->cache_dir + "accum_cov.cereal", std::ios::binary);
    int32_t rows = 0, cols = 0;
    if (is && is.read(reinterpret_cast<char *>(&rows), 4) && is.read(reinterpret_cast<char *>(&cols), 4) &&
        rows == HPFW_FRAME_SIZE && cols == HPFW_FRAME_SIZE) {
        std::vector<float> cov((size_t)rows * cols);
        if (is.read(reinterpret_cast<char *>(cov.data()), (std::streamsize)cov.size() * 4))
            (void)hpfw_gpu_cov_set(c->gpu, cov.data(), 1);
    }
}

void par_collector_save(hpfw_legacy_collector *c, const char *cache)
{
    if (!c || c->filters.empty()) return;
    if (cache && *cache) {
        c->cache_dir = cache;
        if (c->cache_dir.back() != '/') c->cache_dir += '/';
    }
    std::error_code ec;
    std::filesystem::create_directories(c->cache_dir, ec);
    (void)save_filters_cereal(c->cache_dir + "filters.cereal", c->filters);
    std::vector<float> cov((size_t)HPFW_FRAME_SIZE * HPFW_FRAME_SIZE);
    int64_t n_files = 0;
    if (hpfw_gpu_cov_get(c->gpu, cov.data(), &n_files) == 0 && n_files > 0) { // cache.h:34-36
        std::ofstream os(c->cache_dir + "accum_cov.cereal", std::ios::binary);
        const int32_t dim = HPFW_FRAME_SIZE;
        os.write(reinterpret_cast<const char *>(&dim), 4);
        os.write(reinterpret_cast<const char *>(&dim), 4);
        os.write(reinterpret_cast<const char *>(cov.data()), (std::streamsize)cov.size() * 4);
    }
}

// A file of any length: the PCM is padded with zeros up to the next supported (7-smooth) length --
// at most 0.8 % more samples; the reference transforms the exact length with FFTW, so this is a
// deviation, confined to the file entry points and switched off by HPFW_STRICT_LENGTH.
static bool read_clip(const std::string &path, std::vector<int16_t> &pcm, std::string &why)
{
    try { // nothing may throw through the C boundary (or out of a reader thread): e.g. bad_alloc on a huge file
        if (!read_wav_pcm16_mono(path, pcm, why)) return false;
    } catch (const std::exception &e) {
        why = path + ": " + e.what();
        return false;
    }
    if (std::getenv("HPFW_STRICT_LENGTH")) return true;
    const int64_t want = hpfw_gpu_supported_length((int64_t)pcm.size());
    if (want > (int64_t)pcm.size()) pcm.resize((size_t)want, 0);
    return true;
}

static uint64_t *calc_hashprint_impl(hpfw_legacy_collector *c, const char *filename, int *size)
{
    if (size) *size = 0;
    if (!c || !filename || !size) return nullptr;
    std::vector<int16_t> pcm;
    std::string why;
    if (!read_clip(filename, pcm, why)) {
        hpfw_internal_set_error(why.c_str());
        return nullptr;
    }
    hpfw_geometry g;
    if (hpfw_gpu_geometry(c->gpu, (int64_t)pcm.size(), &g) != 0) return nullptr;
    if (g.n_hp <= 0) {
        hpfw_internal_set_error((std::string(filename) + ": clip too short to yield a hashprint").c_str());
        return nullptr;
    }
    auto *hp = new uint64_t[(size_t)g.n_hp];
    if (hpfw_gpu_extract_pcm16_host(c->gpu, pcm.data(), (int64_t)pcm.size(), 1, hp) != 0) {
        delete[] hp;
        return nullptr;
    }
    *size = (int)g.n_hp;
    return hp;
}

void calc_hashprint_result_free(uint64_t *hp) { delete[] hp; }

// ---- prepare(): files in windows, equally long clips batched through the device stages ----------
namespace {

struct Loaded {
    std::vector<int16_t> pcm;
    bool ok = false;
};

// the reference reads its files on a taskflow pool (parallel_collector.h:95-108); here a team of host
// threads decodes one window of files while the device stages take whole groups of clips
void read_window(const char **filenames, const std::vector<int> &files, size_t first, size_t last, std::vector<Loaded> &out)
{
    std::mutex why_mtx;
    std::string first_why; // the reader threads' messages are thread-local: keep the first failure for the caller
    const int count = (int)(last - first);
    out.assign((size_t)count, Loaded());
    unsigned team = std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
    team = std::min<unsigned>(team, (unsigned)count);
    std::atomic<int> next{0};
    auto work = [&] {
        for (int i; (i = next.fetch_add(1)) < count;) {
            std::string why;
            out[(size_t)i].ok = read_clip(filenames[files[first + (size_t)i]], out[(size_t)i].pcm, why);
            if (!out[(size_t)i].ok) {
                std::scoped_lock lock(why_mtx);
                if (first_why.empty()) first_why = why;
            }
        }
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < team; ++t) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
    if (!first_why.empty()) hpfw_internal_set_error(first_why.c_str()); // skipped files (parallel_collector.h:101-103): the message survives
}

struct DevMem {
    void *p = nullptr;
    DevMem() = default;
    DevMem(const DevMem &) = delete;
    DevMem &operator=(const DevMem &) = delete;
    ~DevMem()
    {
        if (p) (void)hipFree(p);
    }
    bool alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1) == hipSuccess; }
};

// dB spectrograms [n][121][C] of n equally long clips of the window (device memory, caller frees)
float *group_spectrograms(hpfw_gpu *gpu, const std::vector<Loaded> &clips, const std::vector<int> &pos, int64_t len,
                          const hpfw_geometry &g)
{
    DevMem pcm;
    float *d_db = nullptr;
    if (!pcm.alloc(pos.size() * (size_t)len * 2) || hipMalloc((void **)&d_db, pos.size() * (size_t)121 * g.c * 4) != hipSuccess) {
        hpfw_internal_set_error("prepare: out of device memory");
        return nullptr;
    }
    bool ok = true;
    for (size_t k = 0; k < pos.size() && ok; ++k)
        ok = hipMemcpy((int16_t *)pcm.p + k * (size_t)len, clips[(size_t)pos[k]].pcm.data(), (size_t)len * 2, hipMemcpyHostToDevice) ==
             hipSuccess;
    ok = ok && hpfw_gpu_stage_spectrogram(gpu, (const int16_t *)pcm.p, len, (int64_t)pos.size(), d_db, nullptr) == 0 &&
         hipDeviceSynchronize() == hipSuccess;
    if (!ok) {
        (void)hipFree(d_db);
        return nullptr;
    }
    return d_db;
}

// hashprints of n clips from their dB spectrograms: out[k] = new uint64_t[g.n_hp]
bool group_hashprints(hpfw_gpu *gpu, const float *d_db, size_t n, const hpfw_geometry &g, uint64_t **out)
{
    DevMem proj, hp;
    if (!proj.alloc(n * 64 * (size_t)g.n_frames * 4) || !hp.alloc(n * (size_t)g.n_hp * 8)) {
        hpfw_internal_set_error("prepare: out of device memory");
        return false;
    }
    if (hpfw_gpu_stage_project(gpu, d_db, (int64_t)n, g.c, (float *)proj.p, nullptr) != 0 ||
        hpfw_gpu_stage_pack(gpu, (const float *)proj.p, (int64_t)n, g.n_frames, (uint64_t *)hp.p, nullptr) != 0)
        return false;
    std::vector<uint64_t> host(n * (size_t)g.n_hp);
    if (hipMemcpy(host.data(), hp.p, host.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) {
        hpfw_internal_set_error("prepare: D2H copy failed");
        return false;
    }
    for (size_t k = 0; k < n; ++k) {
        out[k] = new uint64_t[(size_t)g.n_hp];
        std::memcpy(out[k], host.data() + k * (size_t)g.n_hp, (size_t)g.n_hp * 8);
    }
    return true;
}

struct KeptGroup {
    std::vector<int> files; // indices into the caller's list
    hpfw_geometry g;
    float *d_db;
};

} // namespace

// ParallelCollector::prepare (parallel_collector.h:48-52): preprocess (:82-112) adds every file's
// frame covariance to accum_cov, takes the 64 leading eigenvectors as the new filters and saves them;
// collect_fingerprints (:114-137) then turns the cached spectrograms into hashprints.  Here the
// spectrograms stay in device memory between the two steps (up to HPFW_PREPARE_KEEP_GB, default 32;
// files beyond that are read and transformed again), the files are read in windows by a team of
// host threads, and clips of equal length go through the device stages together.  Per-file errors are
// skipped as the reference does (:101-103), so *got may be < n; results keep the input order; the name
// is the stem of the path (:123,129).  HPFW_PREPARE_KEEP_FILTERS=1 skips the learning step and keeps
// the filters that load() / a previous prepare() installed.
static bool collect_files(hpfw_legacy_collector *c, const char **filenames, int n, bool learn, std::vector<uint64_t *> &hp,
                          std::vector<int> &hp_size)
{
    size_t keep_budget = (size_t)32 << 30;
    if (const char *e = std::getenv("HPFW_PREPARE_KEEP_GB")) keep_budget = (size_t)std::max(0.0, std::atof(e) * 1073741824.0);
    hp.assign((size_t)n, nullptr);
    hp_size.assign((size_t)n, 0);
    std::vector<KeptGroup> kept;
    std::vector<int> again; // files whose spectrogram could not be kept
    size_t kept_bytes = 0;
    int64_t used = 0;

    // one pass over `files`: first = covariance (+ keep the spectrograms); otherwise hashprints at once
    auto pass = [&](const std::vector<int> &files, bool first) {
        size_t at = 0;
        while (at < files.size()) {
            size_t end = at;
            uintmax_t bytes = 0; // a window: at most 256 files and about 1 GiB of audio
            while (end < files.size() && end - at < 256 && (end == at || bytes < ((uintmax_t)1 << 30))) {
                std::error_code ec;
                const uintmax_t sz = std::filesystem::file_size(filenames[files[end]], ec);
                if (!ec) bytes += sz;
                ++end;
            }
            std::vector<Loaded> clips;
            read_window(filenames, files, at, end, clips);
            std::map<int64_t, std::vector<int>> by_len; // length -> positions in the window, in input order
            for (size_t i = 0; i < clips.size(); ++i)
                if (clips[i].ok) by_len[(int64_t)clips[i].pcm.size()].push_back((int)i);
            for (auto &[len, pos] : by_len) {
                hpfw_geometry g;
                if (hpfw_gpu_geometry(c->gpu, len, &g) != 0 || g.n_frames < 2) continue; // skipped
                float *d_db = group_spectrograms(c->gpu, clips, pos, len, g);
                if (!d_db) continue;
                std::vector<int> ids;
                for (int q : pos) ids.push_back(files[at + (size_t)q]);
                const size_t sz = pos.size() * (size_t)121 * g.c * 4;
                if (first && learn) {
                    if (hpfw_gpu_cov_accumulate_db(c->gpu, d_db, (int64_t)pos.size(), g.c, nullptr) == 0) used += (int64_t)pos.size();
                    if (g.n_hp > 0 && kept_bytes + sz <= keep_budget) {
                        kept.push_back(KeptGroup{ids, g, d_db});
                        kept_bytes += sz;
                        continue;
                    }
                    if (g.n_hp > 0) again.insert(again.end(), ids.begin(), ids.end());
                } else if (g.n_hp > 0) {
                    std::vector<uint64_t *> out(pos.size(), nullptr);
                    if (group_hashprints(c->gpu, d_db, pos.size(), g, out.data()))
                        for (size_t k = 0; k < ids.size(); ++k) {
                            hp[(size_t)ids[k]] = out[k];
                            hp_size[(size_t)ids[k]] = (int)g.n_hp;
                        }
                }
                (void)hipDeviceSynchronize();
                (void)hipFree(d_db);
            }
            at = end;
        }
    };

    std::vector<int> all((size_t)n);
    for (int i = 0; i < n; ++i) all[(size_t)i] = i;
    pass(all, true);
    bool failed = false;
    if (learn) {
        if (used > 0) {
            c->filters.assign((size_t)HPFW_FILTERS * HPFW_FRAME_SIZE, 0.0f);
            if (hpfw_gpu_learn_filters(c->gpu, c->filters.data()) != 0) {
                c->filters.clear();
                failed = true;
            } else {
                par_collector_save(c, nullptr);
            }
        }
        for (KeptGroup &k : kept) {
            std::vector<uint64_t *> out(k.files.size(), nullptr);
            if (!failed && group_hashprints(c->gpu, k.d_db, k.files.size(), k.g, out.data()))
                for (size_t q = 0; q < k.files.size(); ++q) {
                    hp[(size_t)k.files[q]] = out[q];
                    hp_size[(size_t)k.files[q]] = (int)k.g.n_hp;
                }
            (void)hipFree(k.d_db);
        }
        if (!failed && !again.empty()) {
            std::sort(again.begin(), again.end());
            pass(again, false);
        }
    }
    return !failed;
}

static FilenameHashprintPair *prepare_impl(hpfw_legacy_collector *c, const char **filenames, int n, int *got)
{
    if (got) *got = 0;
    if (!c || !filenames || n < 0 || !got) return nullptr;
    const bool learn = !(std::getenv("HPFW_PREPARE_KEEP_FILTERS") && !c->filters.empty());
    std::vector<uint64_t *> hp;
    std::vector<int> hp_size;
    if (!collect_files(c, filenames, n, learn, hp, hp_size)) return nullptr;
    auto *res = new FilenameHashprintPair[(size_t)(n > 0 ? n : 1)];
    int w = 0;
    for (int i = 0; i < n; ++i) {
        if (!hp[(size_t)i]) continue;
        const std::string stem = std::filesystem::path(filenames[i]).stem().string();
        res[w].filename = new char[stem.size() + 1];
        std::memcpy(res[w].filename, stem.c_str(), stem.size() + 1);
        res[w].hashprint = hp[(size_t)i];
        res[w].hp_size = hp_size[(size_t)i];
        ++w;
    }
    *got = w;
    return res;
}

// calc_hashprint (parallel_collector.h:54-59) for a list of files in one call, as LiveSongIdentification::search
// needs it for its queries (live_song_id.h:37-41): the files are read and transformed in batches as in
// prepare(), nothing is learned.  Returns n entries in input order, released with prepare_result_free(res, n);
// an entry whose file failed has hashprint == NULL and hp_size == 0.  NULL when no filters are loaded.
static FilenameHashprintPair *calc_hashprints_impl(hpfw_legacy_collector *c, const char **filenames, int n)
{
    if (!c || !filenames || n < 0) return nullptr;
    if (c->filters.empty()) {
        hpfw_internal_set_error("no filters loaded: call par_collector_load or par_collector_prepare first");
        return nullptr;
    }
    std::vector<uint64_t *> hp;
    std::vector<int> hp_size;
    if (!collect_files(c, filenames, n, false, hp, hp_size)) return nullptr;
    auto *res = new FilenameHashprintPair[(size_t)(n > 0 ? n : 1)];
    for (int i = 0; i < n; ++i) {
        const std::string stem = std::filesystem::path(filenames[i]).stem().string();
        res[i].filename = new char[stem.size() + 1];
        std::memcpy(res[i].filename, stem.c_str(), stem.size() + 1);
        res[i].hashprint = hp[(size_t)i];
        res[i].hp_size = hp_size[(size_t)i];
    }
    return res;
}

uint64_t *par_collector_calc_hashprint(hpfw_legacy_collector *c, const char *filename, int *size)
{
    return guarded([&] { return calc_hashprint_impl(c, filename, size); });
}

FilenameHashprintPair *par_collector_prepare(hpfw_legacy_collector *c, const char **filenames, int n, int *got)
{
    FilenameHashprintPair *res = guarded([&] { return prepare_impl(c, filenames, n, got); });
    if (!res && got) *got = 0;
    return res;
}

FilenameHashprintPair *par_collector_calc_hashprints(hpfw_legacy_collector *c, const char **filenames, int n)
{
    return guarded([&] { return calc_hashprints_impl(c, filenames, n); });
}

void prepare_result_free(FilenameHashprintPair *res, int got)
{
    if (!res) return;
    for (int i = 0; i < got; ++i) {
        delete[] res[i].filename;
        delete[] res[i].hashprint;
    }
    delete[] res;
}

} // extern "C"
