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
#include <chrono>
#include <cstdio>
#include <cstring>
#include <filesystem>
#include <future>
#include <fstream>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_set>
#include <vector>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <hip/hip_runtime_api.h>

#include "../../include/hpfw_gpu.h"
#include "legacy_internal.h" // the extern "C" signatures multi.cpp sees, checked against the definitions below

extern "C" void hpfw_internal_set_error(const char *msg); // api.hip: feeds hpfw_gpu_last_error()
extern "C" void hpfw_internal_note_idle(hpfw_gpu *h);      // api.hip: every plan used so far is idle (evictable without a device wait)

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

// cache/spectros/<stem> (cache.h:30-33): cereal's binary image of Spectrogram = Eigen::Matrix<float, 121, Dynamic>
// (utils.h:84-90: int32 rows, int32 cols, column-major payload: element (b, c) at b + 121 c).  The device
// layout is bin-major [121][C]: transposed on the way out and in.
bool save_spectro_cereal(const std::string &path, const float *binmajor, int32_t cols)
{
    std::vector<float> cm((size_t)HPFW_BINS * cols);
    for (int32_t b = 0; b < HPFW_BINS; ++b)
        for (int32_t c = 0; c < cols; ++c) cm[(size_t)c * HPFW_BINS + b] = binmajor[(size_t)b * cols + c];
    std::ofstream os(path, std::ios::binary);
    if (!os) return false;
    const int32_t rows = HPFW_BINS;
    os.write(reinterpret_cast<const char *>(&rows), 4);
    os.write(reinterpret_cast<const char *>(&cols), 4);
    os.write(reinterpret_cast<const char *>(cm.data()), (std::streamsize)cm.size() * 4);
    return (bool)os;
}

// the matrix as stored (column-major); false when the file is not a 121-row float matrix of that size
bool load_spectro_cereal(const std::string &path, std::vector<float> &colmajor, int32_t &cols)
{
    std::ifstream is(path, std::ios::binary);
    if (!is) return false;
    int32_t rows = 0;
    cols = 0;
    is.read(reinterpret_cast<char *>(&rows), 4);
    is.read(reinterpret_cast<char *>(&cols), 4);
    if (!is || rows != HPFW_BINS || cols < 0 || cols > (1 << 24)) return false;
    std::error_code ec;
    if (std::filesystem::file_size(path, ec) != (uintmax_t)8 + (uintmax_t)rows * cols * 4 || ec) return false;
    colmajor.resize((size_t)rows * cols);
    is.read(reinterpret_cast<char *>(colmajor.data()), (std::streamsize)colmajor.size() * 4);
    return (bool)is;
}

// fn(i) for i in [0, n) on a small team of host threads
template <class F>
void host_team(int n, F fn)
{
    unsigned team = std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
    team = std::min<unsigned>(team, (unsigned)std::max(n, 1));
    std::atomic<int> next{0};
    auto work = [&] {
        for (int i; (i = next.fetch_add(1)) < n;) fn(i);
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < team; ++t) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
}

} // namespace

struct hpfw_legacy_collector {
    hpfw_gpu *gpu = nullptr;
    std::string cache_dir = "cache/"; // parallel_collector.h:38
    std::vector<float> filters;
    // prepare() / calc_hashprints(): one window of files is read straight into pinned host memory (clips of equal length
    // side by side) and goes to the device in one copy; both buffers are kept between windows and calls
    // two pinned host arenas in turn: the files of the next window are read while this one is copied and extracted
    // (spare_db: the spectrogram buffer of the last group that was not kept, for the next one -- freeing and allocating
    // hundreds of MB of device memory per window stalls the reader threads, who share the address space)
    void *spare_db = nullptr;
    size_t spare_db_cap = 0;
    // hashprints of a whole window when nothing is learned or cached (calc_hashprints): device buffer and pinned host copy
    void *d_hp_win = nullptr, *h_hp_win = nullptr;
    size_t hp_win_cap = 0;
    // ... on a stream of the collector's own (non-blocking): the tables of the next file's length are generated and
    // uploaded through the default stream while the kernels of the previous file run
    hipStream_t win_stream = nullptr;
    // pinned host copy of a group's spectrograms on their way to cache/spectros/ (grow-only)
    void *h_spec = nullptr;
    size_t h_spec_cap = 0;
    void *arena[2] = {nullptr, nullptr};
    size_t arena_cap[2] = {0, 0};
    // ... and two device copies in turn: window w + 1 is uploaded while the kernels of window w read theirs
    void *d_arena_buf[2] = {nullptr, nullptr}, *d_arena = nullptr; // d_arena: the copy of the window in hand
    size_t d_arena_cap[2] = {0, 0};
    ~hpfw_legacy_collector()
    {
        for (void *a : arena)
            if (a) (void)hipHostFree(a);
        for (void *a : d_arena_buf)
            if (a) (void)hipFree(a);
        if (spare_db) (void)hipFree(spare_db);
        if (win_stream) (void)hipStreamDestroy(win_stream);
        if (h_spec) (void)hipHostFree(h_spec);
        if (d_hp_win) (void)hipFree(d_hp_win);
        if (h_hp_win) (void)hipHostFree(h_hp_win);
    }
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

// A file of any length: the reference hands the exact sample count to NSGConstantQ (cqt.h:54-55) and so does
// this -- lengths with a prime factor above 7 take the chirp-z forward transform (k_bluestein.hip); nothing is padded.
static bool read_clip(const std::string &path, std::vector<int16_t> &pcm, std::string &why)
{
    try { // nothing may throw through the C boundary (or out of a reader thread): e.g. bad_alloc on a huge file
        return read_wav_pcm16_mono(path, pcm, why);
    } catch (const std::exception &e) {
        why = path + ": " + e.what();
        return false;
    }
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

// where the samples of a RIFF/WAVE file lie: the chunk walk of read_wav_pcm16_mono on a file descriptor (pread), without
// reading the payload
struct WavProbe {
    int fd = -1;
    int64_t data_off = 0, frames = 0; // payload offset; samples per channel
    int channels = 0;
    bool ok = false;
};

bool probe_wav(const std::string &path, WavProbe &w, std::string &why)
{
    w = WavProbe();
    const int fd = ::open(path.c_str(), O_RDONLY | O_CLOEXEC);
    if (fd < 0) {
        why = "cannot open " + path;
        return false;
    }
    struct stat stt;
    if (fstat(fd, &stt) != 0) {
        ::close(fd);
        why = "cannot stat " + path;
        return false;
    }
    const int64_t fsize = (int64_t)stt.st_size;
    unsigned char hdr[12];
    if (pread(fd, hdr, 12, 0) != 12 || std::memcmp(hdr, "RIFF", 4) || std::memcmp(hdr + 8, "WAVE", 4)) {
        ::close(fd);
        why = path + ": not a RIFF/WAVE file";
        return false;
    }
    uint16_t fmt = 0, channels = 0, bits = 0;
    uint32_t rate = 0;
    bool have_fmt = false;
    for (int64_t pos = 12; pos + 8 <= fsize;) {
        unsigned char ch[8];
        if (pread(fd, ch, 8, pos) != 8) break;
        uint32_t sz;
        std::memcpy(&sz, ch + 4, 4);
        if (!std::memcmp(ch, "fmt ", 4)) {
            if (sz < 16 || sz > 4096) break; // malformed: reported as "no data chunk" below
            unsigned char b[4096];
            if (pread(fd, b, sz, pos + 8) != (ssize_t)sz) break;
            std::memcpy(&fmt, b, 2);
            std::memcpy(&channels, b + 2, 2);
            std::memcpy(&rate, b + 4, 4);
            std::memcpy(&bits, b + 14, 2);
            if (fmt == 0xFFFE && sz >= 26) std::memcpy(&fmt, b + 24, 2); // WAVE_FORMAT_EXTENSIBLE: the sub-format's tag
            have_fmt = true;
        } else if (!std::memcmp(ch, "data", 4)) {
            if (!have_fmt || fmt != 1 || bits != 16 || (channels != 1 && channels != 2) || rate != 44100) {
                ::close(fd);
                why = path + ": only PCM16 mono/stereo at 44100 Hz is supported";
                return false;
            }
            // a streamed file may carry 0 or 0xffffffff as the size: trust the file's length instead
            const int64_t left = fsize - (pos + 8);
            int64_t bytes = sz;
            if (sz == 0 || bytes > left) bytes = std::min<int64_t>(left, 0xfffffffe);
            w.fd = fd;
            w.data_off = pos + 8;
            w.channels = channels;
            w.frames = bytes / 2 / channels;
            w.ok = true;
            return true;
        }
        pos += 8 + (int64_t)sz + (sz & 1);
    }
    ::close(fd);
    why = path + ": no data chunk";
    return false;
}

struct Loaded {
    const int16_t *pcm = nullptr; // in the collector's pinned arena
    int64_t n = 0;                // samples (mono)
    size_t arena_off = 0;         // byte offset of the clip in the arena (the device copy has the same layout)
    bool ok = false;
};

// One window of files into the collector's pinned arena, clips of equal length side by side (so that a group goes
// through the device stages as one batch, straight from the window's single host-to-device copy).  The reference
// reads its files on a taskflow pool (parallel_collector.h:95-108); here a team of host threads walks the headers, then
// reads the payloads with pread() into their slots (stereo is averaged as MonoLoader's "mix" downmix).  The reader
// threads also prepare the host half of the tables of every length they meet (hpfw_gpu_prepare_length: a corpus of
// full-length tracks brings a new length with almost every file).
// Returns the bytes of the arena in use (0: nothing readable).
size_t read_window(hpfw_legacy_collector *c, int slot, const char **filenames, const std::vector<int> &files, size_t first, size_t last,
                   std::vector<Loaded> &out, std::string &first_why)
{
    const auto t_begin = std::chrono::steady_clock::now();
    std::mutex why_mtx;
    first_why.clear(); // error messages are thread-local and this runs on a thread of its own: the first failure goes back to the caller
    const int count = (int)(last - first);
    out.assign((size_t)count, Loaded());
    std::vector<WavProbe> probes((size_t)count);
    unsigned team = std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
    team = std::min<unsigned>(team, (unsigned)std::max(count, 1));
    auto run = [&](auto fn) {
        std::atomic<int> next{0};
        auto work = [&] {
            for (int i; (i = next.fetch_add(1)) < count;) fn(i);
        };
        std::vector<std::thread> th;
        for (unsigned t = 1; t < team; ++t) th.emplace_back(work);
        work();
        for (auto &t : th) t.join();
    };
    auto fail = [&](const std::string &why) {
        std::scoped_lock lock(why_mtx);
        if (first_why.empty()) first_why = why;
    };
    run([&](int i) {
        std::string why;
        try {
            if (!probe_wav(filenames[files[first + (size_t)i]], probes[(size_t)i], why)) fail(why);
            else if (c->gpu) (void)hpfw_gpu_prepare_length(c->gpu, probes[(size_t)i].frames); // (too short: skipped later)
        } catch (const std::exception &e) {
            fail(std::string(filenames[files[first + (size_t)i]]) + ": " + e.what());
        }
    });
    const auto t_probe = std::chrono::steady_clock::now();
    // slots: by length, then input order; every group starts 16-byte aligned
    std::map<int64_t, std::vector<int>> by_len;
    for (int i = 0; i < count; ++i)
        if (probes[(size_t)i].ok && probes[(size_t)i].frames > 0) by_len[probes[(size_t)i].frames].push_back(i);
    size_t bytes = 0;
    for (auto &kv : by_len) {
        bytes = (bytes + 15) / 16 * 16;
        for (int i : kv.second) {
            out[(size_t)i].arena_off = bytes;
            out[(size_t)i].n = kv.first;
            bytes += (size_t)kv.first * 2;
        }
    }
    if (bytes > c->arena_cap[slot]) {
        if (c->arena[slot]) (void)hipHostFree(c->arena[slot]);
        c->arena[slot] = nullptr;
        c->arena_cap[slot] = 0;
        const size_t want = std::max(bytes + bytes / 4, (size_t)64 << 20);
        if (hipHostMalloc(&c->arena[slot], want, hipHostMallocDefault) != hipSuccess) {
            c->arena[slot] = nullptr;
            first_why = "prepare: out of pinned host memory";
            for (WavProbe &w : probes)
                if (w.fd >= 0) ::close(w.fd);
            return 0;
        }
        c->arena_cap[slot] = want;
    }
    run([&](int i) {
        WavProbe &w = probes[(size_t)i];
        if (!w.ok || w.frames <= 0) {
            if (w.fd >= 0) ::close(w.fd);
            return;
        }
        int16_t *dst = reinterpret_cast<int16_t *>(static_cast<char *>(c->arena[slot]) + out[(size_t)i].arena_off);
        bool ok = true;
        try {
            if (w.channels == 1) {
                int64_t done = 0;
                const int64_t want = w.frames * 2;
                while (done < want) {
                    const ssize_t r = pread(w.fd, reinterpret_cast<char *>(dst) + done, (size_t)(want - done), w.data_off + done);
                    if (r <= 0) break;
                    done += r;
                }
                ok = done == want;
            } else {
                std::vector<int16_t> raw((size_t)w.frames * 2);
                int64_t done = 0;
                const int64_t want = w.frames * 4;
                while (done < want) {
                    const ssize_t r = pread(w.fd, reinterpret_cast<char *>(raw.data()) + done, (size_t)(want - done), w.data_off + done);
                    if (r <= 0) break;
                    done += r;
                }
                ok = done == want;
                for (int64_t k = 0; ok && k < w.frames; ++k) dst[k] = (int16_t)(((int)raw[(size_t)(2 * k)] + (int)raw[(size_t)(2 * k + 1)]) / 2);
            }
        } catch (const std::exception &e) { // e.g. bad_alloc on a huge stereo file
            ok = false;
        }
        ::close(w.fd);
        if (!ok) fail(std::string(filenames[files[first + (size_t)i]]) + ": short read");
        out[(size_t)i].pcm = dst;
        out[(size_t)i].ok = ok;
    });
    if (std::getenv("HPFW_FFI_TIMING")) {
        const auto t_end = std::chrono::steady_clock::now();
        auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        std::fprintf(stderr, "[hpfw ffi]   reader: headers + tables of new lengths %.1f ms, payloads %.1f ms (%u threads)\n", ms(t_begin, t_probe),
                     ms(t_probe, t_end), team);
    }
    return bytes; // (skipped files, parallel_collector.h:101-103: first_why holds the message)
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

// the window's clips on the device: one copy of the arena (its layout is kept)
bool upload_window(hpfw_legacy_collector *c, int slot, size_t bytes)
{
    if (bytes > c->d_arena_cap[slot]) {
        if (c->d_arena_buf[slot]) (void)hipFree(c->d_arena_buf[slot]);
        c->d_arena_buf[slot] = nullptr;
        c->d_arena_cap[slot] = 0;
        const size_t want = std::max(bytes + bytes / 4, (size_t)64 << 20);
        if (hipMalloc(&c->d_arena_buf[slot], want) != hipSuccess) {
            c->d_arena_buf[slot] = nullptr;
            hpfw_internal_set_error("prepare: out of device memory");
            return false;
        }
        c->d_arena_cap[slot] = want;
    }
    c->d_arena = c->d_arena_buf[slot];
    return bytes == 0 || hipMemcpy(c->d_arena, c->arena[slot], bytes, hipMemcpyHostToDevice) == hipSuccess;
}

// dB spectrograms [n][121][C] of n equally long clips of the window (device memory, caller frees); the clips lie side by
// side in the device copy of the arena, the first at `first_off`
// (the buffer comes from release_spectrograms' spare when that is large enough; *cap = its size)
float *group_spectrograms(hpfw_legacy_collector *c, size_t first_off, size_t n, int64_t len, const hpfw_geometry &g, size_t *cap)
{
    float *d_db = nullptr;
    const size_t need = n * (size_t)121 * g.c * 4;
    if (c->spare_db && c->spare_db_cap >= need) {
        d_db = static_cast<float *>(c->spare_db);
        *cap = c->spare_db_cap;
        c->spare_db = nullptr;
        c->spare_db_cap = 0;
    } else if (hipMalloc((void **)&d_db, need) != hipSuccess) {
        hpfw_internal_set_error("prepare: out of device memory");
        return nullptr;
    } else {
        *cap = need;
    }
    const int16_t *d_pcm = reinterpret_cast<const int16_t *>(static_cast<const char *>(c->d_arena) + first_off);
    // (no synchronisation: what follows runs on the default stream too, and the copies to the host wait for it)
    if (hpfw_gpu_stage_spectrogram(c->gpu, d_pcm, len, (int64_t)n, d_db, nullptr) != 0) {
        (void)hipFree(d_db);
        return nullptr;
    }
    return d_db;
}

// a group's spectrogram buffer is done with: the larger of it and the spare stays for the next group
void release_spectrograms(hpfw_legacy_collector *c, float *d_db, size_t cap)
{
    if (cap > c->spare_db_cap) {
        if (c->spare_db) (void)hipFree(c->spare_db);
        c->spare_db = d_db;
        c->spare_db_cap = cap;
    } else {
        (void)hipFree(d_db);
    }
}

// hashprints of n clips from their dB spectrograms: out[k] = new uint64_t[g.n_hp]
// the collector's hashprint buffers (device, and pinned on the host) hold at least `bytes`
bool ensure_hp_buffers(hpfw_legacy_collector *c, size_t bytes)
{
    if (bytes <= c->hp_win_cap) return true;
    if (c->d_hp_win) (void)hipFree(c->d_hp_win);
    if (c->h_hp_win) (void)hipHostFree(c->h_hp_win);
    c->d_hp_win = c->h_hp_win = nullptr;
    c->hp_win_cap = 0;
    const size_t want = std::max(bytes + bytes / 4, (size_t)8 << 20);
    if (hipMalloc(&c->d_hp_win, want) != hipSuccess || hipHostMalloc(&c->h_hp_win, want, hipHostMallocDefault) != hipSuccess) {
        hpfw_internal_set_error("prepare: out of memory for the hashprints");
        return false;
    }
    c->hp_win_cap = want;
    return true;
}

// hashprints of n clips from their dB spectrograms: out[k] = new uint64_t[g.n_hp]
bool group_hashprints(hpfw_legacy_collector *c, const float *d_db, size_t n, const hpfw_geometry &g, uint64_t **out)
{
    const size_t bytes = n * (size_t)g.n_hp * 8;
    if (!ensure_hp_buffers(c, bytes)) return false;
    if (hpfw_gpu_hashprints_from_db(c->gpu, d_db, (int64_t)n, g.c, (uint64_t *)c->d_hp_win, nullptr) != 0) return false;
    if (hipMemcpy(c->h_hp_win, c->d_hp_win, bytes, hipMemcpyDeviceToHost) != hipSuccess) {
        hpfw_internal_set_error("prepare: D2H copy failed");
        return false;
    }
    for (size_t k = 0; k < n; ++k) {
        out[k] = new uint64_t[(size_t)g.n_hp];
        std::memcpy(out[k], static_cast<const uint64_t *>(c->h_hp_win) + k * (size_t)g.n_hp, (size_t)g.n_hp * 8);
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
// The two halves of prepare() around the point where the filters are learned: accumulate() is preprocess()'s
// parallel_for (parallel_collector.h:85-105: spectrogram, covariance into accum_cov, spectrogram cached),
// finish() is collect_fingerprints (:115-137) for the files of this call.  Between the two the caller learns the
// filters -- from this collector's accum_cov alone (collect_files) or from the sum over the collectors of a
// multi-GPU group (hpfw_gpu_group_prepare, multi.cpp).
struct hpfw_prepare_job {
    std::vector<uint64_t *> hp;
    std::vector<int> hp_size;
    std::vector<KeptGroup> kept;
    std::vector<int> again; // files whose spectrogram could not be kept
    size_t kept_bytes = 0;
    int64_t used = 0;
    bool learn = true, cache_spectros = true;
};

// one pass over `files`: first = covariance (+ keep the spectrograms); otherwise hashprints at once
static void prepare_pass(hpfw_legacy_collector *c, const char **filenames, hpfw_prepare_job &job, const std::vector<int> &files,
                         bool first)
{
    const std::string spectro_dir = c->cache_dir + "spectros/";
    size_t keep_budget = (size_t)32 << 30;
    if (const char *e = std::getenv("HPFW_PREPARE_KEEP_GB")) keep_budget = (size_t)std::max(0.0, std::atof(e) * 1073741824.0);
    (void)hipSetDevice(hpfw_gpu_device(c->gpu));
    // the windows: at most 256 files and about 1 GiB of audio each
    std::vector<std::pair<size_t, size_t>> windows;
    for (size_t at = 0; at < files.size();) {
        size_t end = at;
        uintmax_t bytes = 0;
        while (end < files.size() && end - at < 256 && (end == at || bytes < ((uintmax_t)1 << 30))) {
            std::error_code ec;
            const uintmax_t sz = std::filesystem::file_size(filenames[files[end]], ec);
            if (!ec) bytes += sz;
            ++end;
        }
        windows.emplace_back(at, end);
        at = end;
    }
    // window w + 1 is read (into the other arena) by a task of its own while window w is copied and extracted
    struct Read {
        std::vector<Loaded> clips;
        size_t used = 0;
        double ms = 0.0;
        std::string why;
    };
    const int device = hpfw_gpu_device(c->gpu);
    auto start_read = [&](size_t w) {
        return std::async(std::launch::async, [&, w] {
            Read r;
            const auto t_a = std::chrono::steady_clock::now();
            (void)hipSetDevice(device);
            r.used = read_window(c, (int)(w & 1), filenames, files, windows[w].first, windows[w].second, r.clips, r.why);
            r.ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_a).count();
            return r;
        });
    };
    const bool timing = std::getenv("HPFW_FFI_TIMING") != nullptr; // where a window's time goes, on stderr
    // calc_hashprints (nothing learned or cached): the extraction of window w is enqueued on the collector's stream and its
    // hashprints are fetched only after window w + 1 has been uploaded -- the upload runs under the kernels
    struct Part {
        std::vector<int> pos; // positions of the group's clips in the window
        hpfw_geometry g;
        size_t off;           // hashprints before this group in the window's buffer
        bool done;
    };
    struct Pending {
        bool active = false, ok = true;
        std::vector<Part> parts;
        size_t total = 0, at = 0;
    } pend;
    auto finish = [&]() { // the results of the window whose kernels are in flight
        if (!pend.active) return;
        pend.active = false;
        if (c->win_stream && hipStreamSynchronize(c->win_stream) != hipSuccess) {
            hpfw_internal_set_error("prepare: the window's kernels failed");
            pend.ok = false;
        } else if (c->win_stream) {
            // everything this collector has queued on its handle ran on that stream (the tables' stream and the side streams
            // are joined into it), and the next window's groups are queued only after this point
            hpfw_internal_note_idle(c->gpu);
        }
        for (size_t k = 0; pend.ok && k < pend.parts.size(); ++k) {
            const Part &pt = pend.parts[k];
            if (!pt.done) continue;
            for (size_t q = 0; q < pt.pos.size(); ++q) {
                const int id = files[pend.at + (size_t)pt.pos[q]];
                uint64_t *out = new uint64_t[(size_t)pt.g.n_hp];
                std::memcpy(out, static_cast<const uint64_t *>(c->h_hp_win) + pt.off + q * (size_t)pt.g.n_hp, (size_t)pt.g.n_hp * 8);
                job.hp[(size_t)id] = out;
                job.hp_size[(size_t)id] = (int)pt.g.n_hp;
            }
        }
    };
    std::future<Read> ahead;
    if (!windows.empty()) ahead = start_read(0);
    for (size_t w = 0; w < windows.size(); ++w) {
        const size_t at = windows[w].first, end = windows[w].second;
        const auto t_0 = std::chrono::steady_clock::now();
        Read got = ahead.get();
        if (!got.why.empty()) hpfw_internal_set_error(got.why.c_str()); // the message survives the skipped files
        if (w + 1 < windows.size()) ahead = start_read(w + 1);
        std::vector<Loaded> &clips = got.clips;
        const size_t used = got.used;
        const auto t_1 = std::chrono::steady_clock::now();
        if (!upload_window(c, (int)(w & 1), used)) {
            if (ahead.valid()) (void)ahead.get(); // the task works on this function's state: it ends before we return
            break;
        }
        finish(); // (the previous window's kernels ran under this upload)
        const auto t_2 = std::chrono::steady_clock::now();
        // length -> positions in the window, in input order.  A group whose files were all read lies side by side in the
        // arena; one with a failed read in its middle is cut into its contiguous runs
        std::map<int64_t, std::vector<std::vector<int>>> by_len;
        {
            std::map<int64_t, std::vector<int>> all;
            for (size_t i = 0; i < clips.size(); ++i)
                if (clips[i].n > 0) all[clips[i].n].push_back((int)i);
            for (auto &kv : all) {
                std::vector<int> run;
                for (int i : kv.second) {
                    if (clips[(size_t)i].ok) {
                        run.push_back(i);
                    } else if (!run.empty()) {
                        by_len[kv.first].push_back(run);
                        run.clear();
                    }
                }
                if (!run.empty()) by_len[kv.first].push_back(run);
            }
        }
        // Nothing learned, nothing cached (calc_hashprints): every group goes straight from its PCM to hashprints in one
        // buffer for the window -- the groups are only enqueued, one copy and one synchronisation per window
        const bool direct = !(first && job.learn) && !(first && job.cache_spectros);
        if (direct) {
            pend.parts.clear();
            pend.total = 0;
            pend.at = at;
            pend.ok = true;
            for (auto &kv : by_len)
                for (const std::vector<int> &pos : kv.second) {
                    hpfw_geometry g;
                    if (hpfw_gpu_geometry(c->gpu, kv.first, &g) != 0 || g.n_frames < 2 || g.n_hp <= 0) continue; // skipped
                    pend.parts.push_back(Part{pos, g, pend.total, false});
                    pend.total += pos.size() * (size_t)g.n_hp;
                }
            const size_t total = pend.total;
            bool ok = true;
            if (!ensure_hp_buffers(c, total * 8)) ok = false;
            if (ok && !c->win_stream && hipStreamCreateWithFlags(&c->win_stream, hipStreamNonBlocking) != hipSuccess) {
                c->win_stream = nullptr;
                hpfw_internal_set_error("prepare: no stream");
                ok = false;
            }
            const auto t_enq = std::chrono::steady_clock::now();
            for (size_t k = 0; ok && k < pend.parts.size(); ++k) {
                Part &pt = pend.parts[k];
                const int16_t *d_pcm = reinterpret_cast<const int16_t *>(static_cast<const char *>(c->d_arena) + clips[(size_t)pt.pos[0]].arena_off);
                pt.done = hpfw_gpu_extract_pcm16(c->gpu, d_pcm, clips[(size_t)pt.pos[0]].n, (int64_t)pt.pos.size(),
                                                 static_cast<uint64_t *>(c->d_hp_win) + pt.off, c->win_stream) == 0; // a failed group is skipped
            }
            if (ok && total > 0 && hipMemcpyAsync(c->h_hp_win, c->d_hp_win, total * 8, hipMemcpyDeviceToHost, c->win_stream) != hipSuccess) {
                hpfw_internal_set_error("prepare: D2H copy failed");
                ok = false;
            }
            if (timing)
                std::fprintf(stderr, "[hpfw ffi]   %zu groups enqueued in %.1f ms (tables of new lengths included)\n", pend.parts.size(),
                             std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_enq).count());
            pend.ok = ok;
            pend.active = true; // fetched by finish(): after the next window's upload, or after the last window
        }
        // the window's spectrograms on their way to cache/spectros/: one pinned buffer for all of them, written to the cache
        // by the team of host threads when the window's groups are through (a window of files of different lengths is as
        // many groups of one file: written group by group, one thread would do all the writing)
        struct SpecWrite {
            std::string path;
            size_t off; // floats into the pinned buffer
            int32_t c;
        };
        std::vector<SpecWrite> spec_writes;
        size_t spec_used = 0;
        if (!direct && first && job.cache_spectros) {
            size_t total = 0;
            for (auto &kv : by_len)
                for (const std::vector<int> &pos : kv.second) {
                    hpfw_geometry g;
                    if (hpfw_gpu_geometry(c->gpu, kv.first, &g) == 0 && g.n_frames >= 2) total += pos.size() * (size_t)121 * g.c * 4;
                }
            if (total > c->h_spec_cap) { // (pinned: the copies run at the link's rate, not through the driver's staging)
                if (c->h_spec) (void)hipHostFree(c->h_spec);
                c->h_spec = nullptr;
                c->h_spec_cap = 0;
                if (hipHostMalloc(&c->h_spec, total + total / 8, hipHostMallocDefault) == hipSuccess) c->h_spec_cap = total + total / 8;
                else c->h_spec = nullptr;
            }
        }
        for (auto &kv : by_len)
          for (const std::vector<int> &pos : kv.second) {
            if (direct) break;
            const int64_t len = kv.first;
            hpfw_geometry g;
            if (hpfw_gpu_geometry(c->gpu, len, &g) != 0 || g.n_frames < 2) continue; // skipped
            size_t db_cap = 0;
            float *d_db = group_spectrograms(c, clips[(size_t)pos[0]].arena_off, pos.size(), len, g, &db_cap);
            if (!d_db) continue;
            std::vector<int> ids;
            for (int q : pos) ids.push_back(files[at + (size_t)q]);
            const size_t sz = pos.size() * (size_t)121 * g.c * 4;
            if (first && job.cache_spectros) {
                // cache.set_spectro(filename, spectro) (parallel_collector.h:98-100): "it will also be needed
                // when adding new tracks" -- a later prepare() recomputes every cached track's hashprints
                if (c->h_spec && spec_used + sz <= c->h_spec_cap &&
                    hipMemcpy(static_cast<char *>(c->h_spec) + spec_used, d_db, sz, hipMemcpyDeviceToHost) == hipSuccess) {
                    for (size_t k = 0; k < ids.size(); ++k)
                        spec_writes.push_back(SpecWrite{spectro_dir + std::filesystem::path(filenames[ids[k]]).stem().string(),
                                                        spec_used / 4 + k * (size_t)121 * g.c, (int32_t)g.c});
                    spec_used += sz;
                }
            }
            if (first && job.learn) {
                if (hpfw_gpu_cov_accumulate_db(c->gpu, d_db, (int64_t)pos.size(), g.c, nullptr) == 0) job.used += (int64_t)pos.size();
                if (g.n_hp > 0 && job.kept_bytes + sz <= keep_budget) {
                    job.kept.push_back(KeptGroup{ids, g, d_db});
                    job.kept_bytes += sz;
                    continue;
                }
                if (g.n_hp > 0) job.again.insert(job.again.end(), ids.begin(), ids.end());
            } else if (g.n_hp > 0) {
                std::vector<uint64_t *> out(pos.size(), nullptr);
                if (group_hashprints(c, d_db, pos.size(), g, out.data()))
                    for (size_t k = 0; k < ids.size(); ++k) {
                        job.hp[(size_t)ids[k]] = out[k];
                        job.hp_size[(size_t)ids[k]] = (int)g.n_hp;
                    }
            }
            release_spectrograms(c, d_db, db_cap); // (its next user is ordered after this group on the default stream; hipFree waits)
        }
        if (!spec_writes.empty()) {
            const float *host = static_cast<const float *>(c->h_spec);
            host_team((int)spec_writes.size(), [&](int k) {
                (void)save_spectro_cereal(spec_writes[(size_t)k].path, host + spec_writes[(size_t)k].off, spec_writes[(size_t)k].c);
            });
        }
        if (timing) {
            const auto t_3 = std::chrono::steady_clock::now();
            auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
            std::fprintf(stderr, "[hpfw ffi] window of %zu files, %.1f MB: read %.1f ms (waited %.1f ms), upload %.1f ms, device + results %.1f ms\n",
                         end - at, used / 1e6, got.ms, ms(t_0, t_1), ms(t_1, t_2), ms(t_2, t_3));
        }
    }
    finish();
}

static void prepare_accumulate(hpfw_legacy_collector *c, const char **filenames, int n, hpfw_prepare_job &job)
{
    if (job.cache_spectros) {
        std::error_code ec;
        std::filesystem::create_directories(c->cache_dir + "spectros/", ec);
    }
    job.hp.assign((size_t)n, nullptr);
    job.hp_size.assign((size_t)n, 0);
    std::vector<int> all((size_t)n);
    for (int i = 0; i < n; ++i) all[(size_t)i] = i;
    prepare_pass(c, filenames, job, all, true);
}

// hashprints of the spectrograms kept on the device, then of the files that have to be read again; ok = false:
// the filters could not be learned -- release what was kept and produce nothing
static void prepare_finish(hpfw_legacy_collector *c, const char **filenames, hpfw_prepare_job &job, bool ok)
{
    (void)hipSetDevice(hpfw_gpu_device(c->gpu));
    for (KeptGroup &k : job.kept) {
        std::vector<uint64_t *> out(k.files.size(), nullptr);
        if (ok && group_hashprints(c, k.d_db, k.files.size(), k.g, out.data()))
            for (size_t q = 0; q < k.files.size(); ++q) {
                job.hp[(size_t)k.files[q]] = out[q];
                job.hp_size[(size_t)k.files[q]] = (int)k.g.n_hp;
            }
        (void)hipFree(k.d_db);
    }
    job.kept.clear();
    if (ok && !job.again.empty()) {
        std::sort(job.again.begin(), job.again.end());
        prepare_pass(c, filenames, job, job.again, false);
    }
}

static bool collect_files(hpfw_legacy_collector *c, const char **filenames, int n, bool learn, bool cache_spectros,
                          std::vector<uint64_t *> &hp, std::vector<int> &hp_size)
{
    hpfw_prepare_job job;
    job.learn = learn;
    job.cache_spectros = cache_spectros;
    prepare_accumulate(c, filenames, n, job);
    bool failed = false;
    if (learn) {
        if (job.used > 0) {
            c->filters.assign((size_t)HPFW_FILTERS * HPFW_FRAME_SIZE, 0.0f);
            if (hpfw_gpu_learn_filters(c->gpu, c->filters.data()) != 0) {
                c->filters.clear();
                failed = true;
            } else {
                par_collector_save(c, nullptr);
            }
        }
        prepare_finish(c, filenames, job, !failed);
    }
    hp.swap(job.hp);
    hp_size.swap(job.hp_size);
    return !failed;
}

// collect_fingerprints (parallel_collector.h:114-137) walks EVERY spectrogram under cache/spectros/, not only
// the files of this call: tracks indexed by an earlier run come back with hashprints under the filters just
// learned.  The files of this call are already done (their spectrograms never left the device); this adds the
// others: read by a team of host threads, grouped by width, projected and packed on the GPU.
static void collect_cached(hpfw_legacy_collector *c, const std::vector<std::string> &done_stems,
                           std::vector<std::string> &stems, std::vector<uint64_t *> &hp, std::vector<int> &hp_size)
{
    std::error_code ec;
    std::vector<std::string> paths;
    const std::unordered_set<std::string> done(done_stems.begin(), done_stems.end()); // a 100 k-track cache: no quadratic scan
    for (const auto &e : std::filesystem::directory_iterator(c->cache_dir + "spectros/", ec)) {
        if (!e.is_regular_file(ec)) continue;
        // the cache names its files by the track's stem (cache.h:30-33); the name returned is that file name as it is.
        // (The reference returns path(cache file).stem() -- parallel_collector.h:123 -- i.e. a second stem: "a.b.wav" is
        // cached as "a.b" and comes back as "a" there, as "a.b" here; INTEGRATION.md notes the deviation.)
        const std::string stem = e.path().filename().string();
        if (!done.count(stem)) paths.push_back(e.path().string());
    }
    std::sort(paths.begin(), paths.end()); // the reference's order is the directory's (and racy, :129): sorted here
    for (size_t at = 0; at < paths.size();) {
        const size_t end = std::min(paths.size(), at + 256);
        struct Item {
            std::vector<float> cm;
            int32_t cols = 0;
            bool ok = false;
        };
        std::vector<Item> items(end - at);
        host_team((int)items.size(), [&](int i) { items[(size_t)i].ok = load_spectro_cereal(paths[at + (size_t)i], items[(size_t)i].cm, items[(size_t)i].cols); });
        std::vector<uint64_t *> win_hp(items.size(), nullptr);
        std::vector<int> win_size(items.size(), 0);
        std::map<int32_t, std::vector<int>> by_cols;
        for (size_t i = 0; i < items.size(); ++i)
            if (items[i].ok && items[i].cols >= HPFW_CONTEXT + HPFW_LAG) by_cols[items[i].cols].push_back((int)i);
        for (auto &kv : by_cols) {
            const int32_t cols = kv.first;
            const std::vector<int> &pos = kv.second;
            hpfw_geometry g{};
            g.c = cols;
            g.n_frames = cols - (HPFW_CONTEXT - 1);
            g.n_hp = g.n_frames - HPFW_LAG;
            std::vector<float> bm(pos.size() * (size_t)HPFW_BINS * cols);
            host_team((int)pos.size(), [&](int k) {
                const float *cm = items[(size_t)pos[(size_t)k]].cm.data();
                float *out = bm.data() + (size_t)k * HPFW_BINS * cols;
                for (int32_t col = 0; col < cols; ++col)
                    for (int32_t b = 0; b < HPFW_BINS; ++b) out[(size_t)b * cols + col] = cm[(size_t)col * HPFW_BINS + b];
            });
            DevMem d_db;
            std::vector<uint64_t *> out(pos.size(), nullptr);
            if (!d_db.alloc(bm.size() * 4) || hipMemcpy(d_db.p, bm.data(), bm.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
                !group_hashprints(c, (const float *)d_db.p, pos.size(), g, out.data()))
                continue;
            for (size_t k = 0; k < pos.size(); ++k) {
                win_hp[(size_t)pos[k]] = out[k];
                win_size[(size_t)pos[k]] = (int)g.n_hp;
            }
        }
        for (size_t i = 0; i < items.size(); ++i) { // groups ran by width; the results keep the sorted order
            if (!win_hp[i]) continue;
            stems.push_back(std::filesystem::path(paths[at + i]).filename().string());
            hp.push_back(win_hp[i]);
            hp_size.push_back(win_size[i]);
        }
        at = end;
    }
}

static FilenameHashprintPair *prepare_impl(hpfw_legacy_collector *c, const char **filenames, int n, int *got)
{
    if (got) *got = 0;
    if (!c || !filenames || n < 0 || !got) return nullptr;
    const bool learn = !(std::getenv("HPFW_PREPARE_KEEP_FILTERS") && !c->filters.empty());
    const bool cache_spectros = !std::getenv("HPFW_NO_SPECTRO_CACHE");
    std::vector<uint64_t *> hp;
    std::vector<int> hp_size;
    if (!collect_files(c, filenames, n, learn, cache_spectros, hp, hp_size)) return nullptr;
    std::vector<std::string> stems;
    std::vector<uint64_t *> r_hp;
    std::vector<int> r_size;
    for (int i = 0; i < n; ++i) {
        if (!hp[(size_t)i]) continue;
        stems.push_back(std::filesystem::path(filenames[i]).stem().string());
        r_hp.push_back(hp[(size_t)i]);
        r_size.push_back(hp_size[(size_t)i]);
    }
    if (cache_spectros) {
        const std::vector<std::string> done = stems;
        collect_cached(c, done, stems, r_hp, r_size);
    }
    auto *res = new FilenameHashprintPair[std::max<size_t>(stems.size(), 1)];
    for (size_t w = 0; w < stems.size(); ++w) {
        res[w].filename = new char[stems[w].size() + 1];
        std::memcpy(res[w].filename, stems[w].c_str(), stems[w].size() + 1);
        res[w].hashprint = r_hp[w];
        res[w].hp_size = r_size[w];
    }
    *got = (int)stems.size();
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
    if (!collect_files(c, filenames, n, false, false, hp, hp_size)) return nullptr;
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

// ---- the halves of prepare() for a multi-GPU host (libhpfw_gpu_multi.so; declared in legacy_internal.h) ----
hpfw_legacy_collector *hpfw_internal_collector_on_device(int device, const char *cache)
{
    auto *c = new hpfw_legacy_collector();
    if (hpfw_gpu_create(device, &c->gpu) != 0) {
        delete c;
        return nullptr;
    }
    par_collector_load(c, cache);
    return c;
}

hpfw_gpu *hpfw_internal_collector_gpu(hpfw_legacy_collector *c) { return c ? c->gpu : nullptr; }

int hpfw_internal_collector_set_filters(hpfw_legacy_collector *c, const float *f)
{
    if (!c || !f) return HPFW_E_INVALID;
    c->filters.assign(f, f + (size_t)HPFW_FILTERS * HPFW_FRAME_SIZE);
    return hpfw_gpu_set_filters(c->gpu, f);
}

hpfw_prepare_job *hpfw_internal_prepare_accumulate(hpfw_legacy_collector *c, const char **filenames, int n, int learn)
{
    return guarded([&]() -> hpfw_prepare_job * {
        if (!c || !filenames || n < 0) return nullptr;
        auto *job = new hpfw_prepare_job();
        job->learn = learn != 0;
        job->cache_spectros = !std::getenv("HPFW_NO_SPECTRO_CACHE");
        prepare_accumulate(c, filenames, n, *job);
        return job;
    });
}

int64_t hpfw_internal_prepare_used(const hpfw_prepare_job *job) { return job ? job->used : 0; }

// hashprints of the job's files (input order, failed files dropped) [+ with_cached: every other spectrogram of
// the cache, sorted by name]; consumes the job.  ok = 0: learning failed, release and return NULL.
FilenameHashprintPair *hpfw_internal_prepare_finish(hpfw_legacy_collector *c, hpfw_prepare_job *job, const char **filenames, int n,
                                                    int ok, int with_cached, int *got)
{
    if (got) *got = 0;
    FilenameHashprintPair *res = guarded([&]() -> FilenameHashprintPair * {
        if (!c || !job || !got) return nullptr;
        if (job->learn || !ok) prepare_finish(c, filenames, *job, ok != 0);
        if (!ok) return nullptr;
        std::vector<std::string> stems;
        std::vector<uint64_t *> r_hp;
        std::vector<int> r_size;
        for (int i = 0; i < n; ++i) {
            if (!job->hp[(size_t)i]) continue;
            stems.push_back(std::filesystem::path(filenames[i]).stem().string());
            r_hp.push_back(job->hp[(size_t)i]);
            r_size.push_back(job->hp_size[(size_t)i]);
        }
        if (with_cached && job->cache_spectros) {
            std::vector<std::string> done;
            for (int i = 0; i < n; ++i) done.push_back(std::filesystem::path(filenames[i]).stem().string()); // other shards' files too
            collect_cached(c, done, stems, r_hp, r_size);
        }
        auto *out = new FilenameHashprintPair[std::max<size_t>(stems.size(), 1)];
        for (size_t w = 0; w < stems.size(); ++w) {
            out[w].filename = new char[stems[w].size() + 1];
            std::memcpy(out[w].filename, stems[w].c_str(), stems[w].size() + 1);
            out[w].hashprint = r_hp[w];
            out[w].hp_size = r_size[w];
        }
        *got = (int)stems.size();
        return out;
    });
    delete job;
    return res;
}

// the tracks of the cache other than the n named files (whose hashprints the shards have just produced), sorted;
// consumes the (empty) job
FilenameHashprintPair *hpfw_internal_prepare_finish_cached(hpfw_legacy_collector *c, hpfw_prepare_job *job, const char **filenames, int n,
                                                           int *got)
{
    if (got) *got = 0;
    FilenameHashprintPair *res = guarded([&]() -> FilenameHashprintPair * {
        if (!c || !got) return nullptr;
        (void)hipSetDevice(hpfw_gpu_device(c->gpu));
        std::vector<std::string> done, stems;
        std::vector<uint64_t *> r_hp;
        std::vector<int> r_size;
        for (int i = 0; i < n; ++i) done.push_back(std::filesystem::path(filenames[i]).stem().string());
        collect_cached(c, done, stems, r_hp, r_size);
        auto *out = new FilenameHashprintPair[std::max<size_t>(stems.size(), 1)];
        for (size_t w = 0; w < stems.size(); ++w) {
            out[w].filename = new char[stems[w].size() + 1];
            std::memcpy(out[w].filename, stems[w].c_str(), stems[w].size() + 1);
            out[w].hashprint = r_hp[w];
            out[w].hp_size = r_size[w];
        }
        *got = (int)stems.size();
        return out;
    });
    delete job;
    return res;
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
