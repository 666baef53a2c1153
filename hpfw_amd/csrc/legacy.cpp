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
#include <cstdio>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <string>
#include <vector>

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
            std::vector<char> b(sz);
            f.read(b.data(), sz);
            if (sz < 16) break;
            std::memcpy(&fmt, b.data(), 2);
            std::memcpy(&channels, b.data() + 2, 2);
            std::memcpy(&rate, b.data() + 4, 4);
            std::memcpy(&bits, b.data() + 14, 2);
            have_fmt = true;
        } else if (!std::memcmp(id, "data", 4)) {
            if (!have_fmt || fmt != 1 || bits != 16 || (channels != 1 && channels != 2) || rate != 44100) {
                why = path + ": only PCM16 mono/stereo at 44100 Hz is supported";
                return false;
            }
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
    if (!read_wav_pcm16_mono(path, pcm, why)) return false;
    if (std::getenv("HPFW_STRICT_LENGTH")) return true;
    const int64_t want = hpfw_gpu_supported_length((int64_t)pcm.size());
    if (want > (int64_t)pcm.size()) pcm.resize((size_t)want, 0);
    return true;
}

uint64_t *par_collector_calc_hashprint(hpfw_legacy_collector *c, const char *filename, int *size)
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

// ParallelCollector::prepare (parallel_collector.h:48-52) with already-learned filters: per file
// errors are skipped as the reference does (parallel_collector.h:101-103), so *got may be < n.
// The result name is the stem of the path (parallel_collector.h:123,129).
FilenameHashprintPair *par_collector_prepare(hpfw_legacy_collector *c, const char **filenames, int n, int *got)
{
    if (got) *got = 0;
    if (!c || !filenames || n < 0 || !got) return nullptr;
    // preprocess (parallel_collector.h:82-112): add every file's frame covariance to accum_cov, take the
    // 64 leading eigenvectors as the new filters, save().  HPFW_PREPARE_KEEP_FILTERS=1 skips this step
    // and keeps the filters that load() / a previous prepare() installed.
    if (!(std::getenv("HPFW_PREPARE_KEEP_FILTERS") && !c->filters.empty())) {
        int used = 0;
        for (int i = 0; i < n; ++i) {
            std::vector<int16_t> pcm;
            std::string why;
            if (!read_clip(filenames[i], pcm, why)) continue; // skipped, parallel_collector.h:101-103
            if (hpfw_gpu_cov_accumulate_pcm16_host(c->gpu, pcm.data(), (int64_t)pcm.size(), 1) == 0) ++used;
        }
        if (used > 0) {
            c->filters.assign((size_t)HPFW_FILTERS * HPFW_FRAME_SIZE, 0.0f);
            if (hpfw_gpu_learn_filters(c->gpu, c->filters.data()) != 0) {
                c->filters.clear();
                return nullptr;
            }
            par_collector_save(c, nullptr);
        }
    }
    auto *res = new FilenameHashprintPair[(size_t)(n > 0 ? n : 1)];
    int w = 0;
    for (int i = 0; i < n; ++i) {
        int size = 0;
        uint64_t *hp = par_collector_calc_hashprint(c, filenames[i], &size);
        if (!hp) continue;
        const std::string stem = std::filesystem::path(filenames[i]).stem().string();
        res[w].filename = new char[stem.size() + 1];
        std::memcpy(res[w].filename, stem.c_str(), stem.size() + 1);
        res[w].hashprint = hp;
        res[w].hp_size = size;
        ++w;
    }
    *got = w;
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
