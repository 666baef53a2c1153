// gpu_collector.h -- hpfw::GpuCollector: the MI355X counterpart of
// hpfw::ParallelCollector<HashprintHandle<uint64_t, CQT<>, 20, 80>, DriveCache>
// (reference include/hpfw/core/parallel_collector.h:16-140), usable as the `Collector` template
// argument of LiveSongIdentification (reference live_song_id.h:19-21).  Header-only; all work is
// done by libhpfw_gpu.so through the C-ABI (include/hpfw_gpu.h).
//
// Same public surface: Hashprint, FilenameFingerprintPair, prepare(), calc_hashprint(), save(),
// load().  Differences, all deliberate:
//   - audio files must be PCM16 WAV at 44.1 kHz (decode/resample are outside the accelerated path);
//   - prepare() learns the filters as the reference's preprocess() does (covariance of the frames
//     of every file, 64 leading eigenvectors; parallel_collector.h:82-112) unless the environment
//     variable HPFW_PREPARE_KEEP_FILTERS is set and filters were loaded; calc_hashprint() before any
//     filters exist throws instead of projecting with uninitialised filters (reference defect D-9);
//   - prepare() returns results in input order (the reference's order is racy, :129).
#pragma once

#include <cstdint>
#include <filesystem>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../hpfw_gpu.h"

namespace hpfw {

class GpuCollector {
public:
    using Hashprint = std::vector<uint64_t>; // hashprint_handle.h:70

    struct FilenameFingerprintPair { // parallel_collector.h:26-35
        std::string filename;
        Hashprint fingerprint;
    };

    GpuCollector() : c_(par_collector_new())
    {
        if (!c_) throw std::runtime_error(std::string("hpfw::GpuCollector: ") + hpfw_gpu_last_error());
    }
    ~GpuCollector() { par_collector_del(c_); }
    GpuCollector(const GpuCollector &) = delete;
    GpuCollector &operator=(const GpuCollector &) = delete;

    /// parallel_collector.h:48-52 (second pass only: spectrogram -> hashprint for every file)
    auto prepare(const std::vector<std::string> &filenames) -> std::vector<FilenameFingerprintPair>
    {
        std::vector<const char *> names;
        for (const auto &f : filenames) names.push_back(f.c_str());
        int got = 0;
        FilenameHashprintPair *res = par_collector_prepare(c_, names.data(), (int)names.size(), &got);
        if (!res) throw std::runtime_error(std::string("hpfw::GpuCollector::prepare: ") + hpfw_gpu_last_error());
        std::vector<FilenameFingerprintPair> out;
        for (int i = 0; i < got; ++i)
            out.push_back({res[i].filename, Hashprint(res[i].hashprint, res[i].hashprint + res[i].hp_size)});
        prepare_result_free(res, got);
        if (got == 0 && !filenames.empty())
            throw std::runtime_error(std::string("hpfw::GpuCollector::prepare: ") + hpfw_gpu_last_error());
        return out;
    }

    /// parallel_collector.h:54-59
    auto calc_hashprint(const std::string &filename) const -> Hashprint
    {
        int size = 0;
        uint64_t *hp = par_collector_calc_hashprint(c_, filename.c_str(), &size);
        if (!hp) throw std::runtime_error("hpfw::GpuCollector::calc_hashprint('" + filename + "'): " + hpfw_gpu_last_error());
        Hashprint out(hp, hp + size);
        calc_hashprint_result_free(hp);
        return out;
    }

    /// calc_hashprint for a list of files in one batched call (not in the reference): one entry per
    /// file in input order; a file that failed yields an empty hashprint
    auto calc_hashprints(const std::vector<std::string> &filenames) const -> std::vector<Hashprint>
    {
        std::vector<const char *> names;
        for (const auto &f : filenames) names.push_back(f.c_str());
        FilenameHashprintPair *res = par_collector_calc_hashprints(c_, names.data(), (int)names.size());
        if (!res) throw std::runtime_error(std::string("hpfw::GpuCollector::calc_hashprints: ") + hpfw_gpu_last_error());
        std::vector<Hashprint> out;
        for (size_t i = 0; i < names.size(); ++i)
            out.push_back(res[i].hashprint ? Hashprint(res[i].hashprint, res[i].hashprint + res[i].hp_size) : Hashprint());
        prepare_result_free(res, (int)names.size());
        return out;
    }

    void save() const { par_collector_save(c_, cache_.c_str()); } // parallel_collector.h:61-66
    void load() { par_collector_load(c_, cache_.c_str()); }       // parallel_collector.h:68-73
    void set_cache_dir(const std::string &dir) { cache_ = dir; }

private:
    hpfw_legacy_collector *c_;
    std::string cache_ = "cache/"; // parallel_collector.h:38
};

} // namespace hpfw
