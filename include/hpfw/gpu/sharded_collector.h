// sharded_collector.h -- hpfw::ShardedGpuCollector: ParallelCollector<HashprintHandle<uint64_t, CQT<>, 20, 80>,
// DriveCache> (reference include/hpfw/core/parallel_collector.h:16-140) over the GPUs of one node, usable as the
// `Collector` template argument of LiveSongIdentification (reference live_song_id.h:19-21) like GpuCollector.
//
// prepare() shards the files per device: every shard reads its block of files with its own host threads, computes
// the spectrograms (cached under cache/spectros/), adds the frame covariances to its accum_cov; ONE RCCL all-reduce
// sums accum_cov over the devices, the filters are solved once and go to every shard, and every shard hashes the
// spectrograms it kept in its HBM.  With ShardedGpuStorage as the Storage, LiveSongIdentification::index and
// ::search run on all the GPUs without Python.  Header-only over libhpfw_gpu_multi.so (include/hpfw_gpu_multi.h).
// Placement: HPFW_GPU_DEVICES="0,1,...", default every visible device.
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../hpfw_gpu_multi.h"

namespace hpfw {

class ShardedGpuCollector {
public:
    using Hashprint = std::vector<uint64_t>; // hashprint_handle.h:70

    struct FilenameFingerprintPair { // parallel_collector.h:26-35
        std::string filename;
        Hashprint fingerprint;
    };

    ShardedGpuCollector() { check(hpfw_gpu_group_create_env(&g_), "ShardedGpuCollector"); }
    explicit ShardedGpuCollector(const std::vector<int> &devices)
    {
        check(hpfw_gpu_group_create(devices.data(), (int)devices.size(), &g_), "ShardedGpuCollector");
    }
    ~ShardedGpuCollector() { hpfw_gpu_group_destroy(g_); }
    ShardedGpuCollector(const ShardedGpuCollector &) = delete;
    ShardedGpuCollector &operator=(const ShardedGpuCollector &) = delete;

    int shards() const { return hpfw_gpu_group_size(g_); }

    /// parallel_collector.h:48-52
    auto prepare(const std::vector<std::string> &filenames) -> std::vector<FilenameFingerprintPair>
    {
        std::vector<const char *> names;
        for (const auto &f : filenames) names.push_back(f.c_str());
        int got = 0;
        FilenameHashprintPair *res = hpfw_gpu_group_prepare(g_, names.data(), (int)names.size(), &got);
        if (!res) throw std::runtime_error(std::string("hpfw::ShardedGpuCollector::prepare: ") + hpfw_gpu_last_error());
        std::vector<FilenameFingerprintPair> out;
        for (int i = 0; i < got; ++i) out.push_back({res[i].filename, Hashprint(res[i].hashprint, res[i].hashprint + res[i].hp_size)});
        prepare_result_free(res, got);
        if (got == 0 && !filenames.empty())
            throw std::runtime_error(std::string("hpfw::ShardedGpuCollector::prepare: ") + hpfw_gpu_last_error());
        return out;
    }

    /// parallel_collector.h:54-59
    auto calc_hashprint(const std::string &filename) const -> Hashprint
    {
        int size = 0;
        uint64_t *hp = hpfw_gpu_group_calc_hashprint(g_, filename.c_str(), &size);
        if (!hp) throw std::runtime_error("hpfw::ShardedGpuCollector::calc_hashprint('" + filename + "'): " + hpfw_gpu_last_error());
        Hashprint out(hp, hp + size);
        calc_hashprint_result_free(hp);
        return out;
    }

    void save() const { (void)hpfw_gpu_group_save(g_, cache_.c_str()); }                       // parallel_collector.h:61-66
    void load() { check(hpfw_gpu_group_load(g_, cache_.c_str()), "ShardedGpuCollector::load"); } // parallel_collector.h:68-73
    void set_cache_dir(const std::string &dir) { cache_ = dir; }

private:
    static void check(int rc, const char *what)
    {
        if (rc != 0) throw std::runtime_error(std::string("hpfw::") + what + ": " + hpfw_gpu_last_error());
    }
    hpfw_gpu_group *g_ = nullptr;
    std::string cache_ = "cache/"; // parallel_collector.h:38
};

} // namespace hpfw
