// sharded_storage.h -- hpfw::db::ShardedGpuStorage<Collector>: MemoryStorage<Collector> (reference
// include/hpfw/audioproblems/live-song-id/storage.h:8-92) over the GPUs of one node, usable as the `Storage`
// template argument of LiveSongIdentification (reference live_song_id.h:19-21) exactly like GpuStorage.
//
// build() shards the tracks per audio file -- contiguous blocks of the input range, one per shard, each
// resident in its GPU's HBM under its global track ids; find() replicates the query, every shard scans its
// block and keeps its own top-k, ONE RCCL all-gather of the per-shard lists over xGMI follows, and the lists
// are merged by (distance, track id): the answer is the one GpuStorage gives on the unsharded index, for any
// number of shards.  Header-only; the work is done by libhpfw_gpu_multi.so (include/hpfw_gpu_multi.h).
//
// Placement: HPFW_GPU_DEVICES="0,1,2,3,4,5,6,7" (one ordinal per shard; default: every visible device), or
// the explicit constructor.
#pragma once

#include <cstdint>
#include <limits>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../hpfw_gpu_multi.h"

namespace hpfw::db {

template <typename Collector>
class ShardedGpuStorage {
public:
    struct SearchResult { // storage.h:11-15
        std::string filename;
        size_t cnt;
        int64_t offset;
    };

    ShardedGpuStorage() { check(hpfw_gpu_group_create_env(&g_)); }
    explicit ShardedGpuStorage(const std::vector<int> &devices) { check(hpfw_gpu_group_create(devices.data(), (int)devices.size(), &g_)); }
    ~ShardedGpuStorage() { hpfw_gpu_group_destroy(g_); }
    ShardedGpuStorage(const ShardedGpuStorage &) = delete;
    ShardedGpuStorage &operator=(const ShardedGpuStorage &) = delete;

    int shards() const { return hpfw_gpu_group_size(g_); }
    size_t size() const { return names_.size(); }

    /// storage.h:21-25; accepts any range of Collector::FilenameFingerprintPair
    template <typename Range>
    void build(Range &&hashprints)
    {
        names_.clear();
        std::vector<uint64_t> all;
        std::vector<int64_t> off{0};
        for (auto &p : hashprints) {
            names_.push_back(p.filename);
            all.insert(all.end(), p.fingerprint.begin(), p.fingerprint.end());
            off.push_back((int64_t)all.size());
        }
        check(hpfw_gpu_group_index_build(g_, all.empty() ? &dummy_ : all.data(), off.data(), (int64_t)names_.size()));
    }

    /// storage.h:27-64: the first strict minimum in database order
    auto find(const typename Collector::Hashprint &hp) const -> SearchResult
    {
        auto top = find_topk(hp, 1);
        if (top.empty()) return {"", std::numeric_limits<size_t>::max(), 0}; // storage.h:28
        return top[0];
    }

    /// the notebook's "k best tracks" (examples/python/liveid.ipynb cell 9), by (distance, position in the database)
    auto find_topk(const typename Collector::Hashprint &hp, int k) const -> std::vector<SearchResult>
    {
        std::vector<SearchResult> out;
        if (hp.empty() || names_.empty()) return out;
        const int64_t q_off[2] = {0, (int64_t)hp.size()};
        std::vector<hpfw_hit> hits((size_t)k);
        check(hpfw_gpu_group_search_topk(g_, hp.data(), q_off, 1, k, hits.data()));
        for (const hpfw_hit &hit : hits) {
            if (hit.clip == 0xffffffffu) break;
            out.push_back({names_[hit.clip], (size_t)hit.dist, (int64_t)hit.offset});
        }
        return out;
    }

    /// find() for many queries with one scan per shard and one all-gather
    auto find_batch(const std::vector<typename Collector::Hashprint> &hps) const -> std::vector<SearchResult>
    {
        const SearchResult none{"", std::numeric_limits<size_t>::max(), 0};
        std::vector<SearchResult> out(hps.size(), none);
        if (names_.empty()) return out;
        std::vector<uint64_t> all;
        std::vector<int64_t> off{0};
        std::vector<size_t> which;
        for (size_t i = 0; i < hps.size(); ++i) {
            if (hps[i].empty()) continue;
            all.insert(all.end(), hps[i].begin(), hps[i].end());
            off.push_back((int64_t)all.size());
            which.push_back(i);
        }
        if (which.empty()) return out;
        std::vector<hpfw_hit> hits(which.size());
        check(hpfw_gpu_group_search_topk(g_, all.data(), off.data(), (int64_t)which.size(), 1, hits.data()));
        for (size_t q = 0; q < which.size(); ++q)
            if (hits[q].clip != 0xffffffffu) out[which[q]] = {names_[hits[q].clip], (size_t)hits[q].dist, (int64_t)hits[q].offset};
        return out;
    }

private:
    static void check(int rc)
    {
        if (rc != 0) throw std::runtime_error(std::string("hpfw::db::ShardedGpuStorage: ") + hpfw_gpu_last_error());
    }
    hpfw_gpu_group *g_ = nullptr;
    std::vector<std::string> names_;
    mutable uint64_t dummy_ = 0;
};

} // namespace hpfw::db
