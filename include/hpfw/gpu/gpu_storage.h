// gpu_storage.h -- hpfw::db::GpuStorage<Collector>: the MI355X counterpart of
// hpfw::db::MemoryStorage<Collector> (reference include/hpfw/audioproblems/live-song-id/storage.h:8-92):
// build() keeps the hashprints in HBM, find() runs the exhaustive sliding Hamming scan on the GPU and
// returns SearchResult{filename, cnt, offset} of the first strict minimum in database order.
// find_topk() adds the notebook's "10 best tracks" (examples/python/liveid.ipynb cell 9), ordered by
// (distance, position in the database).
#pragma once

#include <cstdint>
#include <limits>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../hpfw_gpu.h"

namespace hpfw::db {

template <typename Collector>
class GpuStorage {
public:
    struct SearchResult { // storage.h:11-15
        std::string filename;
        size_t cnt;
        int64_t offset;
    };

    GpuStorage()
    {
        if (hpfw_gpu_create(0, &h_) != 0) throw std::runtime_error(std::string("hpfw::db::GpuStorage: ") + hpfw_gpu_last_error());
    }
    ~GpuStorage() { hpfw_gpu_destroy(h_); }
    GpuStorage(const GpuStorage &) = delete;
    GpuStorage &operator=(const GpuStorage &) = delete;

    /// storage.h:21-25; accepts any range of Collector::FilenameFingerprintPair (vector, tbb::concurrent_vector)
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
        check(hpfw_gpu_index_clear(h_));
        if (!names_.empty()) check(hpfw_gpu_index_add(h_, all.empty() ? &dummy_ : all.data(), off.data(), (int64_t)names_.size()));
    }

    /// storage.h:27-64
    auto find(const typename Collector::Hashprint &hp) const -> SearchResult
    {
        auto top = find_topk(hp, 1);
        if (top.empty()) return {"", std::numeric_limits<size_t>::max(), 0}; // storage.h:28
        return top[0];
    }

    auto find_topk(const typename Collector::Hashprint &hp, int k) const -> std::vector<SearchResult>
    {
        std::vector<SearchResult> out;
        if (hp.empty() || names_.empty()) return out;
        const int64_t q_off[2] = {0, (int64_t)hp.size()};
        std::vector<hpfw_hit> hits((size_t)k);
        check(hpfw_gpu_search_topk(h_, hp.data(), q_off, 1, k, hits.data()));
        for (const hpfw_hit &hit : hits) {
            if (hit.clip == 0xffffffffu) break;
            out.push_back({names_[hit.clip], (size_t)hit.dist, (int64_t)hit.offset});
        }
        return out;
    }

    size_t size() const { return names_.size(); }

private:
    static void check(int rc)
    {
        if (rc != 0) throw std::runtime_error(std::string("hpfw::db::GpuStorage: ") + hpfw_gpu_last_error());
    }
    hpfw_gpu *h_ = nullptr;
    std::vector<std::string> names_;
    uint64_t dummy_ = 0;
};

} // namespace hpfw::db
