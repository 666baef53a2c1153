// gpu_storage.h -- hpfw::db::GpuStorage<Collector>: the MI355X counterpart of
// hpfw::db::MemoryStorage<Collector> (reference include/hpfw/audioproblems/live-song-id/storage.h:8-92):
// build() keeps the hashprints in HBM, find() runs the exhaustive sliding Hamming scan on the GPU and
// returns SearchResult{filename, cnt, offset} of the first strict minimum in database order.
// find_topk() adds the notebook's "10 best tracks" (examples/python/liveid.ipynb cell 9), ordered by
// (distance, position in the database).
#pragma once

#include <cstdint>
#include <cstdlib>
#include <fstream>
#include <limits>
#include <optional>
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

    /// the device: HPFW_GPU_DEVICE (the collector reads the same variable), default 0; ShardedGpuStorage
    /// (sharded_storage.h) is the storage over several devices
    GpuStorage() : GpuStorage(env_device()) {}
    explicit GpuStorage(int device)
    {
        if (hpfw_gpu_create(device, &h_) != 0) throw std::runtime_error(std::string("hpfw::db::GpuStorage: ") + hpfw_gpu_last_error());
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

    /// find() for many queries in one scan of the database (same result per query as find();
    /// an empty hashprint yields the empty result of storage.h:28)
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
        check(hpfw_gpu_search_topk(h_, all.data(), off.data(), (int64_t)which.size(), 1, hits.data()));
        for (size_t q = 0; q < which.size(); ++q)
            if (hits[q].clip != 0xffffffffu) out[which[q]] = {names_[hits[q].clip], (size_t)hits[q].dist, (int64_t)hits[q].offset};
        return out;
    }

    size_t size() const { return names_.size(); }

    /// AnnStorage<Collector>::find (reference annoy_storage.h:41-63) with exact nearest neighbours in
    /// place of the Annoy forest: per query position the 5 nearest 64-hashprint windows vote
    /// 1 / (distance + 1) for (track, offset); returns the reference's {filename, cnt, offset} of
    /// the winning bucket ({"", 0, 0} when nothing votes).
    struct VoteResult { // annoy_storage.h:16-20
        std::string filename;
        float cnt;
        int64_t offset;
    };
    auto find_votes(const typename Collector::Hashprint &hp) const -> VoteResult
    {
        if (hp.empty() || names_.empty()) return {"", 0.0f, 0};
        const int64_t q_off[2] = {0, (int64_t)hp.size()};
        hpfw_vote v;
        check(hpfw_gpu_search_votes(h_, hp.data(), q_off, 1, &v));
        if (v.clip == 0xffffffffu) return {"", 0.0f, 0};
        return {names_[v.clip], v.cnt, v.offset};
    }

    /// storage.h:67-75: the cereal BinaryOutputArchive image of std::vector<FilenameFingerprintPair>
    /// (parallel_collector.h:26-35): u64 count, then per entry u64 length + bytes of the filename and
    /// u64 length + that many u64 hashprints.  The FORMAT is the reference's: a dump written by either side loads on
    /// the other.  Whether queries hashed on one side match a database hashed on the other is a separate question:
    /// it needs the same filters.cereal and as far as essentia's unverifiable conventions agree (INTEGRATION.md,
    /// "What a switch changes"; hpfw_gpu_set_conventions).
    auto save(const std::optional<std::string> &filename) const -> std::string
    {
        const auto dump_name = filename.value_or("db/dump.cereal");
        std::vector<int64_t> off(names_.size() + 1, 0);
        check(hpfw_gpu_index_get(h_, off.data(), nullptr, 0));
        std::vector<uint64_t> all((size_t)off.back());
        check(hpfw_gpu_index_get(h_, off.data(), all.empty() ? &dummy_ : all.data(), (int64_t)all.size()));
        std::ofstream os(dump_name, std::ios::binary);
        if (!os) throw std::runtime_error("hpfw::db::GpuStorage: cannot write " + dump_name);
        put(os, (uint64_t)names_.size());
        for (size_t i = 0; i < names_.size(); ++i) {
            put(os, (uint64_t)names_[i].size());
            os.write(names_[i].data(), (std::streamsize)names_[i].size());
            const uint64_t n = (uint64_t)(off[i + 1] - off[i]);
            put(os, n);
            os.write(reinterpret_cast<const char *>(all.data() + off[i]), (std::streamsize)(n * 8));
        }
        if (!os) throw std::runtime_error("hpfw::db::GpuStorage: write failed: " + dump_name);
        return dump_name;
    }

    /// storage.h:78-86
    auto load(const std::string &dump_name) -> GpuStorage &
    {
        std::ifstream is(dump_name, std::ios::binary);
        if (!is) throw std::runtime_error("hpfw::db::GpuStorage: cannot read " + dump_name);
        struct Pair {
            std::string filename;
            std::vector<uint64_t> fingerprint;
        };
        std::vector<Pair> db((size_t)get(is, dump_name));
        for (Pair &p : db) {
            p.filename.resize((size_t)get(is, dump_name));
            is.read(p.filename.data(), (std::streamsize)p.filename.size());
            p.fingerprint.resize((size_t)get(is, dump_name));
            is.read(reinterpret_cast<char *>(p.fingerprint.data()), (std::streamsize)(p.fingerprint.size() * 8));
            if (!is) throw std::runtime_error("hpfw::db::GpuStorage: truncated dump " + dump_name);
        }
        build(db);
        return *this;
    }

private:
    static int env_device()
    {
        const char *e = std::getenv("HPFW_GPU_DEVICE");
        return e ? std::atoi(e) : 0;
    }
    static void put(std::ostream &os, uint64_t v) { os.write(reinterpret_cast<const char *>(&v), 8); }
    static uint64_t get(std::istream &is, const std::string &name)
    {
        uint64_t v = 0;
        is.read(reinterpret_cast<char *>(&v), 8);
        if (!is || v > (uint64_t(1) << 40)) throw std::runtime_error("hpfw::db::GpuStorage: malformed dump " + name);
        return v;
    }
    static void check(int rc)
    {
        if (rc != 0) throw std::runtime_error(std::string("hpfw::db::GpuStorage: ") + hpfw_gpu_last_error());
    }
    hpfw_gpu *h_ = nullptr;
    std::vector<std::string> names_;
    mutable uint64_t dummy_ = 0;
};

} // namespace hpfw::db
