// live_song_id.h -- hpfw::LiveSongIdentification with the GPU collector and storage as defaults.
// Same contract as the reference class (include/hpfw/audioproblems/live-song-id/live_song_id.h:19-60):
// the constructor loads the collector's cache, the destructor saves it, index() = build(prepare()),
// search() prints "=> Finding <file>", "=> <name> <cnt> <offset>" and a final
// "=> <wrong> <accuracy>" line and counts a result as wrong when the query path does not contain
// the returned name (live_song_id.h:38-53).  Any Collector / Storage pair with the reference's
// interfaces can be plugged in, as in the reference.
#pragma once

#include <filesystem>
#include <iostream>
#include <string>
#include <vector>

#include "gpu_collector.h"
#include "gpu_storage.h"

namespace hpfw {

template <typename Collector = GpuCollector, typename Storage = db::GpuStorage<GpuCollector>>
class LiveSongIdentification {
public:
    LiveSongIdentification() { collector.load(); }
    ~LiveSongIdentification() { collector.save(); }

    void index(const std::vector<std::string> &filenames) { storage.build(collector.prepare(filenames)); }

    auto search(const std::vector<std::string> &filenames)
    {
        uint16_t wrong = 0;
        for (const auto &query : filenames) {
            std::cout << "=> Finding " << query << std::endl;
            try {
                const auto res = storage.find(collector.calc_hashprint(query));
                const auto name = std::filesystem::path(res.filename).stem().string();
                if (query.find(name) == std::string::npos) {
                    std::cerr << "Wrong result for '" << query << "': got '" << name << "'" << std::endl;
                    ++wrong;
                }
                std::cout << "=> " << res.filename << " " << res.cnt << " " << res.offset << std::endl << std::endl;
            } catch (const std::exception &e) {
                std::cerr << "Error finding '" << query << "': " << e.what() << std::endl;
            }
        }
        std::cout << "=> " << wrong << " " << 1 - wrong / float(filenames.size()) << std::endl;
    }

    /// search() with the queries read, transformed and scanned as one batch (Collector::calc_hashprints,
    /// Storage::find_batch): the same lines on stdout, printed after the batch has run
    auto search_batched(const std::vector<std::string> &filenames)
    {
        uint16_t wrong = 0;
        const auto hps = collector.calc_hashprints(filenames);
        const auto found = storage.find_batch(hps);
        for (size_t i = 0; i < filenames.size(); ++i) {
            const auto &query = filenames[i];
            std::cout << "=> Finding " << query << std::endl;
            if (hps[i].empty()) {
                std::cerr << "Error finding '" << query << "': no hashprint" << std::endl;
                continue;
            }
            const auto &res = found[i];
            const auto name = std::filesystem::path(res.filename).stem().string();
            if (query.find(name) == std::string::npos) {
                std::cerr << "Wrong result for '" << query << "': got '" << name << "'" << std::endl;
                ++wrong;
            }
            std::cout << "=> " << res.filename << " " << res.cnt << " " << res.offset << std::endl << std::endl;
        }
        std::cout << "=> " << wrong << " " << 1 - wrong / float(filenames.size()) << std::endl;
    }

    Collector &get_collector() { return collector; }
    Storage &get_storage() { return storage; }

private:
    Collector collector;
    Storage storage;
};

} // namespace hpfw
