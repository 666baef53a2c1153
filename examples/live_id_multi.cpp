// live_id_multi.cpp -- live_id.cpp with the index sharded over the GPUs of the node:
//   HPFW_GPU_DEVICES=0,1,2,3,4,5,6,7 live_id_multi --index a.wav b.wav ... --search q1.wav q2.wav ... [--batch]
// LiveSongIdentification<ShardedGpuCollector, ShardedGpuStorage>: index() reads, transforms and hashes the files on
// all the devices (one all-reduce of the frame covariance before the filters are solved), the storage shards the
// tracks per file and exchanges per-shard top-k lists with one RCCL all-gather per search
// (include/hpfw/gpu/sharded_collector.h, sharded_storage.h).  Same stdout lines as live_id.
//   --one-collector: GpuCollector (one device, HPFW_GPU_DEVICE) with the sharded storage
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include <hpfw/gpu/live_song_id.h>
#include <hpfw/gpu/sharded_collector.h>
#include <hpfw/gpu/sharded_storage.h>

template <class LiveId>
static int run(const std::vector<std::string> &to_index, const std::vector<std::string> &to_search, bool batch)
{
    LiveId liveid;
    std::cerr << "shards: " << liveid.get_storage().shards() << std::endl;
    liveid.index(to_index);
    if constexpr (requires { liveid.get_collector().calc_hashprints(to_search); }) {
        if (batch) {
            liveid.search_batched(to_search);
            return 0;
        }
    }
    liveid.search(to_search);
    return 0;
}

int main(int argc, char **argv)
{
    std::vector<std::string> to_index, to_search;
    std::vector<std::string> *cur = nullptr;
    bool batch = false, one_collector = false;
    for (int i = 1; i < argc; ++i) {
        if (!std::strcmp(argv[i], "--batch")) batch = true;
        else if (!std::strcmp(argv[i], "--one-collector")) one_collector = true;
        else if (!std::strcmp(argv[i], "--index")) cur = &to_index;
        else if (!std::strcmp(argv[i], "--search")) cur = &to_search;
        else if (cur) cur->push_back(argv[i]);
    }
    if (to_index.empty()) {
        std::cerr << "usage: [HPFW_GPU_DEVICES=0,1,...] live_id_multi --index a.wav b.wav ... --search q1.wav ... [--batch]" << std::endl;
        return 2;
    }
    try {
        using Storage = hpfw::db::ShardedGpuStorage<hpfw::GpuCollector>;
        if (one_collector) return run<hpfw::LiveSongIdentification<hpfw::GpuCollector, Storage>>(to_index, to_search, batch);
        return run<hpfw::LiveSongIdentification<hpfw::ShardedGpuCollector, hpfw::db::ShardedGpuStorage<hpfw::ShardedGpuCollector>>>(
            to_index, to_search, batch);
    } catch (const std::exception &e) {
        std::cerr << "live_id_multi: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
