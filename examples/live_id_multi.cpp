// live_id_multi.cpp -- live_id.cpp with the index sharded over the GPUs of the node:
//   HPFW_GPU_DEVICES=0,1,2,3,4,5,6,7 live_id_multi --index a.wav b.wav ... --search q1.wav q2.wav ... [--batch]
// LiveSongIdentification<GpuCollector, ShardedGpuStorage>: the collector extracts on one device (set
// HPFW_GPU_DEVICE), the storage shards the tracks per file and exchanges per-shard top-k lists with one RCCL
// all-gather per search (include/hpfw/gpu/sharded_storage.h).  Same stdout lines as live_id.
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include <hpfw/gpu/live_song_id.h>
#include <hpfw/gpu/sharded_storage.h>

int main(int argc, char **argv)
{
    std::vector<std::string> to_index, to_search;
    std::vector<std::string> *cur = nullptr;
    bool batch = false;
    for (int i = 1; i < argc; ++i) {
        if (!std::strcmp(argv[i], "--batch")) batch = true;
        else if (!std::strcmp(argv[i], "--index")) cur = &to_index;
        else if (!std::strcmp(argv[i], "--search")) cur = &to_search;
        else if (cur) cur->push_back(argv[i]);
    }
    if (to_index.empty()) {
        std::cerr << "usage: [HPFW_GPU_DEVICES=0,1,...] live_id_multi --index a.wav b.wav ... --search q1.wav ... [--batch]" << std::endl;
        return 2;
    }
    try {
        hpfw::LiveSongIdentification<hpfw::GpuCollector, hpfw::db::ShardedGpuStorage<hpfw::GpuCollector>> liveid;
        std::cerr << "shards: " << liveid.get_storage().shards() << std::endl;
        liveid.index(to_index);
        if (batch)
            liveid.search_batched(to_search);
        else
            liveid.search(to_search);
    } catch (const std::exception &e) {
        std::cerr << "live_id_multi: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
