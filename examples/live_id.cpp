// live_id.cpp -- the reference's examples/cpp/live-id.cpp on the GPU path:
//   live_id --index a.wav b.wav ... --search q1.wav q2.wav ...
// Filters are read from cache/filters.cereal (the reference's own file format) by the collector's
// load() in the LiveSongIdentification constructor.
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include <hpfw/gpu/live_song_id.h>

int main(int argc, char **argv)
{
    std::vector<std::string> to_index, to_search;
    std::vector<std::string> *cur = nullptr;
    for (int i = 1; i < argc; ++i) {
        if (!std::strcmp(argv[i], "--index")) cur = &to_index;
        else if (!std::strcmp(argv[i], "--search")) cur = &to_search;
        else if (cur) cur->push_back(argv[i]);
    }
    if (to_index.empty()) {
        std::cerr << "usage: live_id --index a.wav b.wav ... --search q1.wav ..." << std::endl;
        return 2;
    }
    try {
        hpfw::LiveSongIdentification<> liveid;
        liveid.index(to_index);
        liveid.search(to_search);
    } catch (const std::exception &e) {
        std::cerr << "live_id: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
