// live_id.cpp -- the reference's examples/cpp/live-id.cpp on the GPU path:
//   live_id --index a.wav b.wav ... [--dump db.cereal] --search q1.wav q2.wav ...
//   live_id --db db.cereal --search q1.wav ...
//   ... --batch: run the queries as one batch (search_batched: same output)
//   ... --votes: additionally print AnnStorage-style voting results ("=# <name> <cnt> <offset>")
// index() learns the filters from the indexed tracks (or set HPFW_PREPARE_KEEP_FILTERS=1 to keep
// those of cache/filters.cereal, which the collector's load() reads in the constructor); --dump /
// --db write and read the database in MemoryStorage's cereal format (storage.h:67-86).
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include <hpfw/gpu/live_song_id.h>

int main(int argc, char **argv)
{
    std::vector<std::string> to_index, to_search;
    std::vector<std::string> *cur = nullptr;
    std::string dump, db;
    bool votes = false, batch = false;
    for (int i = 1; i < argc; ++i) {
        if (!std::strcmp(argv[i], "--votes")) votes = true;
        else if (!std::strcmp(argv[i], "--batch")) batch = true;
        else if (!std::strcmp(argv[i], "--dump") && i + 1 < argc) dump = argv[++i];
        else if (!std::strcmp(argv[i], "--db") && i + 1 < argc) db = argv[++i];
        else if (!std::strcmp(argv[i], "--index")) cur = &to_index;
        else if (!std::strcmp(argv[i], "--search")) cur = &to_search;
        else if (cur) cur->push_back(argv[i]);
    }
    if (to_index.empty() && db.empty()) {
        std::cerr << "usage: live_id (--index a.wav b.wav ... [--dump file] | --db file) --search q1.wav ..." << std::endl;
        return 2;
    }
    try {
        hpfw::LiveSongIdentification<> liveid;
        if (!db.empty()) liveid.get_storage().load(db);
        else liveid.index(to_index);
        if (!dump.empty()) liveid.get_storage().save(dump);
        if (batch)
            liveid.search_batched(to_search);
        else
            liveid.search(to_search);
        if (votes)
            for (const auto &q : to_search) {
                const auto r = liveid.get_storage().find_votes(liveid.get_collector().calc_hashprint(q));
                std::cout << "=# " << r.filename << " " << r.cnt << " " << r.offset << std::endl;
            }
    } catch (const std::exception &e) {
        std::cerr << "live_id: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
