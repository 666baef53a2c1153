"""LiveSongIdentification in Python: the reference's C++ class
(include/hpfw/audioproblems/live-song-id/live_song_id.h:19-60) and its notebook
(examples/python/liveid.ipynb cells 2-12) over the GPU collector and the GPU scan.

index(files)  = storage.build(collector.prepare(files))           live_song_id.h:31-33
search(files) = per query: calc_hashprint -> find -> report       live_song_id.h:35-54
top(files, k) = the notebook's "ten best tracks" per query        liveid.ipynb cell 9
"""
import os
from typing import List, Sequence, Tuple

import numpy as np

from . import _lib
from .collector import ParallelCollector


class LiveSongIdentification:
    def __init__(self, cache: str = "", device: int = 0):
        self.collector = ParallelCollector()
        self.collector.load(cache)                       # the constructor loads the cache, live_song_id.h:24
        self._cache = cache
        self._gpu = _lib.Gpu(device)
        self.names: List[str] = []

    def close(self):
        self.collector.save(self._cache)                 # the destructor saves it, live_song_id.h:28
        self._gpu.close()

    def build(self, hashprints: Sequence[Tuple[np.ndarray, str]]):
        """MemoryStorage::build (storage.h:21-25) from prepare()'s (array, name) pairs"""
        self.names = [name for _, name in hashprints]
        self._gpu.index_clear()
        if self.names:
            off = np.zeros(len(hashprints) + 1, np.int64)
            np.cumsum([hp.size for hp, _ in hashprints], out=off[1:])
            flat = np.concatenate([hp for hp, _ in hashprints]) if off[-1] else np.zeros(1, np.uint64)
            self._gpu.index_add(flat, off)

    def index(self, filenames: Sequence[str]):
        self.build(self.collector.prepare(list(filenames)))

    def top(self, filenames: Sequence[str], k: int = 10):
        """per query (label, [(distance, name, offset) x <= k]) ordered by (distance, position in the
        database); None in place of the list for a file that yields no hashprint"""
        hps = self.collector.calc_hashprints(list(filenames))
        good = [i for i, (hp, _) in enumerate(hps) if hp is not None and hp.size]
        out = [(f, None) for f in filenames]
        if good and self.names:
            off = np.zeros(len(good) + 1, np.int64)
            np.cumsum([hps[i][0].size for i in good], out=off[1:])
            hits = self._gpu.search_topk(np.concatenate([hps[i][0] for i in good]), off, k)
            for row, i in zip(hits, good):
                out[i] = (filenames[i], [(int(h["dist"]), self.names[int(h["clip"])], int(h["offset"]))
                                         for h in row if h["clip"] != 0xFFFFFFFF])
        return out

    def search(self, filenames: Sequence[str]):
        """prints what the reference prints (live_song_id.h:38,47-48,53); returns (wrong, accuracy)"""
        wrong = 0
        for label, best in self.top(filenames, 1):
            print("=> Finding", label)
            if not best:
                continue
            dist, name, offset = best[0]
            if os.path.splitext(os.path.basename(name))[0] not in label:
                wrong += 1
            print(f"=> {name} {dist} {offset}\n")
        acc = 1 - wrong / float(len(filenames)) if filenames else 1.0
        print(f"=> {wrong} {acc:g}")
        return wrong, acc
