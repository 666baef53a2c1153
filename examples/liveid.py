"""The reference's notebook (examples/python/liveid.ipynb) on the GPU path:

    python examples/liveid.py --index originals/*.wav --search slices/*.wav [--dump dump.pkl]

prepare() -> pickle dump of [(hashprint array, name)] (cells 4-5) -> ten best tracks per query by the
sliding Hamming scan (cell 9, here one batched scan in HBM instead of a process pool over Cython
loops) -> share of queries whose best track is contained in the query's name (cells 11-12)."""
import argparse
import os
import pickle
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hpfw_amd  # noqa: E402
from hpfw_amd.liveid import LiveSongIdentification  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--index", nargs="*", default=[])
ap.add_argument("--load", help="pickle written by --dump (cell 6)")
ap.add_argument("--dump", help="write prepare()'s result as a pickle (cell 5)")
ap.add_argument("--search", nargs="*", default=[])
ap.add_argument("--cache", default="")
args = ap.parse_args()

liveid = LiveSongIdentification(cache=args.cache)
if args.load:
    with open(args.load, "rb") as fp:
        hashprints = pickle.load(fp)
else:
    hashprints = liveid.collector.prepare(args.index)
if args.dump:
    with open(args.dump, "wb") as fp:
        pickle.dump(hashprints, fp)
liveid.build(hashprints)
ans = liveid.top(args.search, 10)
for label, top in ans:
    print("Finding ", label)
    print("   ", [(d, name) for d, name, _ in top] if top else "INVALID QUERY")
right = sum(1 for label, top in ans if top and top[0][1] in label)
print("accuracy", right / max(len(ans), 1))
liveid.close()
