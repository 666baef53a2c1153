#!/usr/bin/env python3
"""Condense rocprofv3 output (gpurun_out/prof/{trace,fetch,write}, made by tools/profile_round.sh)
into the tracked summaries:
  profiles/<tag>_kernel_stats.csv   per kernel AND launch size: calls / average / min / max / total
                                    (bench.py also launches the kernels on one and on 32 clips in its
                                    streaming section; rocprofv3's own --stats averages over all sizes)
  profiles/<tag>_pmc.json           FETCH_SIZE / WRITE_SIZE per kernel, averaged per launch of the largest size
  profiles/traffic.json             HBM bytes per clip of the dominant kernel, read by bench.py
HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: the counters are in KiB, and on gfx950 FETCH_SIZE
reports half of the bytes of a wide coalesced read (MI355X_MICROARCH.md, HBM section).
usage: python profiles/summarize.py r01 [gpurun_out/prof] [clips_per_launch=1000]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("void ", "")
    head = name.split("(")[0]
    if "<" in head:  # templated kernels: keep the first template argument (size class / group list)
        head = head.split("<")[0] + "<" + name.split("<", 1)[1].split(">")[0].split(",")[0].strip() + ">"
    return head.replace("hpfw::", "")


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    src = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/prof"
    clips = float(sys.argv[3]) if len(sys.argv) > 3 else 1000.0   # clips per back-end launch in the profiled run
    here = os.path.dirname(os.path.abspath(__file__))
    traces = glob.glob(os.path.join(src, "trace", "**", "*_kernel_trace.csv"), recursive=True)
    rows = []
    if traces:
        groups = defaultdict(list)
        for r in csv.DictReader(open(traces[0])):
            if "hpfw::" not in r["Kernel_Name"]:
                continue
            grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
            groups[(short(r["Kernel_Name"]), grid)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        all_ns = sum(sum(v) for v in groups.values())
        for (k, grid), v in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
            rows.append([k, grid, len(v), sum(v), round(sum(v) / len(v)), min(v), max(v), f"{100.0 * sum(v) / all_ns:.3f}"])
        with open(os.path.join(here, f"{tag}_kernel_stats.csv"), "w") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "grid_work_items", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct_of_hpfw_gpu_time"])
            w.writerows(rows)
    pmc = {}
    for which, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        files = glob.glob(os.path.join(src, which, "**", "*_counter_collection.csv"), recursive=True)
        if not files:
            continue
        acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
        for r in csv.DictReader(open(files[0])):
            if "hpfw::" in r["Kernel_Name"] and r["Counter_Name"] == counter:
                a = acc[short(r["Kernel_Name"])][int(r["Grid_Size"])]
                a[0] += float(r["Counter_Value"])
                a[1] += 1
        for k, by_grid in acc.items():
            grid = max(by_grid)                                   # the full-size launches
            tot, n = by_grid[grid]
            pmc.setdefault(k, {})[counter + "_KiB_per_launch"] = tot / n
            pmc[k]["launches_" + which] = n
            pmc[k]["grid_work_items"] = grid
    for k, d in pmc.items():
        f_ = d.get("FETCH_SIZE_KiB_per_launch")
        w_ = d.get("WRITE_SIZE_KiB_per_launch")
        if f_ is not None and w_ is not None:
            d["hbm_bytes_per_launch"] = (2.0 * f_ + w_) * 1024.0
    if pmc:
        json.dump(pmc, open(os.path.join(here, f"{tag}_pmc.json"), "w"), indent=1, sort_keys=True)
        pk = next((v for k, v in sorted(pmc.items()) if k.startswith("project_kernel<true") or k == "project_kernel"), {})
        if "hbm_bytes_per_launch" in pk:
            json.dump({"project_mfma_hbm_bytes_per_clip": pk["hbm_bytes_per_launch"] / clips,
                       "clips_per_launch_profiled": clips, "source": f"{tag}_pmc.json",
                       "formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024 / clips, launches of the largest grid"},
                      open(os.path.join(here, "traffic.json"), "w"), indent=1)
    print(open(os.path.join(here, f"{tag}_kernel_stats.csv")).read() if rows else "no trace")
    print(json.dumps(pmc, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
