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
usage: python profiles/summarize.py r02 [gpurun_out/prof] [clips_per_launch=1000] [output dir]"""
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
    # summaries go next to this script, or to the directory given as the 4th argument (the GPU box writes
    # them under gpurun_out/, the only directory that travels back)
    here = sys.argv[4] if len(sys.argv) > 4 else os.path.dirname(os.path.abspath(__file__))
    os.makedirs(here, exist_ok=True)
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
        traffic = {"clips_per_launch_profiled": clips, "source": f"{tag}_pmc.json",
                   "formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024 / clips, launches of the largest grid"}
        for key, prefixes in (("project_mfma_hbm_bytes_per_clip", ("project_kernel<true", "project_kernel")),
                              ("project_q_hbm_bytes_per_clip", ("hashprint_q_kernel<true", "hashprint_q_kernel", "project_q_kernel")),
                              ("fwd_cols_hbm_bytes_per_clip", ("fwd_cols_q3_kernel", "fwd_cols_q_kernel")),
                              ("fwd_rows_hbm_bytes_per_clip", ("fwd_rows2_kernel", "fwd_rows_kernel"))):
            # prefixes in order of preference; of the kernels one prefix matches, the launch of the largest grid (round 3 took
            # the first key in sorted order, hashprint_q_kernel<false> -- 256-clip launches of the bench's side sections --
            # and divided by 1000 clips: 0.60 MB per clip where the 1000-clip launches of <true> read 2.34 MB)
            pk = {}
            for px in prefixes:
                cand = [v for k, v in pmc.items() if k.startswith(px) and "hbm_bytes_per_launch" in v]
                if cand:
                    pk = max(cand, key=lambda v: v.get("grid_work_items", 0))
                    break
            if "hbm_bytes_per_launch" in pk:
                traffic[key] = pk["hbm_bytes_per_launch"] / clips
                traffic[key.replace("_hbm_bytes_per_clip", "_grid_work_items")] = pk.get("grid_work_items")
        if len(traffic) > 3:
            json.dump(traffic, open(os.path.join(here, "traffic.json"), "w"), indent=1)
    # ---- calibration of FETCH_SIZE / WRITE_SIZE on known byte counts (tools/fetch_calib.bin streams 1 GiB per kernel)
    calib = {}
    for which, counter, kern in (("calib_fetch", "FETCH_SIZE", "calib_read"), ("calib_write", "WRITE_SIZE", "calib_write")):
        files = glob.glob(os.path.join(src, which, "**", "*_counter_collection.csv"), recursive=True)
        if not files:
            continue
        acc = defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(files[0])):
            if kern in r["Kernel_Name"] and r["Counter_Name"] == counter:
                width = {"b2": 2, "b4": 4, "b8": 8, "b16": 16}[r["Kernel_Name"].split("<")[1].split(">")[0].strip()]
                a = acc[width]
                a[0] += float(r["Counter_Value"])
                a[1] += 1
        for width, (tot, n) in sorted(acc.items()):
            calib.setdefault(counter, {})[f"{width}_bytes_per_lane"] = {
                "counter_KiB_per_launch": tot / n, "bytes_streamed": 1 << 30,
                "counter_bytes_over_true_bytes": tot / n * 1024.0 / (1 << 30)}
    if calib:
        calib["note"] = ("coalesced streaming of 1 GiB (past the 256 MiB Infinity Cache), 2 launches each; a ratio of 0.5 for "
                         "FETCH_SIZE is the gfx950 half-counting of MI355X_MICROARCH.md; bytes = counter / ratio")
        json.dump(calib, open(os.path.join(here, f"{tag}_counter_calibration.json"), "w"), indent=1, sort_keys=True)
    # ---- SQ counters (separate passes sq1..sq5), launches of the largest grid of every kernel
    sq = {}
    for d in sorted(glob.glob(os.path.join(src, "sq*"))):
        files = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)
        if not files:
            continue
        acc = defaultdict(lambda: defaultdict(lambda: defaultdict(lambda: [0.0, 0])))
        for r in csv.DictReader(open(files[0])):
            if "hpfw::" in r["Kernel_Name"]:
                a = acc[short(r["Kernel_Name"])][int(r["Grid_Size"])][r["Counter_Name"]]
                a[0] += float(r["Counter_Value"])
                a[1] += 1
        for k, by_grid in acc.items():
            grid = max(by_grid)
            for cname, (tot, n) in by_grid[grid].items():
                sq.setdefault(k, {"grid_work_items": grid})[cname] = tot / n
    for k, d in sq.items():
        busy, wc = d.get("SQ_BUSY_CYCLES"), d.get("SQ_WAVE_CYCLES")
        der = {}
        # SQ_BUSY_CYCLES sums the 32 shader engines' busy cycles, the per-SIMD counters the 1024 SIMDs':
        # x / (32 * SQ_BUSY_CYCLES) is the fraction of the kernel's cycles a SIMD spent on x
        if busy and "SQ_VALU_MFMA_BUSY_CYCLES" in d:
            der["mfma_busy_fraction"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (32.0 * busy)
        if busy and "SQ_INSTS_VALU" in d:
            der["valu_issue_fraction"] = d["SQ_INSTS_VALU"] * 2.0 / (32.0 * busy)   # 2 cycles per wave64 VALU instruction
        if wc:
            for name, key in (("active", "SQ_ACTIVE_INST_ANY"), ("parked_waitcnt_or_barrier", "SQ_WAIT_ANY"),
                              ("issue_stalled", "SQ_WAIT_INST_ANY"), ("lds_issue_stalled", "SQ_WAIT_INST_LDS")):
                if key in d:
                    der["wave_cycles_" + name] = d[key] / wc
            if busy:
                der["resident_waves_per_simd"] = wc * 4.0 / (32.0 * busy)   # SQ_WAVE_CYCLES counts in units of 4 cycles
        if d.get("SQ_LDS_IDX_ACTIVE"):
            der["lds_bank_conflict_fraction"] = d.get("SQ_LDS_BANK_CONFLICT", 0.0) / d["SQ_LDS_IDX_ACTIVE"]
        d["derived"] = der
    if sq:
        json.dump(sq, open(os.path.join(here, f"{tag}_sq.json"), "w"), indent=1, sort_keys=True)
    print(open(os.path.join(here, f"{tag}_kernel_stats.csv")).read() if rows else "no trace")
    print(json.dumps(pmc, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
