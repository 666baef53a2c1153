"""Average per-kernel counters out of rocprofv3 --pmc passes: python tools/pmc_collect.py out.json dir1 dir2 ... [--match substr]
(each dir holds the p_counter_collection.csv of one pass; launches of the largest grid per kernel only)"""
import collections, csv, json, sys
args = [a for a in sys.argv[1:] if not a.startswith("--match=")]
match = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--match=")]
out_path, dirs = args[0], args[1:]
out = {}
for d in dirs:
    rows = list(csv.DictReader(open(f"{d}/p_counter_collection.csv")))
    grid = collections.defaultdict(int)
    for r in rows:
        k = r["Kernel_Name"].split("(")[0]
        grid[k] = max(grid[k], int(r["Grid_Size"]))
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for r in rows:
        k = r["Kernel_Name"].split("(")[0]
        if match and not any(m in k for m in match):
            continue
        if int(r["Grid_Size"]) != grid[k]:
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
    for k, v in acc.items():
        e = out.setdefault(k, {"grid_work_items": grid[k]})
        e.update({c: x / cnt[(k, c)] for c, x in v.items()})
for k, e in out.items():
    dv = {}
    if "SQ_BUSY_CYCLES" in e and "SQ_VALU_MFMA_BUSY_CYCLES" in e:
        dv["mfma_busy_fraction"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (32 * e["SQ_BUSY_CYCLES"])
    if "SQ_BUSY_CYCLES" in e and "SQ_INSTS_VALU" in e:
        dv["valu_issue_fraction"] = 2 * e["SQ_INSTS_VALU"] / (32 * e["SQ_BUSY_CYCLES"])
    if "SQ_BUSY_CYCLES" in e and "SQ_WAVE_CYCLES" in e:
        dv["resident_waves_per_simd"] = 4 * e["SQ_WAVE_CYCLES"] / (32 * e["SQ_BUSY_CYCLES"])
    if "SQ_LDS_BANK_CONFLICT" in e and e.get("SQ_LDS_IDX_ACTIVE"):
        dv["lds_bank_conflict_fraction"] = e["SQ_LDS_BANK_CONFLICT"] / e["SQ_LDS_IDX_ACTIVE"]
    if "SQ_WAIT_ANY" in e and e.get("SQ_WAVE_CYCLES"):
        dv["wave_cycles_waiting"] = e["SQ_WAIT_ANY"] / e["SQ_WAVE_CYCLES"]
    e["derived"] = dv
json.dump(out, open(out_path, "w"), indent=1, sort_keys=True)
print(json.dumps({k: v["derived"] for k, v in out.items()}, indent=1))
