"""Condense one tools/profile_round.sh output directory into the files committed under profiles/:
<tag>_<workload>_kernel_stats.csv (rocprofv3 --stats, our kernels), <tag>_<workload>_pmc.json
(per-kernel counter averages per dispatch) and the workload's entry of profiles/hbm_traffic.json
(FETCH_SIZE x 2 + WRITE_SIZE per launch: the gfx950 correction of guides/MI355X_MICROARCH.md "HBM",
checked here on k_traces whose byte count is known exactly)."""
import collections
import csv
import glob
import json
import os
import sys

out, tag, w = sys.argv[1], sys.argv[2], sys.argv[3]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(root, "profiles")
if not os.path.isdir(prof):
    prof = os.path.join(out, "profiles")
os.makedirs(prof, exist_ok=True)


def short(name):
    return name.split("<")[0].replace("void tpsrhs::", "")


def newest(pattern):
    """gpurun merges every call's files into the same directory: keep the newest run of each pass"""
    files = glob.glob(pattern, recursive=True)
    return [max(files, key=os.path.getmtime)] if files else []


stats = newest(out + "/trace/**/*kernel_stats.csv")
rows = []
if stats:
    rows = [r for r in csv.DictReader(open(stats[0]))]
    with open(os.path.join(prof, f"{tag}_{w}_kernel_stats.csv"), "w") as f:
        wr = csv.writer(f)
        wr.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows:
            if "tpsrhs::" in r["Name"]:
                wr.writerow([r[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev")])
pmc = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
for f in [x for d in glob.glob(out + "/pmc_*/") for x in newest(d + "**/*counter_collection.csv")]:
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if not k.startswith("k_"):
            continue
        pmc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta[k] = {"vgpr": int(r["VGPR_Count"]), "agpr": int(r["Accum_VGPR_Count"]), "lds": int(r["LDS_Block_Size"]),
                   "scratch": int(r.get("Scratch_Size", 0) or 0), "grid": int(r["Grid_Size"])}
summary = {}
for k in pmc:
    summary[k] = {c: sum(v) / len(v) for c, v in pmc[k].items()}
    summary[k].update(meta[k])
    if "SQ_BUSY_CYCLES" in summary[k] and summary[k]["SQ_BUSY_CYCLES"] > 0:
        b = summary[k]["SQ_BUSY_CYCLES"]
        # SQ_ACTIVE_INST_* count cycles summed over the SIMDs of the shader engines; normalise by busy cycles
        for c in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_ANY"):
            if c in summary[k]:
                summary[k][c + "_per_busy_cycle"] = summary[k][c] / b
bench = {}
for name in ("bench_plain.json", "bench_traced.json"):
    p = os.path.join(out, name)
    if os.path.exists(p):
        try:
            bench[name] = json.loads(open(p).read().strip().splitlines()[-1])
        except Exception:
            pass
json.dump({"workload": w, "counters_per_dispatch": summary, "bench": bench}, open(os.path.join(prof, f"{tag}_{w}_pmc.json"), "w"), indent=1)
tp = os.path.join(prof, "hbm_traffic.json")
traffic = json.load(open(tp)) if os.path.exists(tp) else {}
nodes = None
if "bench_plain.json" in bench:
    nodes = bench["bench_plain.json"]["config"]["nodes_per_gpu"]
ent = {"nodes": nodes, "bytes_per_launch": {}, "raw": {}, "unit_note": "FETCH_SIZE and WRITE_SIZE are reported in KiB... see tools/profile_summary.py"}
for k, v in summary.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        # rocprofv3 reports both in kilobytes (1024 B)
        ent["raw"][k] = {"FETCH_SIZE_KB": v["FETCH_SIZE"], "WRITE_SIZE_KB": v["WRITE_SIZE"]}
        ent["bytes_per_launch"][k] = 1024.0 * (2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"])
ent["valu_insts_per_launch"] = {k: v["SQ_INSTS_VALU"] for k, v in summary.items() if "SQ_INSTS_VALU" in v}
# the build the counters were taken from: bench.py reports them only for the same library (lib_sha16 of its line)
ent["lib_sha16"] = bench.get("bench_plain.json", {}).get("lib_sha16")
traffic[w] = ent
json.dump(traffic, open(tp, "w"), indent=1)
for k, v in summary.items():
    print(k, {c: round(x, 3) if isinstance(x, float) else x for c, x in v.items()})
