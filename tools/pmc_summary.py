"""Summarise rocprofv3 --pmc counter_collection.csv files: per-kernel averages per dispatch."""
import collections
import csv
import glob
import sys

for f in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for r in rows:
        k = r["Kernel_Name"].split("<")[0].replace("void tpsrhs::", "")
        if not k.startswith("k_"):
            continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta[k] = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["LDS_Block_Size"], r["Grid_Size"])
    for k in agg:
        print(k, "vgpr/agpr/lds/grid", meta[k])
        for c, v in sorted(agg[k].items()):
            print(f"   {c:28s} {sum(v)/len(v):16.0f}")
