#!/usr/bin/env python3
"""summarise the rocprofv3 counter_collection CSVs written by profiles/prof_pmc.sh into one JSON (per kernel, per counter)"""
import collections, csv, glob, json, sys
src, out = sys.argv[1], sys.argv[2]
res = collections.defaultdict(dict)
for d in ("fetch", "write", "sq", "sq2", "tcc", "lanes"):
    for f in glob.glob(f"{src}/{d}/*counter_collection.csv"):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        meta = {}
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            meta[k] = dict(vgpr=int(r["VGPR_Count"]), sgpr=int(r["SGPR_Count"]), lds=int(r["LDS_Block_Size"]), grid=int(r["Grid_Size"]),
                           wg=int(r["Workgroup_Size"]))
            meta[k].setdefault("ms", 0.0)
        for r in csv.DictReader(open(f.replace("counter_collection", "kernel_trace"))):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if k in meta:
                meta[k]["ms"] = meta[k].get("ms", 0.0) + (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        for k in agg:
            res[k].update(agg[k])
            res[k].update({"_" + a: b for a, b in meta[k].items()})
res = {k: v for k, v in res.items() if v.get("_ms", 0) > 0.05}
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
for k, v in sorted(res.items(), key=lambda kv: -kv[1].get("_ms", 0)):
    print(k, {a: (f"{b:.3e}" if isinstance(b, float) and abs(b) > 1e4 else b) for a, b in v.items()})
