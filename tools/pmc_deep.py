#!/usr/bin/env python3
"""Summarize rocprofv3 --pmc passes (one directory per pass) per kernel: mean counter value per dispatch.
usage: tools/pmc_deep.py <out.json> <kernel substring> [<kernel substring> ...] -- <dir> [<dir> ...]"""
import csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarize_prof import short, find
args = sys.argv[2:]
k = args.index("--")
keys, dirs = args[:k], args[k + 1:]
out = {}
for d in dirs:
    hits = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not hits:
        print("(no counter_collection.csv under %s: pass skipped)" % d)
        continue
    with open(sorted(hits)[-1]) as fh:
        for row in csv.DictReader(fh):
            name = short(row["Kernel_Name"])
            if not any(q in name for q in keys):
                continue
            out.setdefault(name, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
res = {kn: {c: dict(calls=len(v), mean=sum(v) / len(v), max=max(v)) for c, v in cs.items()} for kn, cs in out.items()}
json.dump(res, open(sys.argv[1], "w"), indent=1)
for kn, cs in res.items():
    print(kn)
    for c, v in sorted(cs.items()):
        print("   %-44s mean %16.1f  max %16.1f  (%d dispatches)" % (c, v["mean"], v["max"], v["calls"]))
