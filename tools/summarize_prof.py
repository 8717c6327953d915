#!/usr/bin/env python3
"""Post-process rocprofv3 output of a bench.py run into the summaries committed under profiles/.

  tools/summarize_prof.py stats  <dir with *_kernel_stats.csv>  <out.csv>
  tools/summarize_prof.py pmc    <out.json>  COUNTER=<dir with *_counter_collection.csv> [COUNTER=<dir> ...]

`pmc` writes {COUNTER_KB or COUNTER: {kernel: {calls, mean, max}}}; FETCH_SIZE / WRITE_SIZE are reported by
rocprofv3 in KB and are stored under FETCH_SIZE_KB / WRITE_SIZE_KB unchanged (bench.py applies the gfx950
x2 correction to FETCH_SIZE).
"""
import csv
import glob
import json
import os
import re
import shutil
import sys


def short(name: str) -> str:
    name = re.sub(r"^void\s+", "", name.strip())
    name = name.replace("dpgo::", "")
    depth, out = 0, []
    for ch in name:  # cut the argument list: first '(' outside template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    return "".join(out).strip()


def find(d, pattern):
    hits = glob.glob(os.path.join(d, "**", pattern), recursive=True)
    if not hits:
        raise SystemExit("no %s under %s" % (pattern, d))
    return sorted(hits)[-1]


def main():
    mode = sys.argv[1]
    if mode == "stats":
        shutil.copy(find(sys.argv[2], "*kernel_stats.csv"), sys.argv[3])
        return
    out = {}
    for spec in sys.argv[3:]:
        counter, d = spec.split("=", 1)
        per = {}
        with open(find(d, "*counter_collection.csv")) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                per.setdefault(short(row["Kernel_Name"]), []).append(float(row["Counter_Value"]))
        key = counter + "_KB" if counter in ("FETCH_SIZE", "WRITE_SIZE") else counter
        out[key] = {k: dict(calls=len(v), mean=sum(v) / len(v), max=max(v)) for k, v in per.items()}
    json.dump(out, open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
