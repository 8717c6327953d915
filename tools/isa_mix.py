#!/usr/bin/env python
"""Static instruction mix per basic block of one kernel in a hipcc -S listing.
usage: isa_mix.py file.s <substring of the mangled kernel name> [min instructions per block to print]"""
import re, sys, collections
path, key = sys.argv[1], sys.argv[2]
minn = int(sys.argv[3]) if len(sys.argv) > 3 else 25
lines = open(path).read().split("\n")
start = [i for i, l in enumerate(lines) if key in l and re.match(r"^_Z\S+:", l)][0]
end = [i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end")][0]
blk, cnt = "entry", collections.OrderedDict()
for l in lines[start + 1:end]:
    t = l.strip()
    m = re.match(r"^(\.LBB\S+):", t)
    if m:
        blk = m.group(1)
        continue
    if not t or t[0] in ";." :
        continue
    op = t.split()[0]
    c = cnt.setdefault(blk, collections.Counter())
    if op.startswith("v_"):
        c["v_f64" if "f64" in op else ("v_dpp" if "dpp" in t else ("v_mov/cndmask" if op.startswith(("v_mov", "v_cndmask", "v_accvgpr")) else "v_int/other"))] += 1
    elif op.startswith("s_"):
        c["s_wait/branch/barrier" if op.startswith(("s_waitcnt", "s_cbranch", "s_branch", "s_barrier", "s_nop")) else "s_alu"] += 1
    elif op.startswith("ds_"):
        c["ds"] += 1
    elif op.startswith(("global_", "buffer_", "scratch_", "flat_")):
        c["vmem"] += 1
    else:
        c["other"] += 1
tot = collections.Counter()
for b, c in cnt.items():
    tot.update(c)
    if sum(c.values()) >= minn:
        print(b.ljust(12), str(sum(c.values())).rjust(5), dict(c))
print("total", sum(tot.values()), dict(tot))
