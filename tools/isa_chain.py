#!/usr/bin/env python3
"""Print the memory-instruction / wait skeleton of one kernel from a hipcc --save-temps .s file
(how many dependent memory round trips sit on the critical path of a latency-bound launch).
usage: tools/isa_chain.py file.s <mangled-name substring> [max lines]"""
import re
import sys

txt = open(sys.argv[1]).read()
sub = sys.argv[2]
maxn = int(sys.argv[3]) if len(sys.argv) > 3 else 150
m = re.search(r'^(_ZN4dpgo\S*%s\S*):' % re.escape(sub), txt, re.M)
if not m:
    raise SystemExit("kernel not found")
start = m.end()
end = txt.find('s_endpgm', start)
keep = ('global_load', 's_load', 's_waitcnt', 'global_store', 's_barrier', 'ds_read', 'ds_write', 'ds_bpermute',
        's_cbranch', 's_branch', 'buffer_', 'scratch_')
out = []
for ln in txt[start:end].split('\n'):
    l = ln.strip()
    if not l or l[0] in ';.':
        if re.match(r'\.LBB', l):
            out.append(l.split(';')[0].strip())
        continue
    if l.split()[0].startswith(keep):
        out.append(l.split(';')[0].strip()[:64])
res, prev, cnt, last = [], None, 0, None
for o in out:
    key = o.split()[0]
    if key == prev and not key.startswith(('s_waitcnt', '.LBB', 's_cbranch', 's_branch')):
        cnt += 1
    else:
        if prev:
            res.append('%s x%d' % (prev, cnt) if cnt > 1 else last)
        prev, cnt = key, 1
    last = o
res.append('%s x%d' % (prev, cnt) if cnt > 1 else last)
print(m.group(1)[:60], '...', len(out), 'memory/wait/branch instructions')
print('\n'.join(res[:maxn]))
