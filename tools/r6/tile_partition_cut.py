"""VERDICT r5 item 9, decided offline: how many adjacent pairs of workgroup tiles (64 consecutive poses) of the 100k grid
end up in different XCD shares -- for the shares the kernels use (contiguous eighths of the tile index: tile_iter) and for a
tile-graph bisection (three levels of breadth-first level sets from a pseudo-peripheral tile, halves by visit order).
Result (committed in profiles/r06_ab_results.txt): 559 against 603 of 5 453 pairs -- the index eighths are z-slabs of the
lattice and a tile is 1.3 lattice rows long, so the bisection has nothing to gain; not built.
usage: python tools/r6/tile_partition_cut.py   (CPU only)"""
import collections, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dpgo_amd import synthetic

meas, n, _ = synthetic.synthetic_grid(50, 50, 40, seed=0)
P = 64
T = (n + P - 1) // P
a = np.concatenate([meas.p1, meas.p2]) // P
b = np.concatenate([meas.p2, meas.p1]) // P
m = a != b
pairs = set(zip(a[m].tolist(), b[m].tolist()))
adj = collections.defaultdict(list)
for u, v in pairs:
    adj[u].append(v)
for u in adj:
    adj[u].sort()
und = {(min(u, v), max(u, v)) for u, v in pairs}


def cut(part):
    return sum(1 for u, v in und if part[u] != part[v])


def bfs_order(start, inset):
    seen, order, h = {start}, [start], 0
    while True:
        while h < len(order):
            u = order[h]
            h += 1
            for v in adj[u]:
                if v in inset and v not in seen:
                    seen.add(v)
                    order.append(v)
        rest = sorted(inset - seen)
        if not rest:
            return order
        seen.add(rest[0])
        order.append(rest[0])


def bisect(nodes, depth, label, out):
    if depth == 0:
        out[list(nodes)] = label
        return
    s = set(nodes)
    o = bfs_order(min(nodes), s)
    o = bfs_order(o[-1], s)  # (from a pseudo-peripheral tile)
    half = len(o) // 2
    bisect(o[:half], depth - 1, 2 * label, out)
    bisect(o[half:], depth - 1, 2 * label + 1, out)


eighths = np.zeros(T, dtype=int)
for x in range(8):
    eighths[(T * x) >> 3:(T * (x + 1)) >> 3] = x
parts = np.zeros(T, dtype=int)
bisect(list(range(T)), 3, 0, parts)
print("tiles %d, adjacent tile pairs %d" % (T, len(und)))
print("contiguous eighths of the tile index: %d pairs across shares" % cut(eighths))
print("tile-graph bisection (3 levels)      : %d pairs across shares, share sizes %s" % (cut(parts), np.bincount(parts).tolist()))
