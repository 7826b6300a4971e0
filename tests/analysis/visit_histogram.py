#!/usr/bin/env python3
"""Which entries of the skip list do the walks actually touch?  (CPU oracle, analysis only.)
Prints the share of all entry fetches covered by the N shallowest entries (depth-first by tree depth),
i.e. what an LDS-resident top of the tree would absorb."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "web-ray-tracer_amd"))
import flx_oracle
from flexlight_hip.scene_io import Scene
name = sys.argv[1] if len(sys.argv) > 1 else "dragon"
sc = Scene.golden(name)
g = sc.arrays["geometry"].reshape(-1, 12)
n = g.shape[0]
# depth of every entry from the skip counts
depth = np.zeros(n, np.int32); stack = []
for i in range(n):
    while stack and i > stack[-1]: stack.pop()
    depth[i] = len(stack)
    if g[i, 10] == 1: stack.append(i + int(g[i, 6]))
    if g[i, 10] == 0: depth[i] = 10 ** 6
hist = np.zeros(n, np.uint64)
lib = flx_oracle.lib()
lib.flx_oracle_set_visit_histogram.argtypes = [C.c_void_p]
lib.flx_oracle_set_visit_histogram(hist.ctypes.data)
p = sc.frame_params(width=480, height=270, use_filter=0)
flx_oracle.render(sc, p)
lib.flx_oracle_set_visit_histogram(None)
total = hist.sum()
order = np.lexsort((np.arange(n), depth))
cum = np.cumsum(hist[order]) / total
print("entries", int((g[:, 10] != 0).sum()), "total fetches", int(total), "max depth", int(depth[g[:, 10] != 0].max()))
for N in (64, 256, 512, 761, 1024, 1365, 2048, 2730, 3400, 8192, 16384):
    if N <= n: print("top %5d entries by depth: %.3f of fetches (depth <= %d)" % (N, cum[N - 1], depth[order[N - 1]]))
best = np.sort(hist)[::-1].cumsum() / total
for N in (761, 1024, 2730): print("best possible %d entries: %.3f" % (N, best[N - 1]))
