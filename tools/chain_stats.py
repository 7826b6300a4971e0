#!/usr/bin/env python3
"""What the chained kernels of a frame loop did (flx_set_chain_stats): per launch its duration, when the next frame's view was seen, when its own frame was
complete, tiles made for either frame, what it handed on.  GPU box.  usage: chain_stats.py [--count 8 --index 0] [--frames 24] [--workload dragon|dragon_4k]"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "web-ray-tracer_amd"))
from flexlight_hip import capi
from flexlight_hip.scene_io import Scene
ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="dragon")
ap.add_argument("--count", type=int, default=8)
ap.add_argument("--index", type=int, default=0)
ap.add_argument("--frames", type=int, default=24)
ap.add_argument("--lanes", type=int, default=2)
a = ap.parse_args()
sc = Scene.golden("dragon")
size = dict(width=3840, height=2160) if a.workload == "dragon_4k" else {}
ctx = capi.Context(0)
ctx.update_scene(sc)
p = sc.frame_params(use_filter=0, **size)
if a.count > 1:
    p.tile_rows, p.tile_count, p.tile_index = 8, a.count, a.index
ctx.set_frame_lanes(a.lanes)
ctx.set_frame_chain(1)
ctx.set_chain_stats(1)
for _ in range(a.lanes - 1):
    ctx.frame_begin(p, device=True)
t0 = time.perf_counter()
for _ in range(a.frames):
    ctx.frame_begin(p, device=True); ctx.frame_end()
dt = time.perf_counter() - t0
while ctx.frames_in_flight():
    ctx.frame_end()
st = ctx.chain_stats()
rows = sorted([r for r in st if r[21] != 0], key=lambda r: int(r[21]))
print("%d frames, %.3f ms per frame (wall clock); times in us from the launch's start" % (a.frames, dt * 1e3 / a.frames))
print("second table: S's tile queue dry (first / last workgroup); shade waves: share of their time making tiles / shading batches; walk waves: mean walking lanes per trip, trips per wave")
print("%4s %8s %8s %8s %8s %8s %8s %8s | %6s %6s | %7s %7s %7s | %6s %7s %6s %7s | %5s %5s %7s %7s | %4s %4s | gap to next" %
      ("seq", "dur", "end_min", "Sseen", "Sseen_mx", "Pdone_mn", "Pdone_mx", "stop", "tilesP", "tilesS", "pullWlk", "pullShd", "pullRdy", "susp", "restart", "chunk", "dumped", "bP", "bS", "lanesP", "lanesS", "fin", "stop"))
NONE = np.uint64(0xffffffffffffffff)
def us(v, s): return "%8.1f" % ((int(v) - int(s)) / 100.0) if v != NONE and v != 0 else "%8s" % "-"
for i, r in enumerate(rows):
    s = r[0]
    gap = "%7.1f" % ((int(rows[i + 1][0]) - int(r[1])) / 100.0) if i + 1 < len(rows) else ""
    print("%4d %s %s %s %s %s %s %s | %6d %6d | %7d %7d %7d | %6d %7d %6d %7d | %5d %5d %7d %7d | %4d %4d | %s" %
          (r[21], us(r[1], s), us(r[20], s), us(r[2], s), us(r[22], s), us(r[4], s), us(r[5], s), us(r[3], s), r[6], r[7], r[11], r[12], r[13], r[23], r[8], r[9], r[10], r[14], r[15], r[16], r[17], r[18], r[19], gap))

for r in rows:
    s = r[0]
    tot = max(int(r[29]), 1)
    print("%4d  Sdry %s %s | shade tile %4.1f%% batch %4.1f%% of their time (%.0f us per wave) | walk: %.1f lanes per trip, %.0f trips per wave" %
          (r[21], us(r[25], s), us(r[26], s), 100.0 * int(r[27]) / tot, 100.0 * int(r[28]) / tot, tot / 100.0 / 768.0, int(r[30]) / max(int(r[31]), 1), int(r[31]) / (13 * 256.0)))

print("the frame two ahead (depth 3): view seen at, tiles made, batches, lanes")
for r in rows:
    print("%4d  %s  tiles %5d  batches %6d  lanes %8d" % (r[21], us(r[47], r[0]), r[44], r[45], r[46]))
print("workgroups through with their own frame by (us): <100 <200 <400 <600 <800 <1000 <1200 <1400 <1600 <1800 <2000 later")
for r in rows:
    print("%4d  %s" % (r[21], " ".join("%4d" % int(v) for v in r[32:44])))

print("P's own resume lists taken: walk shade ready(units) susp | folds of P's paths by bounce 0 1 2 3+ (after 500 us) | last fresh path of P began its walk at")
for r in rows:
    print("%4d  %7d %7d %7d %7d | %7d %7d %7d %7d  (%6d %6d %6d %6d) | %s" % (r[21], r[48], r[49], r[50], r[51], r[52], r[53], r[54], r[55], r[57], r[58], r[59], r[60], us(r[56], r[0])))
