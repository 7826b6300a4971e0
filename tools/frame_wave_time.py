#!/usr/bin/env python3
"""Where a walk wave of the frame kernel spends its time (a counted frame: shader-clock stamps summed over the walk waves; flx_get_tail_diag 27..33):
the fold / refill / set-up block against the trips.  dragon 1080p 8 spp 4 bounces unless argv: width height spp bounces."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "web-ray-tracer_amd"))
from flexlight_hip import capi
from flexlight_hip.scene_io import Scene
sc = Scene.golden("dragon")
a = [int(x) for x in sys.argv[1:5]] + [1920, 1080, 8, 4][len(sys.argv[1:5]):]
ctx = capi.Context(0)
ctx.update_scene(sc)
p = sc.frame_params(width=a[0], height=a[1], samples=a[2], max_reflections=a[3], use_filter=0)
if os.environ.get("FLX_TILES"):
    p.tile_rows, p.tile_count, p.tile_index = 8, int(os.environ["FLX_TILES"]), 0
for _ in range(2):
    _, cnt, _ = ctx.render(p, counters=True)
t = ctx.get_tail_diag()
waves, life = t[26], t[24]
blk, fold, refill, setup, trips, nb, no = t[27:34]
print("walk waves %d, lifetime %.2f M cycles each" % (waves, life / waves / 1e6))
print("  trips            %5.1f %% of the lifetime (%d outer iterations per wave, %.0f cycles per 8 trips)" % (100.0 * trips / life, no / waves, trips / no))
print("  fold/refill/setup %5.1f %%   (%d blocks per wave = every %.2f outer iterations, %.0f cycles per block: fold %.0f, refill %.0f, set-up %.0f)" % (
    100.0 * blk / life, nb / waves, no / nb, blk / nb, fold / nb, refill / nb, setup / nb))
print("      of the fold's %.0f: popping the next paths' ids and issuing their loads %.0f, the fold proper (its loads' wait included) %.0f, the hand-over to the shade ring %.0f" % (fold / nb, t[34] / nb, t[35] / nb, (fold - t[34] - t[35]) / nb))
print("  the rest          %5.1f %%   (prologue tiles, waiting with nothing to walk)" % (100.0 * (life - trips - blk) / life))
