#!/usr/bin/env python3
"""Where should the front of a thin frame run?  A rank's 1/N of the dragon 1080p frame rendered synchronously (flx_render) with the front of the frame in its own kernel
(flx_set_frame_front 3: k_wf_front, then the frame kernel) and inside the frame kernel (2); frame time (flx_last_frame_ms), min / median of 15."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "web-ray-tracer_amd"))
from flexlight_hip import capi
from flexlight_hip.scene_io import Scene
sc = Scene.golden(os.environ.get("FLX_SCENE", "dragon"))
ctx = capi.Context(0); ctx.update_scene(sc); ctx.set_frame_chain(0)
for tiles in (1, 2, 4, 8, 16, 32):
    for idx in sorted({0, tiles // 2}):
        row = []
        for mode in (1, 3, 2):
            ctx.set_frame_front(mode)
            p = sc.frame_params(width=1920, height=1080, samples=8, max_reflections=4, use_filter=0)
            if tiles > 1: p.tile_rows, p.tile_count, p.tile_index = 8, tiles, idx
            for _ in range(3): ctx.render(p)
            ms = []
            for _ in range(15):
                ctx.render(p); ms.append(ctx.last_frame_ms()[0])
            row.append("%s %.3f / %.3f" % ({1: "automatic", 3: "own kernel", 2: "inside"}[mode], min(ms), float(np.median(ms))))
        print("1/%-2d of the frame (share %2d): %s ms" % (tiles, idx, "   ".join(row)), flush=True)
