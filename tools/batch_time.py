#!/usr/bin/env python3
"""Frame time of the dragon frame rendered in batches of F frames (flx_render_batch_device), for a rank's share 1/N of the
frame.  usage: [FLX_SCENE=theater] [FLX_PIPELINE=3] [FLX_BATCHES=1,8] [FLX_SIZE=3840x2160] batch_time.py [N ...]   (GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "web-ray-tracer_amd"))
import torch
from flexlight_hip import capi
from flexlight_hip.scene_io import Scene
sc = Scene.golden(os.environ.get("FLX_SCENE", "dragon"))
ctx = capi.Context(0)
ctx.update_scene(sc)
if os.environ.get("FLX_PIPELINE"):
    ctx.set_pipeline(int(os.environ["FLX_PIPELINE"]))
for n in [int(a) for a in sys.argv[1:]] or [1, 8]:
    for F in [int(x) for x in os.environ.get("FLX_BATCHES", "1,2,4,8").split(",")]:
        w, h = ([int(v) for v in os.environ["FLX_SIZE"].split("x")] if os.environ.get("FLX_SIZE") else (None, None))
        p = sc.frame_params(width=w, height=h, use_filter=0)
        p.tile_rows, p.tile_count, p.tile_index = 8, n, 0
        rows = ctx.tile_row_count(p)
        out = torch.empty((F, rows, p.width, 4), dtype=torch.float32, device="cuda")
        ms = []
        for it in range(3 + 6):
            ctx.render_batch_device([p] * F, out.data_ptr())
            ctx.sync()
            if it >= 3:
                ms.append(ctx.last_frame_ms()[0])
        print("1/%d of the frame, batches of %d: %.3f ms per batch, %.3f ms per frame" % (n, F, sum(ms) / len(ms), sum(ms) / len(ms) / F), flush=True)
        del out
