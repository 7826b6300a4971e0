#!/usr/bin/env python3
"""One workload of bench.py, a few frames, one frame per launch, through libflexlight_hip.so — and nothing else: the program
bench.py (and profiles/README.md's commands) put under `rocprofv3 --pmc ... --` to read hardware counters of the frame's
kernels.  No torch (its import is most of a short run's time); the frames stay in device memory (flx_frame_begin with FLX_FRAME_DEVICE on one
lane) after --warmup frames that do not count for steady-state figures.

    rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d out -- python3 tools/pmc_pass.py --workload dragon --frames 3
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "web-ray-tracer_amd"))


def main():
    from bench import WORKLOADS
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="dragon", choices=sorted(WORKLOADS))
    ap.add_argument("--frames", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=2, help="frames rendered before those (the first launches pay code upload and cold caches)")
    ap.add_argument("--host-output", action="store_true", help="copy every frame to pageable host memory (flx_render) instead of leaving it in device memory")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--batch", type=int, default=1, help="frames per launch (flx_render_batch) instead of one")
    ap.add_argument("--scheduler", type=int, default=0, help="flx_set_walk_scheduler: 0 lanes, 1 queues, 2 lanes + cooperative finisher")
    ap.add_argument("--suspend", type=int, default=0, help="walks a walk workgroup may hand over (flx_set_walk_scheduler)")
    ap.add_argument("--tile-rows", type=int, default=0, help="with --tile-count N: render rank --tile-index's strips of the frame (what one of N ranks traces)")
    ap.add_argument("--tile-index", type=int, default=0)
    ap.add_argument("--tile-count", type=int, default=0)
    args = ap.parse_args()
    from flexlight_hip import capi
    from flexlight_hip.scene_io import Scene
    w = WORKLOADS[args.workload]
    if len(w) > 2 and args.width is None and args.height is None:
        args.width, args.height = w[2:]
    scene = Scene.golden(w[0])
    tile = (args.tile_rows, args.tile_index, args.tile_count) if args.tile_count > 1 else (0, 0, 0)
    p = scene.frame_params(width=args.width, height=args.height, tile=tile)
    with capi.Context(0) as ctx:
        ctx.update_scene(scene)
        ctx.set_walk_scheduler(args.scheduler, args.suspend)
        ctx.set_frame_lanes(1)                 # one frame after the other, as bench.py's timed steps
        for _ in range(args.warmup + args.frames):
            if args.batch > 1:
                ctx.render_batch([p] * args.batch)
            elif args.host_output:
                ctx.render(p)
            else:                              # the frame stays in device memory: nothing but the frame's kernels between two launches
                ctx.frame_begin(p, device=True)
                ctx.frame_end()
        print("pmc_pass: %s %dx%d, %d launches of %d frame(s), pipeline %d" % (args.workload, p.width, p.height, args.frames, args.batch, ctx.last_pipeline()))


if __name__ == "__main__":
    main()
