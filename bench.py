#!/usr/bin/env python3
"""bench.py — Mray/s of the FlexLight path-tracing hot path on MI355X (BASELINE.json metric).

A "step" is one frame of the workload through libflexlight_hip.so's C ABI with the scene resident in
HBM: path-trace pass (+ denoise chain when the workload has filter on) and, for N > 1, the RCCL
all-gather of the row-strip tiles plus the reassembly of the frame on every rank.  The K frames of the
timed region are rendered in batches of up to --batch frames per pass of the pipeline
(flx_render_batch_device: every frame complete and bit-identical to its own render) —
filter frames one by one — and the frame-after-frame rate is reported beside it ("frame_after_frame").

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch F] [--workload dragon|dragon_4k|cornell_obj|cornell|theater]

N > 1 is launched by torch.distributed.run, one rank per GPU (RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_* from the environment).  The frame is cut into strips of --tile-rows image rows dealt
round-robin to the ranks: total work is fixed as N grows -> "scaling": "strong".

Rank 0 prints ONE JSON line: metric/value as BASELINE.json defines them (nominal path segments
spp x bounces x W x H per second), a "roofline" object for the dominant kernel (algorithmic bytes per
launch from the frame's work counters / its HIP-event duration, against the 8 TB/s HBM peak) and a
"cpu_baseline" object (the CPU oracle timed on this box's host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "web-ray-tracer_amd"))

HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s
FP32_VECTOR_PEAK_TFLOPS = 157.3     # MI355X FP32 vector (non-matrix) peak, MI355X_MICROARCH.md

WORKLOADS = {
    # name: (scene fixture, BASELINE.json config it is)
    "dragon": ("dragon", "configs[2]: dragon.obj (dragon_lp.obj, 73 694 entries) 1080p, 8 spp, 4 bounces"),
    "cornell_obj": ("cornell_obj", "configs[1]: cornell.obj 1080p, 4 spp, 3 bounces, filter on"),
    "cornell": ("cornell", "configs[0]: examples/cornell.js 256x256, 1 spp, 1 bounce, filter off"),
    "theater": ("theater", "configs[4]: examples/theater.js 1080p, 16 spp, 6 bounces"),
    "dragon_4k": ("dragon", "configs[3]: dragon.obj (dragon_lp.obj) 3840x2160, 8 spp, 4 bounces", 3840, 2160),
}


def algorithmic_bytes(cnt, n_lights, pixels, use_filter):
    """B_frame = 48 N_visit + 160 N_shade + 24 L N_shade + 4 N_tex + B_out W H (+ B_filter)."""
    visits = cnt["primary_visits"] + cnt["closest_visits"] + cnt["shadow_visits"]
    b = 48 * visits + 160 * cnt["shades"] + 24 * n_lights * cnt["shades"] + 4 * cnt["atlas_texels"]
    b += (20 if use_filter else 16) * pixels
    if use_filter:
        b += 224 * pixels
    return b


def cpu_baseline(scene, params_full, seconds_budget=12.0):
    """CPU oracle (kind 'port': this repo's C restatement of the reference GLSL — the reference has no
    CPU path) on a bounded sample of the same workload: the same scene / spp / bounces, the full frame
    when one frame fits the budget (repeated until ~seconds_budget of CPU work), otherwise a centred
    sub-resolution frame sized from a quick probe."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import flx_oracle
    threads = min(16, os.cpu_count() or 1)
    spp, bounces = params_full.samples, params_full.max_reflections
    W, H = params_full.width, params_full.height
    probe = scene.frame_params(width=240, height=135, samples=spp, max_reflections=bounces, use_filter=0)
    t0 = time.time()
    flx_oracle.render(scene, probe, threads=threads)
    dt = max(time.time() - t0, 1e-3)
    est_full = dt * (W * H) / (240 * 135)
    if est_full <= seconds_budget:
        w, h = W, H
        frames = max(1, int(seconds_budget / max(est_full, 1e-3)))
    else:
        scale = (seconds_budget / dt) ** 0.5
        w, h = max(8, int(240 * scale) // 8 * 8), max(8, int(135 * scale) // 8 * 8)
        frames = 1
    p = scene.frame_params(width=w, height=h, samples=spp, max_reflections=bounces, use_filter=0)
    t0 = time.time()
    for _ in range(frames):
        flx_oracle.render(scene, p, threads=threads)
    dt = time.time() - t0
    # one core beside it: every 16th strip of the same frame, ~1/16 of its rays, about 3 s
    p1 = scene.frame_params(width=w, height=h, samples=spp, max_reflections=bounces, use_filter=0, tile=(8, 0, 16))
    rows1 = len(flx_oracle.tile_rows(p1))
    t1 = time.time()
    flx_oracle.render(scene, p1, threads=1)
    dt1 = max(time.time() - t1, 1e-6)
    return {
        "value": frames * spp * bounces * w * h / dt / 1e6, "unit": "Mray/s", "cores": threads, "kind": "port",
        "sample": "same scene/spp/bounces, %dx%d frame x %d (%.1f s of CPU oracle, %d OpenMP threads, filter off)" % (w, h, frames, dt, threads),
        "single_thread": {"value": spp * bounces * w * rows1 / dt1 / 1e6, "unit": "Mray/s", "cores": 1,
                          "sample": "every 16th 8-row strip of that frame (%d rows, %.1f s)" % (rows1, dt1)},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--workload", default="dragon", choices=sorted(WORKLOADS))
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--tile-rows", type=int, default=8)
    ap.add_argument("--batch", type=int, default=32, help="most frames per pass of the pipeline (flx_render_batch_device; a pass also holds at most 2^28 paths: 16 whole 1080p frames of 8 samples); 1 = frame after frame; filter frames are never batched")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-frame-after-frame", action="store_true", help="skip the one-frame-per-pass measurement after the timed region (profiling runs: every launch of a kernel is then a whole batch)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse the rank logic)")
    ap.add_argument("--one-device", action="store_true", help="rehearsal: every rank uses GPU 0 (needs --backend gloo)")
    ap.add_argument("--verify", action="store_true", help="after the run, rank 0 renders the whole frame on its own and compares the gathered frame with it (bit for bit)")
    args = ap.parse_args()

    import numpy as np
    import torch
    from flexlight_hip import capi, tiles
    from flexlight_hip.scene_io import Scene

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with %d ranks" % (args.gpus, args.gpus))
        args.gpus = world
    if args.one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    force_dist = os.environ.get("FLX_BENCH_FORCE_DIST") == "1"      # rehearse the RCCL code path with a single rank
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    fixture, config_name = WORKLOADS[args.workload][:2]
    if len(WORKLOADS[args.workload]) > 2 and args.width is None and args.height is None:
        args.width, args.height = WORKLOADS[args.workload][2:]
    scene = Scene.golden(fixture)
    full = scene.frame_params(width=args.width, height=args.height)
    use_filter = int(full.use_filter)
    W, H = full.width, full.height
    multi = world > 1 or force_dist
    tile = (args.tile_rows, rank, world) if multi else (0, 0, 0)
    params = scene.frame_params(width=args.width, height=args.height, tile=tile)
    filter_multi = bool(use_filter) and multi          # strips -> RGBA8 render targets -> gather -> whole-frame chain

    ctx = capi.Context(local_rank)
    ctx.update_scene(scene)
    stream = torch.cuda.Stream()              # a real (non-null) stream shared by the kernels and the RCCL gather:
    torch.cuda.set_stream(stream)             # no host sync between trace kernel, all-gather and reassembly
    ctx.set_stream(stream.cuda_stream)

    # Frames per pass: a walk kernel lasts as long as its longest walk whatever the number of paths, so the
    # bench renders the K frames of the timed region in batches of F (the last one may be smaller): throughput mode.
    rows_local = ctx.tile_row_count(params)
    rows_max = tiles.padded_rows(H, args.tile_rows, world) if multi else H
    F = 1 if use_filter else max(1, min(args.batch, capi.MAX_BATCH_FRAMES))
    F = max(1, min(F, (1 << 28) // max(1, rows_max * W * full.samples)))        # at most 2^28 paths (34 GB of path records) per pass; the same F on every rank
    local = torch.zeros((F, rows_max, W, 4), dtype=torch.float32, device="cuda")
    gathered = torch.empty((world, F, rows_max, W, 4), dtype=torch.float32, device="cuda") if multi else None
    frame = torch.empty((F, H, W, 4), dtype=torch.float32, device="cuda") if multi else None      # the frames of the last batch, every rank has them
    perms = {}
    if multi:
        rows_of = []                      # image rows of every rank's packed rows
        for r in range(world):
            pr = scene.frame_params(width=args.width, height=args.height, tile=(args.tile_rows, r, world))
            rows_of.append(list(capi.Context.tile_rows(pr)))

        def perm_for(f):
            """one gather instead of select + scatter: flexlight_hip/tiles.py gather_index, cached per batch size"""
            if f not in perms:
                perms[f] = torch.from_numpy(tiles.gather_index(rows_of, f, rows_max, H)).cuda()
            return perms[f]
        perm = perm_for(1)

    if filter_multi:
        planes_local = torch.zeros((5, rows_max, W), dtype=torch.int32, device="cuda")
        planes_all = torch.empty((world, 5, rows_max, W), dtype=torch.int32, device="cuda")
        planes = torch.empty((5, H, W), dtype=torch.int32, device="cuda")
        if rows_local != rows_max:            # the C ABI packs [5][rows_local][W]; ranks with a strip less use a view of that shape
            planes_tight = torch.zeros((5, rows_local, W), dtype=torch.int32, device="cuda")

    # N > 1, batches: the gather of a batch (all-gather + reassembly, on their own stream) runs while the next batch is traced;
    # two sets of buffers, events in both directions (a slot's strips are not overwritten before their gather has read them)
    comm_stream = torch.cuda.Stream() if multi else None
    if multi and F > 1:
        slots = [(local, gathered, frame), (torch.zeros_like(local), torch.empty_like(gathered), torch.empty_like(frame))]
        traced = [torch.cuda.Event(), torch.cuda.Event()]
        gathered_done = [torch.cuda.Event(), torch.cuda.Event()]
    state = {"slot": 0, "last": 0}

    def render_batch(f):
        """f frames of this rank's strips, then (N > 1) the gather: every rank ends up with the f whole frames"""
        if not multi:
            ctx.render_batch_device([params] * f, local.data_ptr())
            return
        k = state["slot"]
        state["slot"], state["last"] = k ^ 1, k
        loc, gat, frm = slots[k]
        stream.wait_event(gathered_done[k])                  # (a no-op the first time round)
        ctx.render_batch_device([params] * f, loc.data_ptr())
        traced[k].record(stream)
        n = f * rows_max * W * 4
        with torch.cuda.stream(comm_stream):
            comm_stream.wait_event(traced[k])
            dist.all_gather_into_tensor(gat.view(-1)[:world * n], loc.view(-1)[:n])
            torch.index_select(gat.view(-1)[:world * n].view(world * f * rows_max, W, 4), 0, perm_for(f), out=frm.view(F * H, W, 4)[:f * H])
            gathered_done[k].record(comm_stream)

    def batch_sizes(k):
        """k frames in the fewest batches of at most F frames, sized evenly (20 frames, F = 8: 7 + 7 + 6)"""
        n = (k + F - 1) // F
        return [k // n + (1 if i < k % n else 0) for i in range(n)] if n else []

    def run_frames(k):
        """k frames, in batches of F"""
        if F == 1:
            for _ in range(k):
                step()
            return
        for f in batch_sizes(k):
            render_batch(f)

    def step():
        if filter_multi:
            if rows_local == rows_max:
                ctx.render_planes_device(params, planes_local.data_ptr())
            else:
                ctx.render_planes_device(params, planes_tight.data_ptr())
                planes_local[:, :rows_local, :] = planes_tight
            dist.all_gather_into_tensor(planes_all.view(-1), planes_local.view(-1))
            for k in range(5):
                torch.index_select(planes_all[:, k].reshape(world * rows_max, W), 0, perm, out=planes[k])
            ctx.filter_planes_device(full, planes.data_ptr(), frame.data_ptr())      # every rank ends up with the frame
            return
        ctx.render_device(params, local.data_ptr())         # filter-on frames: trace + denoise chain, all on the GPU
        if multi:
            dist.all_gather_into_tensor(gathered.view(-1), local.view(-1))
            torch.index_select(gathered.view(world * F * rows_max, W, 4), 0, perm, out=frame[0])

    def fence():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
            torch.cuda.synchronize()

    if multi and F > 1:
        for f in set(batch_sizes(args.steps) + batch_sizes(args.warmup) + [F]):
            perm_for(f)                  # index tables built before the timed region
    if F > 1:
        render_batch(F)                  # untimed: the context sizes its path records and lists for a full batch here, not in the timed region
    run_frames(args.warmup)
    fence()
    kernel_ms = []
    t0 = time.perf_counter()
    run_frames(args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # Dominant-kernel duration, measured live with HIP events on the launch stream, outside the timed
    # region so the event syncs do not perturb it: same frame, K more launches.
    def trace_share():                   # this rank's share of the trace, without gather / chain
        if filter_multi:
            ctx.render_planes_device(params, (planes_local if rows_local == rows_max else planes_tight).data_ptr())
        elif F > 1:
            ctx.render_batch_device([params] * F, local.data_ptr())
        else:
            ctx.render_device(params, local.data_ptr())

    # frame after frame (latency mode), for the record: the same share of the frame, one frame per pass
    single_ms = None
    if F > 1 and not args.no_frame_after_frame:
        for it in range(3 + 10):
            if it == 3:
                torch.cuda.synchronize(); t1 = time.perf_counter()
            ctx.render_device(params, local.data_ptr())
        torch.cuda.synchronize()
        single_ms = (time.perf_counter() - t1) / 10 * 1e3

    ctx.set_wavefront_groups(1)          # one chain, so the bounce-0 walk kernel is ONE launch over the whole share of the frame
    for _ in range(min(args.steps, 10)):
        trace_share()
        frame_ms, trace_ms = ctx.last_frame_ms()
        kernel_ms.append(trace_ms)
    # Work counters of this rank's share of the frame (a counted launch; not timed), and of bounce 0 alone
    # (the same frame cut after one bounce: identical paths) for the dominant kernel's roofline.
    ctx.set_counters_enabled(True)
    if filter_multi:
        trace_share()
    else:
        ctx.render_device(params, local.data_ptr())      # ONE frame's share: the counters below are per frame
    ctx.sync()
    cnt = ctx.get_counters()
    # entries the round-0 walk kernel visited in that frame (its own tally, diag slot 15; 0 when another kernel walked)
    b0_visits = ctx.get_diag()[15] if not use_filter else 0
    ctx.set_counters_enabled(False)
    torch.cuda.synchronize()

    verified = None
    if args.verify and multi and rank == 0:
        torch.cuda.synchronize()                             # `frame` holds the last gathered frame of the timed loop
        whole = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
        ctx.render_device(full, whole.data_ptr())
        ctx.sync()
        last = batch_sizes(args.steps)[-1] - 1 if F > 1 else 0           # position of the last frame in the last batch
        last_frames = slots[state["last"]][2] if F > 1 else frame
        verified = bool(torch.equal(whole.view(torch.int32), last_frames[last].view(torch.int32)))
    if rank == 0:
        spp, bounces = full.samples, full.max_reflections
        rays = spp * bounces * W * H
        ms_per_step = elapsed / args.steps * 1e3
        value = rays / (elapsed / args.steps) / 1e6
        n_lights = scene.arrays["lights"].size // 6
        k_ms = float(np.mean(kernel_ms))
        frame_bytes = algorithmic_bytes(cnt, n_lights, rows_local * W, use_filter)      # per frame
        pipe = ctx.last_pipeline()
        if pipe == 1:              # per-pixel kernel: the whole trace
            kernel_name, bytes_launch = "k_trace_pixels", frame_bytes - (244 * rows_local * W if use_filter else 0)
        elif pipe == 2:            # persistent path kernel (tiny scenes): every bounce of every path, no primary walk, no output
            kernel_name = "k_paths (persistent path kernel)"
            bytes_launch = 48 * (cnt["closest_visits"] + cnt["shadow_visits"]) + (160 + 24 * n_lights) * cnt["shades"] + 4 * cnt["atlas_texels"]
        else:                      # the bounce-0 walk kernel's share of B_frame: the 48-byte entries its walks visit
            visits = b0_visits if b0_visits else cnt["closest_visits"] + cnt["shadow_visits"]      # (a one-bounce frame, or another walk kernel: all of them)
            kernel_name, bytes_launch = "k_wf_walk_pre<false, true> (walk kernel of bounce 0)", 48 * visits
        bytes_launch *= F              # one launch walks the F frames of a batch
        achieved = bytes_launch / (k_ms * 1e-3) / 1e9
        traffic = None
        try:                       # HBM bytes of that kernel from rocprofv3 PMC passes (profiles/, not measurable from inside bench.py)
            with open(os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")) as fh:
                t = json.load(fh)
            if t["workload"] == args.workload and world == 1 and W == 1920 and H == 1080 and kernel_name == t["kernel"] and t.get("frames_per_launch", 1) == F:
                traffic = t["traffic_bytes_per_launch"]
        except (OSError, KeyError, ValueError):
            pass
        line = {
            "metric": "Mray/s at 1080p (spp x bounces x pixels / s)", "value": value, "unit": "Mray/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": config_name, "width": W, "height": H, "spp": spp, "bounces": bounces, "filter": bool(use_filter),
                "scene_entries": int(scene.meta["textureLength"]), "parallelism": "row-strip tiles x%d, %d rows/strip, RCCL all-gather%s" % (world, args.tile_rows, " of a batch overlapped with the trace of the next" if F > 1 else "") if world > 1 else "single GPU",
                "rays_per_frame": rays, "frames_per_pass": F,
                "frames": "the static camera of the BASELINE config for every frame, as in the reference's frame loop; every frame is traced in full, nothing is reused between frames",
            },
            "roofline": {
                "bound": "hbm", "kernel": kernel_name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "algorithmic_bytes_per_launch": bytes_launch, "kernel_ms": k_ms,
                "frame_algorithmic_bytes": frame_bytes, "frame_achieved": frame_bytes / (ms_per_step * 1e-3) / 1e9,
                "note": "algorithmic bytes = 48 B x entries visited + 160 B x shades + 24 B x lights x shades + 4 B x texels + 16 B x pixels (SURVEY.md 8d); the <=12 MB scene is cache resident, real HBM traffic is far lower",
                # SURVEY.md 8d's secondary figure: ~30 flop per entry visited against the FP32 vector peak (the binding limits are
                # per-lane latency and VALU issue under divergence)
                "secondary": {"bound": "valu_fp32", "achieved": 30.0 * bytes_launch / 48.0 / (k_ms * 1e-3) / 1e12, "peak": FP32_VECTOR_PEAK_TFLOPS,
                              "unit": "TFLOP/s", "frac": 30.0 * bytes_launch / 48.0 / (k_ms * 1e-3) / 1e12 / FP32_VECTOR_PEAK_TFLOPS},
            },
            "counters": cnt,
        }
        if single_ms is not None:
            line["frame_after_frame"] = {"ms_per_frame": single_ms, "value": rays / (single_ms * 1e-3) / 1e6, "unit": "Mray/s",
                                         "note": "same share of the frame rendered one frame per pass (flx_render_device), without the gather; `value` above renders %d frames per pass (flx_render_batch_device): every frame complete, latency %d frames" % (F, F)}
        if verified is not None:
            line["gathered_frame_equals_single_context_frame"] = verified
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(scene, full)
        print(json.dumps(line), flush=True)
    ctx.close()
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
