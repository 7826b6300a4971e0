#!/usr/bin/env python3
"""bench.py — Mray/s of the FlexLight path-tracing hot path on MI355X (BASELINE.json metric).

A "step" is ONE frame of the workload through libflexlight_hip.so's C ABI, one frame per pass of the pipeline, the scene
resident in HBM: path-trace pass (+ denoise chain when the workload has the filter on) and, for N > 1, the library's own
RCCL all-gather of the row-strip tiles and its reassembly kernel (flx_render_gathered_device: no torch collective in the
timed region).  The timed region is that frame-after-frame rate (frame time = first kernel launch ..
last byte of the gathered frame; `one_frame_per_pass`).  The reference's loop never waits for the GPU between frames
(pathtracerWGL2.js:254-303), so the same K frames are also timed through the library's frame loops — `pipelined` (two frames in
flight) and, N > 1, `shared` (every rank's frame server, three in flight, no exchange) — between the same fences: named entries of
the line with their own `value`, the fastest verified one repeated as `throughput_best`.  `value` / `ms_per_step` are ALWAYS the timed
region's (one frame per pass), so that `ms_per_step` >= the dominant kernel's time per step (`roofline.kernel_ms`).  Rendering several frames per pass (flx_render_batch, a throughput mode with a latency of
F frames) is reported beside it as `batched`, with a different camera for every frame of a batch.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload dragon|dragon_100k|dragon_4k|cornell_obj|cornell|theater]

N > 1: one rank per GPU.  Under torch.distributed.run the ranks are there already (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from
the environment); started plainly — `python bench.py --gpus 8` — this process launches the N ranks itself as child processes
(before it touches a GPU: it never does) and relays rank 0's line.  torch.distributed (gloo) only hands the RCCL communicator
id to the ranks and provides the barrier.  The frame is cut into strips of --tile-rows image rows dealt round robin to the
ranks: total work is fixed as N grows -> "scaling": "strong".  The gathered frame is verified against one
context's frame (bit for bit) and the line says how it was exchanged and over how many RCCL ranks.

Rank 0 prints ONE JSON line.  `roofline` describes the dominant kernel of a frame — the frame kernel (k_wf_frame: the primary
rays, every bounce's shading and every bounce walk of the frame in one persistent launch; k_trace_pixels / k_paths for the
small scenes; `roofline.kernel` names it) — by the limit that binds it: VALU issue (wave-instructions per second against
256 CUs x 4 SIMDs x 1/2 per cycle x 2.4 GHz), from SQ_INSTS_VALU of a rocprofv3 --pmc pass THIS run makes over the same frame
(tools/pmc_pass.py, before the parent touches the GPU), with the measured HBM traffic (FETCH_SIZE / WRITE_SIZE passes) and
SURVEY.md 8d's algorithmic bytes beside it.  `cpu_baseline` is the CPU oracle timed on this box's host cores on a bounded sample.

N > 1 also reports `pipelined` (the frame loop over the communicator, two lanes) and `shared`: the same loop WITHOUT a collective — every rank's frame
server (one persistent launch, three frames in flight) resolves its strips straight into one image in rank 0's device memory (flx_share_*: hipIpc mapping,
stores over xGMI), verified against one context's frame.
`gathered_rgba8` is the frame-after-frame loop with the frames travelling as the canvas' RGBA8 (a quarter of the bytes).  These and `batched` are measured
AFTER the line is complete, under --secondary-timeout: a phase that does not come back ends the run with the line as it stands (`secondary_incomplete`).

Exit status: non-zero when a rank fails, when the ranks do not finish within --rank-timeout seconds (they are killed), or when
the gathered frame differs from the single-context frame (the line is still printed, with the field false).
"""
import argparse
import csv
import glob
import json
import os
import shutil
import signal
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "web-ray-tracer_amd"))

HBM_PEAK_GBS = 8000.0                 # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s
FP32_VECTOR_PEAK_TFLOPS = 157.3       # MI355X FP32 vector peak (same guide): 256 CUs x 4 SIMDs x 32 lanes x 2 flop x 2.4 GHz
VALU_ISSUE_PEAK = 256 * 4 * 0.5 * 2.4e9      # wave64 VALU instructions per second: one per 2 cycles per SIMD (v_fma_f32 row of the guide) = 1.2288e12

DATA = {      # what the frame is rendered from (no dataset is downloaded: the fixtures travel with the repo)
    "dragon": "the reference's own scene arrays (tests/golden/ref_dragon.flxs.gz: what its scene.js emits for objects/dragon_lp.obj + the example's scene), BASELINE camera; no weights, no dataset",
    "dragon_4k": "the reference's own scene arrays (tests/golden/ref_dragon.flxs.gz), BASELINE camera, 3840x2160",
    "dragon_100k": "synthetic: dragon_lp.obj with every triangle split 1 -> 4 (tools/make_dragon_100k.py)",
    "cornell_obj": "the reference's own scene arrays (tests/golden/ref_cornell_obj.flxs.gz)",
    "cornell": "the reference's own scene arrays (tests/golden/ref_cornell.flxs.gz)",
    "theater": "the reference's own scene arrays (tests/golden/ref_theater.flxs.gz: examples/theater.js with its three atlases)",
}

WORKLOADS = {
    # name: (scene fixture, BASELINE.json config it is[, width, height])
    "dragon": ("dragon", "configs[2]: dragon.obj (dragon_lp.obj, 73 694 entries) 1080p, 8 spp, 4 bounces"),
    "dragon_100k": ("dragon_100k", "configs[2] with a SYNTHETIC >= 100k-triangle dragon (dragon_lp.obj, every triangle split 1 -> 4 at its edge midpoints: 174 276 triangles in the dragon; objects/dragon.obj is absent from the reference mount), 1080p, 8 spp, 4 bounces"),
    "cornell_obj": ("cornell_obj", "configs[1]: cornell.obj 1080p, 4 spp, 3 bounces, filter on"),
    "cornell": ("cornell", "configs[0]: examples/cornell.js 256x256, 1 spp, 1 bounce, filter off"),
    "theater": ("theater", "configs[4]: examples/theater.js 1080p, 16 spp, 6 bounces"),
    "dragon_4k": ("dragon", "configs[3]: dragon.obj (dragon_lp.obj) 3840x2160, 8 spp, 4 bounces", 3840, 2160),
}

PMC_WARMUP_FRAMES = 2       # frames tools/pmc_pass.py renders before the ones that count (its --warmup)
KT_FRAMES = 12              # frames of the kernel-trace pass
PMC_PASSES = [      # separate rocprofv3 passes (FETCH_SIZE and WRITE_SIZE do not fit one pass: MI355X_MICROARCH.md, PMC slots)
    ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVES", "SQ_INSTS_SALU", "SQ_WAIT_INST_ANY"],
    ["FETCH_SIZE"],
    ["WRITE_SIZE"],
    # the vector instructions by class (profiles/r04_valu_rates.txt: which instruction which counter counts, and what a class costs to issue)
    ["SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT64", "SQ_INSTS_VALU_CVT"],
    ["SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64"],
]

# SIMD cycles a wave64 vector instruction of a class takes to issue with four waves per SIMD (tools/micro/valu_rates.hip on this part, profiles/r04_valu_rates.txt):
# full rate 2 (f32 add / sub / mul / fma, moves, logic, selects, add_u32), half rate 4 (min / max / compares / shifts / mbcnt / conversions, f64 arithmetic), quarter rate 8 (f32
# transcendentals; f64 ones 16).  "int32" mixes full and half rate (3).  What no class counter counts ("other": compares, min / max, selects, moves, logic, shifts) lies between 2 and 4;
# the frame kernel's stepping loop holds 70 % half-rate ones among them (tools/asm_mix.py): 3.4.
ISSUE_CYCLES = {"f32": 2.0, "trans_f32": 8.0, "int32": 3.0, "int64": 4.0, "cvt": 4.0, "f64": 4.0, "trans_f64": 16.0}
ISSUE_CYCLES_OTHER = (2.0, 3.4, 4.0)


def issue_cycles(k, kernel_ms):
    """the dominant kernel's vector instructions priced per class -> share of the launch in which the SIMDs' vector pipes issue, as (low, estimate, high) with the classes' counts"""
    need = ["SQ_INSTS_VALU", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT64", "SQ_INSTS_VALU_CVT"]
    if any(k.get(n) is None for n in need) or not kernel_ms:
        return None
    n = {"f32": k["SQ_INSTS_VALU_ADD_F32"] + k["SQ_INSTS_VALU_MUL_F32"] + k["SQ_INSTS_VALU_FMA_F32"], "trans_f32": k["SQ_INSTS_VALU_TRANS_F32"], "int32": k["SQ_INSTS_VALU_INT32"],
         "int64": k["SQ_INSTS_VALU_INT64"], "cvt": k["SQ_INSTS_VALU_CVT"],
         "f64": (k.get("SQ_INSTS_VALU_ADD_F64") or 0) + (k.get("SQ_INSTS_VALU_MUL_F64") or 0) + (k.get("SQ_INSTS_VALU_FMA_F64") or 0), "trans_f64": k.get("SQ_INSTS_VALU_TRANS_F64") or 0}
    other = max(0.0, k["SQ_INSTS_VALU"] - sum(n.values()))
    counted = sum(ISSUE_CYCLES[c] * v for c, v in n.items())
    available = 256 * 4 * kernel_ms * 1e-3 * 2.4e9
    lo, est, hi = [(counted + c * other) / available for c in ISSUE_CYCLES_OTHER]
    return {"frac_low": lo, "frac": est, "frac_high": hi, "instructions_by_class": dict(n, other=other), "cycles_per_class": dict(ISSUE_CYCLES, other=list(ISSUE_CYCLES_OTHER)),
            "simd_cycles_available": available,
            "note": "the kernel's vector instructions priced by what their class costs to issue on gfx950 (measured: profiles/r04_valu_rates.txt) over the SIMD cycles of the launch: the share of the launch in "
                    "which the vector pipes are issuing; `frac` above prices every instruction as an FMA (1 per 2 cycles).  `other` = SQ_INSTS_VALU minus the class counters (compares, min / max, selects, "
                    "moves, logic, shifts): 2 .. 4 cycles, 3.4 for the estimate"}


def algorithmic_bytes(cnt, n_lights, pixels, use_filter):
    """B_frame = 48 N_visit + 160 N_shade + 24 L N_shade + 4 N_tex + B_out W H (+ B_filter)."""
    visits = cnt["primary_visits"] + cnt["closest_visits"] + cnt["shadow_visits"]
    b = 48 * visits + 160 * cnt["shades"] + 24 * n_lights * cnt["shades"] + 4 * cnt["atlas_texels"]
    b += (20 if use_filter else 16) * pixels
    if use_filter:
        b += 224 * pixels
    return b


def _run_group(cmd, cwd, env, timeout):
    """run a child in its own process group; on timeout the WHOLE group is killed and waited for (rocprofv3's grandchild would
    otherwise keep running on the GPU beside the timed region).  -> (returncode or None on timeout, stderr tail)"""
    p = subprocess.Popen(cmd, cwd=cwd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, start_new_session=True)
    try:
        _, err = p.communicate(timeout=timeout)
        return p.returncode, err.decode(errors="replace")[-300:]
    except subprocess.TimeoutExpired:
        for sig in (signal.SIGTERM, signal.SIGKILL):
            try:
                os.killpg(p.pid, sig)
            except ProcessLookupError:
                break
            try:
                p.communicate(timeout=20)
                break
            except subprocess.TimeoutExpired:
                continue
        if p.poll() is None:
            raise SystemExit("bench.py: a rocprofv3 pass hung and did not die when killed; not touching the GPU after it")
        return None, "timed out after %d s (process group killed)" % timeout


def collect_pmc(args, tile=None):
    """rocprofv3 --pmc passes over tools/pmc_pass.py (the same workload — a rank's strips of it when `tile` = (rows, index, count) —
    one frame per launch, through the same library) plus one --kernel-trace --stats pass for the kernels' average durations, run as
    child processes BEFORE this process initialises the GPU.  Returns {kernel name: {counter: mean per dispatch}}, plus "_meta";
    {} with a reason when rocprofv3 is missing or a pass fails (the bench line then carries nulls)."""
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not exe:
        return {"_meta": {"error": "rocprofv3 not found"}}
    acc, commands = {}, []
    work = tempfile.mkdtemp(prefix="flx_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):          # the profiled child is a plain one-GPU program
        env.pop(k, None)
    child = ["--workload", args.workload, "--frames", "3"]
    if args.width:
        child += ["--width", str(args.width)]
    if args.height:
        child += ["--height", str(args.height)]
    if tile:
        child += ["--tile-rows", str(tile[0]), "--tile-index", str(tile[1]), "--tile-count", str(tile[2])]

    def clean(name):
        return name.split("(")[0].replace("void ", "").replace("flx::", "").strip()
    class_error = None
    try:
        passes = [list(c) for c in PMC_PASSES]
        i = 0
        while i < len(passes):
            counters = passes[i]
            out = os.path.join(work, "p%d" % i)
            shutil.rmtree(out, ignore_errors=True)
            cmd = [exe, "--pmc"] + counters + ["--output-format", "csv", "-d", out, "--", sys.executable, os.path.join(ROOT, "tools", "pmc_pass.py")] + child
            line = "rocprofv3 --pmc %s --output-format csv -d <dir> -- python3 tools/pmc_pass.py %s" % (" ".join(counters), " ".join(child))
            rc, err = _run_group(cmd, "/tmp", env, 420)
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if rc != 0 or not files:
                if i == 0 and "SQ_THREAD_CYCLES_VALU" in counters:       # a counter this rocprofv3 does not know: the pass once more without it
                    passes[0] = [c for c in counters if c != "SQ_THREAD_CYCLES_VALU"]
                    continue
                commands.append(line)
                if i >= 3:                                                  # the per-class passes are an extra: without them the line carries no `issue_cycles`, everything else stands
                    class_error = "pmc pass %d failed (rc %s): %s" % (i, rc, err)
                    break
                return {"_meta": {"error": "pmc pass %d failed (rc %s): %s" % (i, rc, err), "commands": commands}}
            commands.append(line)
            per = {}
            for path in files:
                with open(path) as fh:
                    for row in csv.DictReader(fh):
                        per.setdefault(clean(row["Kernel_Name"]), {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
            for name, ctrs in per.items():
                for c, v in ctrs.items():
                    acc.setdefault(name, {})[c] = sum(v) / len(v)
                    acc[name]["dispatches"] = len(v)
            i += 1
        # the kernels' average durations as rocprofv3's kernel trace sees them (beside the HIP-event time of the timed run)
        out = os.path.join(work, "kt")
        kt_child = [c if c != "3" else str(KT_FRAMES) for c in child]
        cmd = [exe, "--kernel-trace", "--stats", "--output-format", "csv", "-d", out, "--", sys.executable, os.path.join(ROOT, "tools", "pmc_pass.py")] + kt_child
        rc, err = _run_group(cmd, "/tmp", env, 420)
        files = glob.glob(os.path.join(out, "**", "*kernel_stats.csv"), recursive=True)
        if rc == 0 and files:
            commands.append("rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python3 tools/pmc_pass.py %s" % " ".join(kt_child))
            with open(files[0]) as fh:
                for row in csv.DictReader(fh):
                    acc.setdefault(clean(row["Name"]), {})["kernel_trace_avg_ms"] = float(row["AverageNs"]) / 1e6
            # the same pass's per-dispatch trace: the summary's average includes the first launches (cold caches, code upload); the steady state is
            # the dispatches after the program's warm-up frames — their median and minimum are what the timed steps see
            for tf in glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True):
                per = {}
                with open(tf) as fh:
                    for row in csv.DictReader(fh):
                        try:
                            per.setdefault(clean(row["Kernel_Name"]), []).append((int(row["Start_Timestamp"]), int(row["End_Timestamp"])))
                        except (KeyError, ValueError):
                            continue
                for name, spans in per.items():
                    spans.sort()
                    steady = [(b - a) / 1e6 for a, b in spans[PMC_WARMUP_FRAMES * max(1, len(spans) // (PMC_WARMUP_FRAMES + KT_FRAMES)):]] or [(b - a) / 1e6 for a, b in spans]
                    steady.sort()
                    acc.setdefault(name, {}).update(kernel_trace_median_ms=steady[len(steady) // 2], kernel_trace_min_ms=steady[0], kernel_trace_dispatches=len(spans))
    finally:
        shutil.rmtree(work, ignore_errors=True)
    acc["_meta"] = {"commands": commands, "collected": "by this bench.py run, before its timed region, one frame per launch" + (" (rank 0's strips: tile %s)" % (tile,) if tile else "")}
    if class_error:
        acc["_meta"]["class_passes"] = class_error
    return acc


def launch_ranks(args):
    """`python bench.py --gpus N` without torch.distributed.run around it: this process — which never touches a GPU — runs rank 0's
    counter passes, then starts the N ranks as fresh child processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set), relays
    rank 0's JSON line and exits with the worst exit code."""
    n = args.gpus
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env0 = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(n), FLX_BENCH_SELF_LAUNCHED="1")
    pmc_file = None
    if not args.no_pmc:
        pmc = collect_pmc(args, tile=(args.tile_rows, 0, n))
        fd, pmc_file = tempfile.mkstemp(prefix="flx_pmc_", suffix=".json", dir="/tmp")
        with os.fdopen(fd, "w") as fh:
            json.dump(pmc, fh)
        env0["FLX_BENCH_PMC_JSON"] = pmc_file
    procs = []
    try:
        for r in range(n):
            env = dict(env0, RANK=str(r), LOCAL_RANK=str(r))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
        # rank 0's line is read by a thread; this process polls EVERY rank: the first one that fails, or the wall-clock limit, ends them all
        # (fresh child processes are killed, nothing that has touched a GPU is re-executed)
        import threading
        chunks = []
        reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
        reader.start()
        deadline = time.time() + args.rank_timeout
        failed = None
        while any(p.poll() is None for p in procs):
            bad = [r for r, p in enumerate(procs) if p.poll() not in (None, 0)]
            if bad or time.time() > deadline:
                failed = ("rank %d exited with %s" % (bad[0], procs[bad[0]].returncode)) if bad else ("the ranks did not finish within %d s" % args.rank_timeout)
                for p in procs:
                    if p.poll() is None:
                        p.terminate()
                t_kill = time.time() + 10
                while any(p.poll() is None for p in procs) and time.time() < t_kill:
                    time.sleep(0.1)
                for p in procs:
                    if p.poll() is None:
                        p.kill()
                break
            time.sleep(0.2)
        reader.join(timeout=10)
        out = b"".join(c for c in chunks if c)
        rcs = [p.wait() for p in procs]
        if failed:
            sys.stderr.write("bench.py: %s; the remaining ranks were stopped\n" % failed)
            rcs = [rc if rc else 1 for rc in rcs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        if pmc_file:
            try:
                os.unlink(pmc_file)
            except OSError:
                pass
    sys.stdout.write(out.decode(errors="replace"))
    sys.stdout.flush()
    worst = max((abs(rc) for rc in rcs), default=0)
    if worst:
        sys.stderr.write("bench.py: rank exit codes %s\n" % rcs)
    raise SystemExit(1 if worst else 0)


def usable_cores():
    """(cores this process can really run on, its cgroup CPU quota or None): the affinity mask (= nproc) capped by the cgroup's
    quota — a GPU box hands a one-GPU job 16 of the host's 256 cores that way, and 256 OpenMP threads on a 16-core quota are
    slower than 16 (28.8 against 46 Mray/s on the dragon frame)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(period)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / period
        except (OSError, ValueError):
            pass
    if quota:
        n = max(1, min(n, int(quota + 0.5)))
    return n, quota


def cpu_baseline(scene, params_full, seconds_budget=12.0, js_workload=None):
    """CPU oracle (kind 'port': this repo's C restatement of the reference GLSL — the reference has no CPU path) on a bounded
    sample of the same workload, on ALL host cores of this box: the same scene / spp / bounces, the full frame when one frame
    fits the budget (repeated until ~seconds_budget of CPU work), otherwise a centred sub-resolution frame sized from a probe."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import flx_oracle
    threads, quota = usable_cores()
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
    # one core beside it: every 16th strip of the same frame, ~1/16 of its rays, a few seconds
    p1 = scene.frame_params(width=w, height=h, samples=spp, max_reflections=bounces, use_filter=0, tile=(8, 0, 16))
    rows1 = len(flx_oracle.tile_rows(p1))
    t1 = time.time()
    flx_oracle.render(scene, p1, threads=1)
    dt1 = max(time.time() - t1, 1e-6)
    js = js_baseline(scene, params_full) if js_workload else None
    return {
        **({"js": js} if js else {}),
        "value": frames * spp * bounces * w * h / dt / 1e6, "unit": "Mray/s", "cores": threads, "kind": "port", "nproc": os.cpu_count(), "cpu_quota": quota,
        "sample": "same scene/spp/bounces, %dx%d frame x %d (%.1f s of CPU oracle, %d OpenMP threads = every core this process can run on: nproc %d, cgroup quota %s; filter off)" % (w, h, frames, dt, threads, os.cpu_count() or 0, ("%.1f cores" % quota) if quota else "none"),
        "single_thread": {"value": spp * bounces * w * rows1 / dt1 / 1e6, "unit": "Mray/s", "cores": 1,
                          "sample": "every 16th 8-row strip of that frame (%d rows, %.1f s)" % (rows1, dt1)},
    }


def js_baseline(scene, params_full, seconds_budget=6.0):
    """BASELINE configs[0] names "the reference JS/CPU headless path"; the reference has no CPU renderer (its only CPU ray code is modules/math.js:113-137), so this is
    THIS REPOSITORY'S restatement of the fragment program in plain JavaScript (oracle/js/flx_oracle_js.js: bit-identical to the C oracle, tests/test_oracle_js_cpu.py),
    single thread under Node, on the whole configs[0] frame repeated for a few seconds: what a JavaScript fallback of the reference's shader would cost."""
    import shutil
    node = shutil.which("node")
    if not node:
        return {"error": "no node on this box"}
    import tempfile
    p = params_full
    pj = {"width": int(p.width), "height": int(p.height), "camera": [float(v) for v in p.camera], "view_matrix": [float(v) for v in p.view_matrix], "samples": int(p.samples),
          "max_reflections": int(p.max_reflections), "min_importancy": float(p.min_importancy), "ambient": [float(v) for v in p.ambient], "random_seed": float(p.random_seed),
          "texture_width": int(p.texture_width), "use_filter": 0, "is_temporal": 0}
    fixture = os.path.join(ROOT, "tests", "golden", "ref_%s.flxs.gz" % scene.meta.get("name", "cornell"))
    with tempfile.TemporaryDirectory(prefix="flx_js_") as tmp:
        pf = os.path.join(tmp, "params.json")
        with open(pf, "w") as fh:
            json.dump(pj, fh)
        cmd = [node, os.path.join(ROOT, "oracle", "js", "flx_oracle_js.js"), fixture, pf]
        try:
            one = json.loads(subprocess.check_output(cmd + ["--repeat", "1"], timeout=120).decode().splitlines()[-1])
            reps = max(1, min(50, int(seconds_budget * 1e3 / max(one["ms_per_frame"], 1e-3))))
            r = json.loads(subprocess.check_output(cmd + ["--repeat", str(reps)], timeout=180).decode().splitlines()[-1])
        except Exception as e:      # noqa: BLE001 — a baseline that cannot run is reported, not fatal
            return {"error": repr(e)[:200]}
    return {"value": r["mray_per_s"], "unit": "Mray/s", "cores": 1, "kind": "port (JavaScript)", "ms_per_frame": r["ms_per_frame"], "node": r.get("node"),
            "sample": "the whole frame x %d, filter off, one thread under Node (oracle/js/flx_oracle_js.js: this repository's restatement of the shader — the reference has no JS/CPU renderer)" % r["frames"]}


def moved(scene, p, i):
    """frame i of a camera move (position, view direction, ambient): the frames of a batch are different frames"""
    from flexlight_hip.scene_io import view_matrix
    cam = scene.meta["camera"]
    q = type(p).from_buffer_copy(p)
    q.camera[:] = [cam["x"] + 0.05 * i, cam["y"] + 0.02 * i, cam["z"] - 0.03 * i]
    q.view_matrix[:] = view_matrix(cam["fx"] + 0.004 * i, cam["fy"] - 0.002 * i, cam["fov"], p.width, p.height).tolist()
    return q


def shared_loop(args, dist, capi, scene, params, full, rank, world, local_rank, frames, lanes=3):
    """The frame loop of N ranks without a collective (include/flexlight_hip.h: flx_share_*): -> the `shared` entry of the line (rank 0; None elsewhere).
    Library calls sit in try blocks and every torch.distributed call outside them is reached by every rank whatever failed, so a rank that fails (the
    others then fail within flx_share's 5 s instead of waiting) cannot wedge the run."""
    import ctypes
    import numpy as np
    import torch
    W, H = full.width, full.height
    err, ctx_s, handle, last_ptr, lat, served = None, None, None, None, [], None

    def agree(ok):
        t = torch.tensor([1 if ok else 0], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    def run(n):
        nonlocal last_ptr
        inflight = 0
        for _ in range(n):
            if inflight == lanes:
                last_ptr, ms = ctx_s.frame_end_shared()
                lat.append(ms)
                inflight -= 1
            ctx_s.frame_begin_shared(params)
            inflight += 1
        while inflight:
            last_ptr, ms = ctx_s.frame_end_shared()
            lat.append(ms)
            inflight -= 1

    try:
        ctx_s = capi.Context(local_rank)
        ctx_s.update_scene(scene)
        if args.one_device:                      # rehearsal on one GPU: every rank's server launch takes its part of the CUs, so that they run at once
            ctx_s.set_server_groups(max(1, ctx_s.device_info()[1] // world))
        served = True                                  # (else: every rank's two lanes, its strips copied into the image when the frame is taken; the library decides from every rank's share)
        for r in range(world):
            q = type(params).from_buffer_copy(params)
            q.tile_index = r
            served = served and ctx_s.frame_server_takes(q)
        if rank == 0:
            handle = ctx_s.share_create(W, H, lanes, world, 0)
    except Exception as e:                       # noqa: BLE001 — reported in the line
        err = "%s: %s" % (type(e).__name__, e)
    box = [handle]
    dist.broadcast_object_list(box, src=0)
    try:
        if err is None and rank != 0:
            if box[0] is None:
                raise RuntimeError("rank 0 could not create the share")
            ctx_s.share_join(box[0], rank)
    except Exception as e:                       # noqa: BLE001
        err = "%s: %s" % (type(e).__name__, e)
    ok = agree(err is None and box[0] is not None)
    dt, equal = None, None
    if ok:
        try:
            run(2 * lanes)                       # warm-up: the workspaces of the frames in flight, the server's first launch
        except Exception as e:                   # noqa: BLE001
            err = "%s: %s" % (type(e).__name__, e)
        torch.cuda.synchronize()
        dist.barrier()
        lat.clear()
        t1 = time.perf_counter()
        try:
            if err is None:
                run(frames)
        except Exception as e:                   # noqa: BLE001
            err = "%s: %s" % (type(e).__name__, e)
        torch.cuda.synchronize()
        dist.barrier()
        dt = time.perf_counter() - t1
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        ok = agree(err is None)
        if ok and rank == 0 and last_ptr:
            try:                                 # the image the loop's last frame left in rank 0's memory against one context's render of the whole frame
                try:
                    hiprt = ctypes.CDLL("libamdhip64.so")
                except OSError:
                    hiprt = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")
                got = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
                whole = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
                torch.cuda.synchronize()
                if hiprt.hipMemcpy(ctypes.c_void_p(got.data_ptr()), ctypes.c_void_p(last_ptr), ctypes.c_size_t(H * W * 16), 3) != 0:
                    raise RuntimeError("hipMemcpy of the shared image")
                ctx_v = capi.Context(local_rank)
                ctx_v.update_scene(scene)
                ctx_v.render_device(full, whole.data_ptr())
                ctx_v.sync()
                ctx_v.close()
                equal = bool(torch.equal(whole.view(torch.int32), got.view(torch.int32)))
            except Exception as e:               # noqa: BLE001
                err = "verification: %s: %s" % (type(e).__name__, e)
    try:
        if ctx_s is not None:
            ctx_s.close()                        # (leaves the share)
    except Exception:                            # noqa: BLE001
        pass
    errs = [None] * world
    dist.all_gather_object(errs, err)            # (also the barrier behind the loop: every rank has left the share)
    if rank != 0:
        return None
    if not ok or dt is None:
        return {"error": "; ".join("rank %d: %s" % (r, e) for r, e in enumerate(errs) if e) or "a rank failed", "frames_in_flight": lanes}
    return {"frames_in_flight": lanes, "frame_server": bool(served), "ms_per_frame": dt / frames * 1e3, "frames": frames, "frame_gpu_ms_median": float(np.median(lat[-frames:])) if lat else None,
            "image_equals_single_context_frame": equal, "error": err,
            "note": "flx_frame_begin_shared / flx_frame_end_shared on every rank: each rank's frame server (one persistent launch, %d frames in flight) resolves its strips straight "
                    "into one image in rank 0's device memory (hipIpc mapping, stores over xGMI) — no collective, no reassembly kernel, no copy (`frame_server` false: a frame the "
                    "server does not take — every rank's two lanes, its strips copied into that image when the frame is taken); rank 0 hands a frame out when every "
                    "rank's strips of it are complete%s" % (lanes, "; REHEARSAL on one device: every rank's launch takes 1/%d of the CUs" % world if args.one_device else "")}


class SecondaryGuard:
    """A wall-clock limit over the secondary measurements (main(): "measured LAST"): when it runs out, rank 0 prints the line it has — with `secondary_incomplete`
    naming the phase that did not come back — and every rank's process ends (os._exit: the main thread may be stuck inside a library call)."""

    def __init__(self, rank, line, seconds):
        import threading
        self.rank, self.line, self.seconds = rank, line, seconds
        self.lock = threading.Lock()
        self.current, self.timer, self.finished = None, None, False

    def start(self):
        import threading
        if self.seconds > 0:
            self.timer = threading.Timer(self.seconds, self._fire)
            self.timer.daemon = True
            self.timer.start()

    def phase(self, name):
        with self.lock:
            self.current = name
        if os.environ.get("FLX_BENCH_TEST_HANG") == name:      # tests: this phase never comes back
            time.sleep(1e6)

    def put(self, key, value):
        with self.lock:
            if self.line is not None:
                self.line[key] = value

    def done(self):
        with self.lock:
            self.finished = True
        if self.timer:
            self.timer.cancel()

    def _fire(self):
        with self.lock:
            if self.finished:
                return
            sys.stderr.write("bench.py: rank %d: the secondary measurement `%s` did not finish within %d s; ending the run with the line as it stands\n" % (self.rank, self.current, self.seconds))
            if self.rank == 0 and self.line is not None:
                self.line["secondary_incomplete"] = {"phase": self.current, "limit_s": self.seconds}
                print(json.dumps(self.line), flush=True)
            sys.stderr.flush()
            bad = self.rank == 0 and self.line is not None and self.line.get("gathered_frame_equals_single_context_frame") is False
            os._exit(3 if bad else 0)                  # (the line is valid without the phase that hung; a failed verification still fails the run)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="dragon", choices=sorted(WORKLOADS))
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--tile-rows", type=int, default=8)
    ap.add_argument("--batch", type=int, default=16, help="frames per pass of the `batched` secondary measurement (flx_render_batch_device; at most 32 and at most 2^28 paths per pass); 0 = skip it")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 counter passes (roofline.achieved / traffic are then null)")
    ap.add_argument("--one-device", action="store_true", help="rehearsal of the rank logic on a one-GPU box: every rank uses GPU 0 and the strips are gathered with torch.distributed (gloo) instead of RCCL, which refuses two ranks on one device")
    ap.add_argument("--verify", action="store_true", help="rank 0 renders the whole frame on its own after the run and compares the gathered frame with it, bit for bit (the default for N > 1)")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--rank-timeout", type=int, default=900, help="N > 1 launched by bench.py itself: seconds after which ranks still running are stopped and the run fails")
    ap.add_argument("--headline", default="one_frame_per_pass", choices=["auto", "one_frame_per_pass"],
                    help="kept for old command lines: `value` / `ms_per_step` are the timed region (one frame per pass) either way; the frame loops are `pipelined` / `shared` / `throughput_best`")
    ap.add_argument("--secondary-timeout", type=int, default=300, help="seconds the measurements after the timed region (`pipelined`, `shared`, `batched`) may take in all before the run ends with the line as it stands; 0 = no limit")
    ap.add_argument("--no-shared", action="store_true", help="N > 1: skip the frame loop without a collective (flx_share_*: the `shared` entry of the line)")
    ap.add_argument("--gather", choices=["root", "all"], default="root", help="N > 1: root = only rank 0, which presents the frame, receives the strips (ncclSend / ncclRecv; the reference presents from its one context); all = ncclAllGather, every rank ends up with the frame")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args)                   # never returns
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world
    # FLX_BENCH_FORCE_DIST=1 with one rank: the N > 1 code path — gloo bootstrap, the communicator id handed around, ncclCommInitRank,
    # flx_render_gathered_device in the timed loop — rehearsed on a one-GPU box with a communicator of one rank
    multi = world > 1 or os.environ.get("FLX_BENCH_FORCE_DIST") == "1"

    # Counter passes first: child processes under rocprofv3, before this process touches the GPU.
    pmc = {}
    if not args.no_pmc and rank == 0:
        if os.environ.get("FLX_BENCH_PMC_JSON"):               # collected by the launching process (launch_ranks) before any rank existed
            with open(os.environ["FLX_BENCH_PMC_JSON"]) as fh:
                pmc = json.load(fh)
        else:                                                    # N = 1, or rank 0 under torch.distributed.run (the other ranks wait at the rendezvous)
            pmc = collect_pmc(args, tile=(args.tile_rows, 0, world) if world > 1 else None)
    verify = (args.verify or multi) and not args.no_verify

    import numpy as np
    import torch
    from flexlight_hip import capi, tiles
    from flexlight_hip.scene_io import Scene

    if args.one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if multi:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        import datetime
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(minutes=30))      # bootstrap + barrier only; the data path is RCCL inside the library

    fixture, config_name = WORKLOADS[args.workload][:2]
    if len(WORKLOADS[args.workload]) > 2 and args.width is None and args.height is None:
        args.width, args.height = WORKLOADS[args.workload][2:]
    scene = Scene.golden(fixture)
    full = scene.frame_params(width=args.width, height=args.height)
    use_filter = int(full.use_filter)
    W, H = full.width, full.height
    tile = (args.tile_rows, rank, world) if multi else (0, 0, 0)
    params = scene.frame_params(width=args.width, height=args.height, tile=tile)

    ctx = capi.Context(local_rank)
    ctx.update_scene(scene)
    stream = torch.cuda.Stream()              # a real (non-null) stream: kernels, RCCL gather and reassembly are all enqueued on it
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)
    rccl = multi and not args.one_device
    to_root = rccl and args.gather == "root"
    if rccl:
        ids = [capi.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        ctx.comm_init_rank(ids[0], world, rank)

    rows_local = ctx.tile_row_count(params)
    F = 0 if (use_filter and multi) else max(0, min(args.batch, capi.MAX_BATCH_FRAMES))      # (gathered filter frames go one by one)
    if F:
        F = max(1, min(F, (1 << 28) // max(1, (tiles.padded_rows(H, args.tile_rows, world) if multi else H) * W * full.samples)))
    frames_out = torch.zeros((max(F, 1), H, W, 4), dtype=torch.float32, device="cuda")      # whole frames (gathered when N > 1)
    if args.one_device and multi:            # rehearsal transport: gloo all-gather of the packed strips + the index table of flexlight_hip/tiles.py
        rows_max = tiles.padded_rows(H, args.tile_rows, world)
        rows_of = []
        for r in range(world):
            pr = scene.frame_params(width=args.width, height=args.height, tile=(args.tile_rows, r, world))
            rows_of.append(list(capi.Context.tile_rows(pr)))
        local = torch.zeros((max(F, 1), rows_max, W, 4), dtype=torch.float32, device="cuda")
        perms = {}

        def gather_gloo(plist):
            f = len(plist)
            if use_filter:
                raise SystemExit("--one-device rehearses frames without filter")
            ctx.render_batch_device(plist, local.data_ptr())
            ctx.sync()
            n = f * rows_max * W * 4
            parts = [torch.empty(n, dtype=torch.float32) for _ in range(world)]
            dist.all_gather(parts, local.view(-1)[:n].cpu())
            if f not in perms:
                perms[f] = torch.from_numpy(tiles.gather_index(rows_of, f, rows_max, H))
            g = torch.cat(parts).view(world * f * rows_max, W, 4)
            frames_out.view(-1, W, 4)[:f * H] = torch.index_select(g, 0, perms[f]).cuda()

    def render(plist):
        """the frames of plist in ONE pass: whole frames land in frames_out (device memory), nothing waits on the host"""
        if rccl and to_root:
            ctx.render_gathered_root_device(plist, 0, frames_out.data_ptr())
        elif rccl:
            ctx.render_gathered_device(plist, frames_out.data_ptr())
        elif multi:
            gather_gloo(plist)
        elif len(plist) == 1:
            ctx.render_device(plist[0], frames_out.data_ptr())
        else:
            ctx.render_batch_device(plist, frames_out.data_ptr())

    def fence():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
            torch.cuda.synchronize()

    batch_params = [moved(scene, params, i) for i in range(F)] if F else []
    if F:
        render(batch_params)                 # untimed: the context sizes its workspace for the largest pass here, not in a timed region
    for _ in range(args.warmup):
        render([params])
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        render([params])                     # EXACTLY K steps: one frame per pass, frame after frame
    fence()
    elapsed = time.perf_counter() - t0
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    verified = None
    if verify and multi and rank == 0:
        torch.cuda.synchronize()
        whole = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
        ctx.render_device(full, whole.data_ptr())
        ctx.sync()
        verified = bool(torch.equal(whole.view(torch.int32), frames_out[0].view(torch.int32)))

    # N > 1: the same frame on ONE context (rank 0's GPU alone, the other ranks wait at a barrier), one frame per pass: the denominator of the speed-ups the line states in
    # both currencies (`speedup`: latency mode = the timed region, gather included; throughput mode = the fastest verified frame loop, with its frames in flight)
    single_ms = None
    if multi and world > 1:
        dist.barrier()
        if rank == 0:
            torch.cuda.synchronize()
            whole1 = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
            for _ in range(3):
                ctx.render_device(full, whole1.data_ptr())
            ctx.sync()
            t1 = time.perf_counter()
            n1 = max(5, min(args.steps, 20))
            for _ in range(n1):
                ctx.render_device(full, whole1.data_ptr())
            ctx.sync()
            single_ms = (time.perf_counter() - t1) / n1 * 1e3
            del whole1
        dist.barrier()

    # ---- secondary measurements, outside the timed region --------------------------------------------------
    # per-frame GPU time between HIP events on the launch stream (median of >= 20 frames) and the dominant kernel's
    gpu_ms, kernel_ms = [], []
    for _ in range(max(20, min(args.steps, 40))):
        render([params])
        a, b = ctx.last_frame_ms()
        gpu_ms.append(a)
        kernel_ms.append(b)
    # work counters of ONE frame (this rank's share), a counted launch
    ctx.set_counters_enabled(True)
    if rccl:
        render([params])
    else:
        local1 = torch.zeros((max(rows_local, 1), W, 4), dtype=torch.float32, device="cuda")
        ctx.render_device(params, local1.data_ptr())
    ctx.sync()
    cnt = ctx.get_counters()
    b0_visits = ctx.get_diag()[15] if not use_filter else 0      # entries the round-0 walk kernel visited (its own tally; 0 when another kernel walked)
    pipe = ctx.last_pipeline()
    organisation = ctx.last_organisation()                      # wavefront pipeline: 1 rounds, 2 frame kernel, 3 frame kernel with the front of the frame inside
    ctx.set_counters_enabled(False)
    torch.cuda.synchronize()

    line, rays = None, 0
    if rank == 0:
        spp, bounces = full.samples, full.max_reflections
        rays = spp * bounces * W * H
        ms_per_step = elapsed / args.steps * 1e3
        value = rays / (elapsed / args.steps) / 1e6
        n_lights = scene.arrays["lights"].size // 6
        k_ms = float(np.median(kernel_ms))
        frame_bytes = algorithmic_bytes(cnt, n_lights, rows_local * W, use_filter)      # this rank's share of one frame
        if pipe == 1:              # per-pixel kernel: the whole trace
            # k_trace_samples<COUNT, LOCK, S> where the frame has 2, 4 or 8 samples and at most 4 bounces (a pixel's samples side by side), else k_trace_pixels<COUNT, LOCK>
            kernel_sym = next((n for n in pmc if n.startswith("k_trace_samples<false")), None) or next((n for n in pmc if n.startswith("k_trace_pixels<false")), "k_trace_pixels<false, false>")
            kernel_name, bytes_launch = kernel_sym.split("<")[0], frame_bytes - (244 * rows_local * W if use_filter else 0)
        elif pipe == 2:            # persistent path kernel (tiny scenes): every bounce of every path, no primary walk, no output
            kernel_name, kernel_sym = "k_paths (persistent path kernel)", next((n for n in pmc if n.startswith("k_paths<false")), "k_paths<false, false>")
            bytes_launch = 48 * (cnt["closest_visits"] + cnt["shadow_visits"]) + (160 + 24 * n_lights) * cnt["shades"] + 4 * cnt["atlas_texels"]
        elif organisation >= 2:
            # the frame kernel (its tally holds EVERY bounce's visits), <COUNT, FRONT>: all bounce walks of the frame and the shading of bounces >= 1 in one
            # persistent launch — and, with the front of the frame inside it, the primary rays and the bounce-0 shading as well
            if organisation == 3:
                kernel_name, kernel_sym = "k_wf_frame<false, true> (frame kernel: primary rays, every bounce's shading, all bounce walks)", "k_wf_frame<false, true>"
                bytes_launch = 48 * (cnt["closest_visits"] + cnt["shadow_visits"] + cnt["primary_visits"]) + (160 + 24 * n_lights) * cnt["shades"]
            else:
                later_shades = max(0, cnt["shades"] - spp * cnt["primary_hits"])
                kernel_name, kernel_sym = "k_wf_frame<false, false> (frame kernel: all bounce walks + shading of bounces >= 1)", "k_wf_frame<false, false>"
                bytes_launch = 48 * (cnt["closest_visits"] + cnt["shadow_visits"]) + (160 + 24 * n_lights) * later_shades
        else:                      # rounds: the bounce-0 walk kernel's share of B_frame: the 48-byte entries its walks visit
            visits = b0_visits if b0_visits else cnt["closest_visits"] + cnt["shadow_visits"]
            kernel_name, kernel_sym, bytes_launch = "k_wf_walk_pre<false, true> (walk kernel of bounce 0)", "k_wf_walk_pre<false, true>", 48 * visits
        k = pmc.get(kernel_sym, {})
        valu = k.get("SQ_INSTS_VALU")
        achieved = valu / (k_ms * 1e-3) if valu else None
        fetch_kb, write_kb = k.get("FETCH_SIZE"), k.get("WRITE_SIZE")
        # gfx950: FETCH_SIZE tallies the L2's 128-byte fabric read requests at 64 B each (MI355X_MICROARCH.md, HBM; confirmed for this kernel's
        # record pieces by TCC_EA0_RDREQ_128B x 128 B = 2 x FETCH_SIZE, profiles/r03_pmc_cache.txt): doubled.  WRITE_SIZE is exact.
        traffic = (2.0 * fetch_kb + write_kb) * 1024.0 if fetch_kb is not None and write_kb is not None else None
        lane_util = None
        if k.get("SQ_THREAD_CYCLES_VALU") and k.get("SQ_ACTIVE_INST_VALU"):
            lane_util = k["SQ_THREAD_CYCLES_VALU"] / (64.0 * k["SQ_ACTIVE_INST_VALU"])
        frac = achieved / VALU_ISSUE_PEAK if achieved else None
        walks = cnt["closest_walks"] + cnt["shadow_walks"]
        segments = walks + cnt["primary_hits"]                 # rays actually traced through the tree: bounce walks + the primary rays that hit
        line = {
            "metric": "Mray/s at 1080p (spp x bounces x pixels / s)", "value": value, "unit": "Mray/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": DATA[args.workload],
            "headline": {"mode": "one_frame_per_pass", "frames_in_flight": 1, "note": "`value` / `ms_per_step`: K frames, one after the other, each complete (and gathered) before the next begins"},
            "config": {
                "workload": config_name, "width": W, "height": H, "spp": spp, "bounces": bounces, "filter": bool(use_filter),
                "scene_entries": int(scene.meta["textureLength"]),
                "parallelism": ("row-strip tiles x%d, %d rows/strip, %s" % (world, args.tile_rows, (("ncclSend / ncclRecv to rank 0 (flx_render_gathered_root_device)" if to_root else "ncclAllGather (flx_render_gathered_device)") + " + reassembly kernel inside libflexlight_hip.so") if rccl else "REHEARSAL: strips gathered over gloo on one device")) if multi else "single GPU",
                "rays_per_frame": rays, "frames_per_pass": 1,
                "frames": "one frame per pass, frame after frame (the static camera of the BASELINE config, as in the reference's frame loop); every frame is traced in full, nothing is reused between frames",
            },
            "frame_gpu_ms": {"median": float(np.median(gpu_ms)), "min": float(np.min(gpu_ms)), "max": float(np.max(gpu_ms)), "frames": len(gpu_ms),
                             "note": "HIP events on the launch stream around one frame: first kernel .. last byte of the (gathered) frame on this rank"},
            "roofline": {
                "bound": "valu_issue", "kernel": kernel_name, "achieved": achieved, "peak": VALU_ISSUE_PEAK, "unit": "wave-instr/s",
                "frac": frac,
                "frac_is": "VALU wave-instructions issued per second over the part's issue rate (NOT SURVEY.md 8d's HBM form: see survey_8d_frac)",
                "effective_lane_frac": frac * lane_util if frac and lane_util else None,      # frac x the share of lanes an executed vector instruction has switched on
                "survey_8d_frac": bytes_launch / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "survey_8d_frac_note": "SURVEY.md 8d's figure (48 B x entries visited / kernel time / 8 TB/s): not physical when it nears or exceeds 1 — the scene is LDS / L2 resident, the visits never reach HBM (hbm_measured is what does)",
                "traffic": traffic, "kernel_ms": k_ms, "kernel_ms_rocprofv3_kernel_trace": k.get("kernel_trace_avg_ms"),
                "kernel_ms_rocprofv3_steady": {"median": k.get("kernel_trace_median_ms"), "min": k.get("kernel_trace_min_ms"), "dispatches": k.get("kernel_trace_dispatches"),
                                               "note": "the kernel trace's dispatches after tools/pmc_pass.py's %d warm-up frames, output left in device memory (the summary's average above includes the first launches)" % PMC_WARMUP_FRAMES},
                "valu_insts_per_launch": valu, "valu_lane_utilisation": lane_util,
                "issue_cycles": issue_cycles(k, k_ms),
                "hbm_measured": {"bytes_per_launch": traffic, "GBps": traffic / (k_ms * 1e-3) / 1e9 if traffic else None, "peak_GBps": HBM_PEAK_GBS,
                                 "frac": traffic / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if traffic else None,
                                 "note": "2 x FETCH_SIZE + WRITE_SIZE (KB, separate rocprofv3 passes of this run): on gfx950 FETCH_SIZE counts 64 B per 128-byte fabric read request (guide's HBM section; TCC_EA0_RDREQ_128B of this kernel in profiles/r03_pmc_cache.txt); Infinity-Cache hits are included, so HBM proper is at most this"},
                "algorithmic": {"bytes_per_launch": bytes_launch, "GBps": bytes_launch / (k_ms * 1e-3) / 1e9,
                                "frame_bytes": frame_bytes, "frame_GBps": frame_bytes / (ms_per_step * 1e-3) / 1e9,
                                "note": "SURVEY.md 8d: 48 B x entries visited (+ 160 B x shades + 24 B x lights x shades + 4 B x texels + 16 B x pixels for the frame), from this frame's work counters; the <= 12 MB scene is LDS / L2 resident, so this is NOT a fraction of HBM bandwidth — the kernel is bound by VALU issue, not by memory"},
                "secondary": {"bound": "valu_fp32", "achieved": 30.0 * bytes_launch / 48.0 / (k_ms * 1e-3) / 1e12, "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "frac": 30.0 * bytes_launch / 48.0 / (k_ms * 1e-3) / 1e12 / FP32_VECTOR_PEAK_TFLOPS, "note": "~30 flop per entry visited (SURVEY.md 8d)"},
                "pmc": dict(pmc.get("_meta", {"error": "not collected (--no-pmc or N > 1)"}), counters_per_launch=k),
            },
            "counters": cnt,
            "traced": {"walks_per_frame": walks, "walks_per_s": walks / (ms_per_step * 1e-3) * (1 if not multi else 1), "segments_per_frame": segments,
                       "Gsegments_per_s": segments / (ms_per_step * 1e-3) / 1e9, "scope": "this rank's share of the frame" if multi else "the frame",
                       "note": "`value` counts nominal rays (spp x bounces x pixels); these are the rays the frame really traces (its own work counters: closest-hit walks + shadow walks + primary rays that hit), paths that left the scene or fell below minImportancy trace none"},
        }
        if multi:
            line["gather"] = {"exchange": ("root" if to_root else "all_gather") if rccl else "gloo rehearsal", "uses_rccl": bool(rccl), "rccl_ranks": ctx.comm_count() if rccl else 0,
                              "launched_by": "bench.py itself (child processes)" if os.environ.get("FLX_BENCH_SELF_LAUNCHED") else "torch.distributed.run"}
        if verified is not None:
            line["gathered_frame_equals_single_context_frame"] = verified
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(scene, full, js_workload=(args.workload == "cornell"))      # configs[0]: also this repository's JavaScript restatement (one thread under Node)

    # ---- the batched mode and the frame loops: measured LAST, under a wall-clock limit -------------------------------------------
    # Everything the line must carry is in it now.  What follows runs code paths that no hardware with N > 1 GPUs has executed yet (the frame loop over a split
    # communicator, the shared image over hipIpc): if one of them does not come back within --secondary-timeout seconds, every rank's guard ends its process —
    # rank 0 after printing the line with what it has and the name of the phase that hung — instead of the run hanging until somebody's limit kills it without a line.
    guard = SecondaryGuard(rank, line, args.secondary_timeout)
    if multi:
        dist.barrier()                       # (rank 0 has been timing the CPU oracle: the guards start together)
    guard.start()
    guard.phase("batched")
    batched = None
    if F > 1:
        passes = max(2, (args.steps + F - 1) // F)
        fence()
        t1 = time.perf_counter()
        for _ in range(passes):
            render(batch_params)
        fence()
        dtb = time.perf_counter() - t1
        if multi:
            t = torch.tensor([dtb], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dtb = float(t.item())
        batched = {"frames_per_pass": F, "ms_per_frame": dtb / (passes * F) * 1e3, "passes": passes,
                   "cameras": "a different camera position and view direction for every frame of a batch",
                   "note": "throughput mode (flx_render_batch): every frame complete and bit-identical to its own render, latency %d frames" % F}
    if rank == 0 and batched:
        batched["value"] = rays / (batched["ms_per_frame"] * 1e-3) / 1e6
        batched["unit"] = "Mray/s"
        guard.put("batched", batched)
    guard.phase("gathered_rgba8")
    # N > 1 over RCCL: the same frame-after-frame loop with the frames travelling as the canvas' RGBA8 (every rank quantises its strips before the exchange:
    # a quarter of the bytes; flx_render_gathered_rgba8_device) — what a presenter that shows the frame needs
    rgba8 = None
    if rccl and not use_filter:
        out8 = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
        n8 = max(args.steps, 20)
        for phase in range(2):
            fence()
            t1 = time.perf_counter()
            for _ in range(3 if phase == 0 else n8):
                ctx.render_gathered_rgba8_device([params], 0 if to_root else -1, out8.data_ptr())
            fence()
            dt8 = time.perf_counter() - t1
        if multi:
            t = torch.tensor([dt8], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt8 = float(t.item())
        rgba8 = {"ms_per_frame": dt8 / n8 * 1e3, "frames": n8, "bytes_exchanged_per_frame": tiles.padded_rows(H, args.tile_rows, world) * W * 4 * world,
                 "bytes_exchanged_per_frame_float": tiles.padded_rows(H, args.tile_rows, world) * W * 16 * world,
                 "note": "flx_render_gathered_rgba8_device, frame after frame: trace, quantise the strips (flx_present's bytes), %s of the RGBA8 strips, reassembly" % ("ncclSend / ncclRecv to rank 0" if to_root else "ncclAllGather")}
        if rank == 0 and verified is not None:
            whole8 = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
            wholef = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
            ctx.render_device(full, wholef.data_ptr())
            ctx.present_device(W, H, wholef.data_ptr(), whole8.data_ptr())
            ctx.sync()
            rgba8["equals_present_of_single_context_frame"] = bool(torch.equal(whole8, out8))
    if rank == 0 and rgba8:
        rgba8["value"] = rays / (rgba8["ms_per_frame"] * 1e-3) / 1e6
        rgba8["unit"] = "Mray/s"
        guard.put("gathered_rgba8", rgba8)
    loop_frames = args.steps if args.steps >= 8 else 20      # the frame loops time EXACTLY K frames between the same fences as the timed region (a loop of fewer than 8 frames is ramp-up and drain)
    guard.phase("pipelined")
    # two frames in flight (flx_frame_begin / flx_frame_end with two lanes: what the JavaScript frame loop runs): still one frame per
    # pass and frames complete in order, but frame k + 1's kernels fill the CUs the tails of frame k's kernels leave idle
    pipelined = None
    if not multi or rccl:
        lat = []
        last_ptr = None
        for phase in range(2):               # 0 = warm-up (the second lane sizes its workspace), 1 = timed: EXACTLY K frames (20 when K < 8: then it is no candidate for the headline)
            n = 6 if phase == 0 else loop_frames
            fence()
            t1 = time.perf_counter()
            for i in range(n):
                if rccl:                     # the same loop over the communicator: every lane gathers over its own (flx_frame_begin_gathered)
                    ctx.frame_begin_gathered(params, root=0 if to_root else -1)
                else:
                    ctx.frame_begin(params, device=True)
                if ctx.frames_in_flight() == 2:
                    last_ptr, ms_f = ctx.frame_end()
                    lat.append(ms_f)
            while ctx.frames_in_flight():
                last_ptr, ms_f = ctx.frame_end()
                lat.append(ms_f)
            fence()
            dtp = time.perf_counter() - t1
            if multi:
                t = torch.tensor([dtp], dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dtp = float(t.item())
        pipelined = {"frames_in_flight": 2, "ms_per_frame": dtp / n * 1e3, "frames": n, "frame_gpu_ms_median": float(np.median(lat[-n:])),
                     "note": ("flx_frame_begin_gathered / flx_frame_end on every rank, each of the two lanes gathering over its own communicator" if rccl else "flx_frame_begin / flx_frame_end") +
                             ", pixels left in device memory: one frame per pass, frames complete in order; a frame completes every ms_per_frame, its own GPU time (first kernel .. last, overlapped with its neighbours) is frame_gpu_ms_median"}
    if pipelined:
        # the loop's last frame (still in its slot: nothing was begun since) against one frame per pass of the same frame, bit for bit
        got = None
        if rank == 0 and last_ptr and isinstance(last_ptr, int):
            import ctypes
            try:
                hiprt = ctypes.CDLL("libamdhip64.so")
            except OSError:
                hiprt = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")
            got = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
            torch.cuda.synchronize()
            if hiprt.hipMemcpy(ctypes.c_void_p(got.data_ptr()), ctypes.c_void_p(last_ptr), ctypes.c_size_t(H * W * 16), 3) != 0:
                got = None
        render([params])
        fence()
        if rank == 0:
            pipelined["last_frame_equals_one_frame_per_pass"] = bool(torch.equal(got.view(torch.int32), frames_out[0].view(torch.int32))) if got is not None else None
    if rank == 0 and pipelined:
        pipelined["value"] = rays / (pipelined["ms_per_frame"] * 1e-3) / 1e6
        pipelined["unit"] = "Mray/s"
        guard.put("pipelined", pipelined)
    guard.phase("shared")
    # N > 1, the same loop WITHOUT a collective (flx_share_*): every rank's frame server — one persistent launch that takes the loop's frames as they
    # are posted, three in flight — resolves its strips straight into ONE image in rank 0's device memory (mapped through hipIpc: the other ranks'
    # stores go over xGMI); a page of shared memory carries done / released.  A context of its own, after the timed region; a failure here is
    # reported in the line and never touches `value`.
    shared = None
    if multi and not use_filter and not args.no_shared:
        shared = shared_loop(args, dist, capi, scene, params, full, rank, world, local_rank, loop_frames)

    if rank == 0 and shared:
        if shared.get("ms_per_frame"):
            shared["value"] = rays / (shared["ms_per_frame"] * 1e-3) / 1e6
            shared["unit"] = "Mray/s"
        guard.put("shared", shared)
    guard.done()
    if rank == 0:
        # `value`: whole-job throughput of K frames.  The reference's loop renders frame after frame without waiting for the GPU (pathtracerWGL2.js:254-303), so the
        # K frames of a frame LOOP — each rendered in full, complete in order, timed between the same barrier + synchronize fences, max over ranks — are as much
        # "K steps" as K frames one at a time — but a different quantity (throughput with a latency of several frames): reported beside `value`, never as it.
        cands = []
        if pipelined and pipelined.get("frames") == args.steps and pipelined.get("last_frame_equals_one_frame_per_pass") is True and verified is not False:
            cands.append(("pipelined", pipelined, 2, "flx_frame_begin%s / flx_frame_end, two frames in flight on two lanes" % ("_gathered" if rccl else "")))
        if shared and shared.get("frames") == args.steps and shared.get("error") is None and shared.get("image_equals_single_context_frame") is True and shared.get("ms_per_frame"):
            cands.append(("shared", shared, shared.get("frames_in_flight"), "flx_frame_begin_shared / flx_frame_end_shared: every rank's frame server resolves its strips into one image in rank 0's memory, no exchange"))
        line["one_frame_per_pass"] = {"ms_per_step": line["ms_per_step"], "value": line["value"], "unit": "Mray/s",
                                      "note": "the timed region = `value` / `ms_per_step` (SURVEY.md 8d: K frames one after the other, first kernel launch .. last byte of the gathered frame); every derived "
                                              "figure of the line — frame_gpu_ms, roofline, traced — belongs to this mode"}
        # The frame loops (several frames in flight) are throughput figures with a latency of their own: they never replace `value` (round 4 let the fastest one do so, and
        # `ms_per_step` fell below the dominant kernel's own time per step); the fastest verified one is named here.
        if cands:
            name, c, inflight, how = min(cands, key=lambda x: x[1]["ms_per_frame"])
            line["throughput_best"] = {"mode": name, "value": c["value"], "unit": "Mray/s", "ms_per_frame": c["ms_per_frame"], "frames_in_flight": inflight, "frames": c.get("frames"), "verified": True,
                                       "speedup_over_one_frame_per_pass": line["ms_per_step"] / c["ms_per_frame"],
                                       "note": "the K frames of the `%s` loop (%s): every frame rendered in full and complete in order, timed between barrier + synchronize on both sides, max over "
                                               "ranks, its last frame equal to one context's frame bit for bit.  A throughput figure (latency per frame: that loop's frame_gpu_ms_median), NOT `value`" % (name, how)}
        line["headline"] = {"mode": "one_frame_per_pass", "frames_in_flight": 1}
        if single_ms:
            # both currencies side by side: SURVEY.md 8d's frame time (one frame at a time, gather included) and the frame loop's throughput
            sp = {"single_context_ms": single_ms, "single_context": "the whole frame on rank 0's GPU alone (the other ranks wait at a barrier), one frame per pass, measured in this run",
                  "latency_mode": {"ms_per_frame": line["ms_per_step"], "speedup": single_ms / line["ms_per_step"], "frames_in_flight": 1, "frames_of_latency": 1,
                                   "what": "the timed region = `value`: every frame complete and gathered before the next begins (SURVEY.md 8d)"}}
            if line.get("throughput_best"):
                tb = line["throughput_best"]
                src = shared if tb["mode"] == "shared" else pipelined
                sp["throughput_mode"] = {"mode": tb["mode"], "ms_per_frame": tb["ms_per_frame"], "speedup": single_ms / tb["ms_per_frame"], "frames_in_flight": tb["frames_in_flight"],
                                         "frames_of_latency": tb["frames_in_flight"], "frame_latency_ms": (src or {}).get("frame_gpu_ms_median") or (src or {}).get("latency_ms_median"),
                                         "what": "a frame completes every ms_per_frame while `frames_in_flight` are in flight: the application sees each frame that many frames after it began it"}
            sp["note"] = "two different quantities: do not read the throughput figure as SURVEY.md 8d's frame time; the driver computes scaling from `value` (latency mode) of the per-N runs"
            line["speedup"] = sp
        print(json.dumps(line), flush=True)
    if rccl:
        ctx.sync()
        ctx.comm_destroy()
    ctx.close()
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and verified is False:
        sys.stderr.write("bench.py: the gathered frame differs from the single-context frame\n")
        raise SystemExit(3)
    if rank == 0 and shared and shared.get("image_equals_single_context_frame") is False:
        sys.stderr.write("bench.py: the image the ranks completed without a collective differs from the single-context frame\n")
        raise SystemExit(3)


if __name__ == "__main__":
    main()
