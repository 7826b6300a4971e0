"""bench.py end to end on the GPU box: the single-GPU line (with the counter passes it makes itself under rocprofv3), and the
multi-rank path rehearsed with two ranks on the one GPU (the strips gathered over gloo — RCCL refuses two ranks on one device;
the RCCL path of the library is covered by tests/test_group_gpu.py) with --verify: the frame the ranks gather equals the frame
one context renders, bit for bit."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--width", "480", "--height", "270", "--steps", "6", "--warmup", "2", "--no-cpu-baseline"]


def _last_json(out):
    return json.loads([l for l in out.decode().splitlines() if l.startswith("{")][-1])


def test_bench_line_single_gpu():
    d = _last_json(subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", "4"] + SMALL, timeout=900, stderr=subprocess.DEVNULL))
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["unit"] == "Mray/s" and d["value"] > 0
    assert d["config"]["frames_per_pass"] == 1 and d["config"]["width"] == 480          # the headline is one frame per pass
    assert abs(d["value"] - d["config"]["rays_per_frame"] / (d["ms_per_step"] * 1e-3) / 1e6) < 1e-6 * d["value"]
    assert d["batched"]["frames_per_pass"] == 4 and d["batched"]["ms_per_frame"] > 0
    assert d["frame_gpu_ms"]["frames"] >= 20 and d["frame_gpu_ms"]["median"] > 0
    r = d["roofline"]
    assert r["bound"] == "valu_issue" and r["kernel_ms"] > 0 and r["algorithmic"]["bytes_per_launch"] > 0
    # the counter passes of this very run: a physical fraction, at most 1 (where rocprofv3 cannot run the line says why and carries nulls)
    if "error" in r["pmc"]:
        assert r["achieved"] is None and r["frac"] is None and r["traffic"] is None, r["pmc"]
    else:
        assert r["achieved"] > 0 and 0 < r["frac"] <= 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
        assert r["traffic"] > 0 and 0 < r["hbm_measured"]["frac"] <= 1
        ic = r["issue_cycles"]                  # the same instructions priced by what their class costs to issue (profiles/r04_valu_rates.txt): never below the all-FMA price
        if ic is not None:
            assert r["frac"] <= ic["frac_low"] * 1.001 and ic["frac_low"] <= ic["frac"] <= ic["frac_high"] <= 1.05
            assert abs(sum(ic["instructions_by_class"].values()) - r["valu_insts_per_launch"]) <= 1e-6 * r["valu_insts_per_launch"]
    assert d["pipelined"]["frames_in_flight"] == 2 and d["pipelined"]["ms_per_frame"] > 0
    assert d["pipelined"]["last_frame_equals_one_frame_per_pass"] is True
    # K = 6 < 8: the loops time 20 frames, not K — no candidate for the headline
    assert d["headline"]["mode"] == "one_frame_per_pass" and d["one_frame_per_pass"]["ms_per_step"] == d["ms_per_step"] and d["pipelined"]["frames"] == 20
    assert d["counters"]["primary_hits"] > 0


def test_bench_value_is_one_frame_per_pass_and_the_loops_stand_beside_it():
    """`value` / `ms_per_step` are ALWAYS the timed region (one frame per pass); the frame loops are named entries with their own `value`, the fastest verified one
    repeated as `throughput_best`; and `ms_per_step` is not below the dominant kernel's own time per step"""
    args = [a if a != "6" else "12" for a in SMALL]
    d = _last_json(subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", "0", "--no-pmc"] + args, timeout=900, stderr=subprocess.DEVNULL))
    one, pl = d["one_frame_per_pass"], d["pipelined"]
    assert d["steps"] == 12 and pl["frames"] == 12 and pl["last_frame_equals_one_frame_per_pass"] is True
    assert d["headline"]["mode"] == "one_frame_per_pass" and d["ms_per_step"] == one["ms_per_step"] and d["value"] == one["value"]
    assert abs(d["value"] - d["config"]["rays_per_frame"] / (d["ms_per_step"] * 1e-3) / 1e6) < 1e-6 * d["value"]
    tb = d["throughput_best"]
    assert tb["mode"] == "pipelined" and tb["frames_in_flight"] == 2 and tb["frames"] == 12 and tb["value"] == pl["value"] and tb["ms_per_frame"] == pl["ms_per_frame"]
    r = d["roofline"]
    assert d["ms_per_step"] >= r["kernel_ms"] * r.get("launches_per_step", 1) * 0.999, (d["ms_per_step"], r["kernel_ms"])      # the dominant kernel's time per step fits in the step


def test_bench_two_ranks_gather_the_frame():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29701", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--one-device", "--verify", "--batch", "4"] + SMALL
    d = _last_json(subprocess.check_output(cmd, timeout=900, env=env, stderr=subprocess.DEVNULL))
    assert d["n_gpus"] == 2 and d["batched"]["frames_per_pass"] == 4
    assert d["gathered_frame_equals_single_context_frame"] is True
    assert "row-strip tiles x2" in d["config"]["parallelism"]


def test_bench_two_ranks_complete_one_image_without_a_collective():
    """the `shared` entry of the N > 1 line: the two ranks' frame servers (half of the CUs each on this one GPU) resolve their strips into ONE image in rank 0's
    memory (flx_share_*: hipIpc mapping + a page of shared memory), three frames in flight; the image equals one context's render of the whole frame"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    args = [a if a != "270" else "272" for a in SMALL]          # strips of whole 8-row tiles: a frame the frame server takes
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--one-device", "--batch", "0", "--no-pmc"] + args, timeout=1200, env=env, stderr=subprocess.DEVNULL)
    d = _last_json(out)
    sh = d["shared"]
    assert sh.get("error") is None, sh
    assert sh["frames_in_flight"] == 3 and sh["ms_per_frame"] > 0 and sh["value"] > 0
    assert sh["image_equals_single_context_frame"] is True
    assert d["gathered_frame_equals_single_context_frame"] is True
    assert sh["frame_server"] is True
    assert sh["frames"] == 20 and d["headline"]["mode"] == "one_frame_per_pass"      # (K = 6: the loop timed 20 frames)
    # K = 12: the shared loop times exactly K frames; it is a named entry and `throughput_best`, never `value`
    args12 = [a if a != "6" else "12" for a in args]
    d = _last_json(subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--one-device", "--batch", "0", "--no-pmc"] + args12, timeout=1200, env=env, stderr=subprocess.DEVNULL))
    assert d["shared"]["frames"] == 12 and d["shared"]["image_equals_single_context_frame"] is True
    assert d["headline"]["mode"] == "one_frame_per_pass" and d["ms_per_step"] == d["one_frame_per_pass"]["ms_per_step"]
    assert d["throughput_best"]["mode"] in ("shared", "pipelined") and d["throughput_best"]["ms_per_frame"] <= d["shared"]["ms_per_frame"]
    # the speed-up in both currencies, against the same frame on one context measured in this run
    sp = d["speedup"]
    assert sp["single_context_ms"] > 0 and sp["latency_mode"]["frames_in_flight"] == 1 and abs(sp["latency_mode"]["speedup"] - sp["single_context_ms"] / d["ms_per_step"]) < 1e-9
    assert sp["throughput_mode"]["mode"] == d["throughput_best"]["mode"] and sp["throughput_mode"]["frames_in_flight"] >= 2
    assert abs(sp["throughput_mode"]["speedup"] - sp["single_context_ms"] / d["throughput_best"]["ms_per_frame"]) < 1e-9
    # a frame the server does not take (a last strip of 6 rows): every rank's two lanes, the strips copied into the image
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--one-device", "--batch", "0", "--no-pmc"] + SMALL, timeout=1200, env=env, stderr=subprocess.DEVNULL)
    sh = _last_json(out)["shared"]
    assert sh.get("error") is None and sh["frame_server"] is False and sh["image_equals_single_context_frame"] is True, sh


def test_bench_five_ranks_on_one_device():
    """the N-rank code path with as many ranks as a one-GPU box lets near its card beside the test runner (the box allows six processes on the GPU; the eight-rank case is the
    driver's to start, on eight GPUs): the share page with five ranks' done words, the tile policy of five, five servers beside each other on a fifth of the CUs each — the image
    they complete equals one context's frame bit for bit, and the line states the speed-up in both currencies"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    args = ["--width", "480", "--height", "280", "--steps", "12", "--warmup", "2", "--no-cpu-baseline", "--gpus", "5", "--one-device", "--batch", "0", "--no-pmc"]      # 280 rows = 5 ranks x 7 strips of 8
    d = _last_json(subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py")] + args, timeout=1200, env=env, stderr=subprocess.DEVNULL))
    assert d["n_gpus"] == 5 and d["gathered_frame_equals_single_context_frame"] is True
    sh = d["shared"]
    assert sh.get("error") is None and sh["frame_server"] is True and sh["image_equals_single_context_frame"] is True and sh["frames"] == 12, sh
    assert d["headline"]["mode"] == "one_frame_per_pass" and d["speedup"]["latency_mode"]["speedup"] > 0 and d["speedup"]["throughput_mode"]["speedup"] > 0


def test_bench_line_survives_a_secondary_measurement_that_hangs():
    """the measurements after the timed region run under a wall-clock limit: when one does not come back the line is printed as it stands, names the phase, and the run ends"""
    env = dict(os.environ, FLX_BENCH_TEST_HANG="pipelined")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", "4", "--no-pmc", "--secondary-timeout", "5"] + SMALL, timeout=600, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    assert p.returncode == 0
    d = _last_json(p.stdout)
    assert d["secondary_incomplete"] == {"phase": "pipelined", "limit_s": 5}
    assert d["value"] > 0 and d["batched"]["frames_per_pass"] == 4 and "pipelined" not in d and d["roofline"]["kernel_ms"] > 0


def test_bench_rccl_path_with_one_rank():
    """the code path `bench.py --gpus N` takes for N > 1 — gloo bootstrap, communicator id, ncclCommInitRank, frames through
    flx_render_gathered_device (trace, ncclAllGather, reassembly in the library) in the timed loop — with a communicator of one rank"""
    env = dict(os.environ, FLX_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29703")
    d = _last_json(subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", "4", "--verify"] + SMALL, timeout=900, env=env, stderr=subprocess.DEVNULL))
    assert d["n_gpus"] == 1 and d["value"] > 0 and "ncclSend / ncclRecv to rank 0" in d["config"]["parallelism"]
    assert d["gathered_frame_equals_single_context_frame"] is True
    assert d["batched"]["frames_per_pass"] == 4
    assert d["gather"] == {"exchange": "root", "uses_rccl": True, "rccl_ranks": 1, "launched_by": "torch.distributed.run"} or d["gather"]["rccl_ranks"] == 1
    assert d["pipelined"]["frames_in_flight"] == 2 and "communicator" in d["pipelined"]["note"]      # the frame loop over the communicator
    g8 = d["gathered_rgba8"]                                                                        # the frames travelling as the canvas' RGBA8
    assert g8["ms_per_frame"] > 0 and g8["equals_present_of_single_context_frame"] is True and g8["bytes_exchanged_per_frame"] * 4 == g8["bytes_exchanged_per_frame_float"]
    assert d["shared"]["frame_server"] is False and d["shared"]["image_equals_single_context_frame"] is True      # (270 rows: a last strip of 6 the frame server does not take: the lanes)
    e = _last_json(subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", "0", "--gather", "all", "--no-pmc"] + SMALL, timeout=900, env=env, stderr=subprocess.DEVNULL))
    assert "ncclAllGather" in e["config"]["parallelism"] and e["gathered_frame_equals_single_context_frame"] is True and e["gather"]["exchange"] == "all_gather"


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with nothing around it: bench.py starts the two ranks itself (child processes, before it touches a
    GPU), collects rank 0's counter passes for its strips, verifies the gathered frame by default and says how the frame was
    exchanged — the N = 1 line's schema plus `gather` and the verification flag"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--one-device", "--batch", "4"] + SMALL, timeout=1200, env=env, stderr=subprocess.DEVNULL)
    d = _last_json(out)
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["value"] > 0
    assert d["gathered_frame_equals_single_context_frame"] is True            # verified without --verify
    assert d["gather"]["launched_by"].startswith("bench.py itself") and d["gather"]["uses_rccl"] is False
    one = _last_json(subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", "4", "--no-pmc"] + SMALL, timeout=900, env=env, stderr=subprocess.DEVNULL))
    assert set(one) - {"pipelined"} <= set(d), sorted(set(one) - set(d))    # the N = 1 schema is a subset of the N > 1 line's (`pipelined` needs RCCL lanes)
    assert set(one["roofline"]) == set(d["roofline"])
    r = d["roofline"]
    if "error" not in r["pmc"]:                                               # rank 0's strips under rocprofv3: the line has a roofline for N > 1 too
        assert r["achieved"] > 0 and 0 < r["frac"] <= 1 and "tile" in r["pmc"]["collected"]
