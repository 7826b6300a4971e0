"""bench.py end to end on the GPU box: the single-GPU line, and the multi-rank path rehearsed with two ranks on the one GPU
(gloo instead of RCCL — two ranks cannot share a device under RCCL), with --verify: the frame the ranks gather equals the
frame one context renders, bit for bit, with frames batched per pass."""
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
    d = _last_json(subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", "4"] + SMALL, timeout=600, stderr=subprocess.DEVNULL))
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["unit"] == "Mray/s" and d["value"] > 0
    assert d["config"]["frames_per_pass"] == 4 and d["config"]["width"] == 480
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["achieved"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert d["frame_after_frame"]["ms_per_frame"] > 0
    assert d["counters"]["primary_hits"] > 0


@pytest.mark.parametrize("batch", [4, 1])
def test_bench_two_ranks_gather_the_frame(batch):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29700 + batch), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--one-device",
           "--verify", "--batch", str(batch)] + SMALL
    d = _last_json(subprocess.check_output(cmd, timeout=600, env=env, stderr=subprocess.DEVNULL))
    assert d["n_gpus"] == 2 and d["config"]["frames_per_pass"] == batch
    assert d["gathered_frame_equals_single_context_frame"] is True
    assert "row-strip tiles x2" in d["config"]["parallelism"]
