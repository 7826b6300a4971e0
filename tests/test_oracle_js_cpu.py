"""oracle/js/flx_oracle_js.js — the oracle's per-pixel function in plain JavaScript (bench.py's `cpu_baseline.js`: this repository's restatement under Node, one thread) —
against the C oracle, bit for bit, on BASELINE configs[0] at its full size and on small frames of the other scenes (several samples and bounces, textures, nine lights).
CPU only; test infrastructure checking test infrastructure."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JS = os.path.join(ROOT, "oracle", "js", "flx_oracle_js.js")


def params_json(p):
    return {"width": int(p.width), "height": int(p.height), "camera": [float(v) for v in p.camera], "view_matrix": [float(v) for v in p.view_matrix],
            "samples": int(p.samples), "max_reflections": int(p.max_reflections), "min_importancy": float(p.min_importancy), "ambient": [float(v) for v in p.ambient],
            "random_seed": float(p.random_seed), "texture_width": int(p.texture_width), "use_filter": 0, "is_temporal": 0}


@pytest.mark.parametrize("name,kw", [
    ("cornell", dict(width=256, height=256, samples=1, max_reflections=1)),              # BASELINE configs[0]
    ("cornell", dict(width=96, height=96, samples=3, max_reflections=4)),
    ("cornell_obj", dict(width=96, height=54, samples=2, max_reflections=3)),
    ("theater", dict(width=64, height=36, samples=2, max_reflections=3)),
])
def test_js_restatement_equals_the_c_oracle(oracle, scenes, tmp_path, name, kw):
    node = shutil.which("node")
    if not node:
        pytest.skip("no node in this environment")
    sc = scenes(name)
    p = sc.frame_params(use_filter=0, **kw)
    p.random_seed = 3.0
    want, want_cnt = oracle.render(sc, p)[:2]
    pj = tmp_path / "params.json"
    pj.write_text(json.dumps(params_json(p)))
    out = tmp_path / "frame.f32"
    info = json.loads(subprocess.check_output([node, JS, os.path.join(ROOT, "tests", "golden", "ref_%s.flxs.gz" % name), str(pj), "--out", str(out)], timeout=600).decode().splitlines()[-1])
    got = np.fromfile(str(out), np.float32).reshape(p.height, p.width, 4)
    bad = np.flatnonzero((got.view(np.uint32) != want.view(np.uint32)).any(axis=-1).reshape(-1))
    assert bad.size == 0, "%d pixels differ, first at %s: js %s oracle %s" % (bad.size, divmod(int(bad[0]), p.width), got.reshape(-1, 4)[bad[0]], want.reshape(-1, 4)[bad[0]])
    assert {k: int(v) for k, v in info["counters"].items()} == {k: int(v) for k, v in want_cnt.items()}
    assert info["ms_per_frame"] > 0
