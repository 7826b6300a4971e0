"""End to end through the JavaScript path on the GPU: FlexLight facade -> scene graph -> flattening ->
N-API addon -> libflexlight_hip.so, compared with the same frame through the ctypes binding of the
fixture arrays and with the CPU oracle.  The cornell scene is built purely through the API (no asset
files), so this runs on the GPU box."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

from parity_util import assert_parity

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("filt", [0, 1])
def test_cornell_through_node_matches_capi_and_oracle(hip, oracle, scenes, tmp_path, filt):
    node = shutil.which("node")
    addon = os.path.join(ROOT, "web-ray-tracer_amd", "napi", "flexlight_napi.node")
    assert node, "node is part of the image"
    assert os.path.exists(addon), "N-API addon not built (run __graft_entry__.build())"
    out = tmp_path / "frame.f32"
    w, h, spp, bounces = 96, 64, 2, 3
    info = json.loads(subprocess.check_output(
        [node, os.path.join(ROOT, "tools", "render_scene.js"), "cornell", "--out", str(out), "--width", str(w), "--height", str(h),
         "--spp", str(spp), "--bounces", str(bounces), "--filter", str(filt), "--assets", "/nonexistent"], timeout=300).decode().splitlines()[-1])
    got = np.fromfile(out, np.float32).reshape(h, w, 4)
    sc = scenes("cornell")
    p = sc.frame_params(width=w, height=h, samples=spp, max_reflections=bounces, use_filter=filt)
    hip.update_scene(sc)
    via_capi, cnt, _ = hip.render(p, counters=True)
    want, want_cnt, _ = oracle.render(sc, p)
    assert np.array_equal(got, via_capi, equal_nan=True)
    rms, mism = assert_parity(got, want, "cornell via node")
    assert mism == 0
    js_cnt = info["counters"]
    assert js_cnt["shades"] == want_cnt["shades"] and js_cnt["closestVisits"] == want_cnt["closest_visits"]
    assert js_cnt["primaryHits"] == want_cnt["primary_hits"]


def test_antialiasing_through_node(hip, oracle, scenes, tmp_path):
    """config.antialiasing = 'fxaa' in the JavaScript renderer = the FXAA pass over the frame it renders;
    'taa' jitters the camera, so only its plumbing is checked here (the pass itself: tests/test_postfx_gpu.py)."""
    node = shutil.which("node")
    w, h, spp, bounces = 96, 64, 2, 3
    base = [node, os.path.join(ROOT, "tools", "render_scene.js"), "cornell", "--width", str(w), "--height", str(h), "--spp", str(spp),
            "--bounces", str(bounces), "--filter", "0", "--assets", "/nonexistent"]
    out = tmp_path / "fxaa.f32"
    subprocess.check_output(base + ["--out", str(out), "--aa", "fxaa"], timeout=300)
    got = np.fromfile(out, np.float32).reshape(h, w, 4)
    sc = scenes("cornell")
    hip.update_scene(sc)
    plain, _, _ = hip.render(sc.frame_params(width=w, height=h, samples=spp, max_reflections=bounces, use_filter=0))
    assert np.array_equal(got.view(np.uint32), oracle.fxaa(plain).view(np.uint32))
    out = tmp_path / "taa.f32"
    subprocess.check_output(base + ["--out", str(out), "--aa", "taa", "--frames", "3"], timeout=300)
    taa = np.fromfile(out, np.float32).reshape(h, w, 4)
    assert np.isfinite(taa).all() and taa.max() > 0 and not np.array_equal(taa, plain)


def test_render_batch_through_node(hip, scenes, tmp_path):
    """renderer.renderBatch(cameras): four frames of a camera move in one pass, each equal to the renderFrame() of its camera
    (compared inside Node, bit for bit) and — frame 0, the fixture's camera — to the ctypes binding's frame."""
    node = shutil.which("node")
    w, h, spp, bounces = 96, 64, 2, 3
    out = tmp_path / "batch.f32"
    info = json.loads(subprocess.check_output(
        [node, os.path.join(ROOT, "tools", "render_scene.js"), "cornell", "--out", str(out), "--width", str(w), "--height", str(h), "--spp", str(spp),
         "--bounces", str(bounces), "--filter", "0", "--assets", "/nonexistent", "--batch", "4"], timeout=300).decode().splitlines()[-1])
    assert info["frames"] == 4 and info["batchEqualsFrames"] is True
    got = np.fromfile(out, np.float32).reshape(4, h, w, 4)
    sc = scenes("cornell")
    hip.update_scene(sc)
    first, cnt, _ = hip.render(sc.frame_params(width=w, height=h, samples=spp, max_reflections=bounces, use_filter=0), counters=True)
    assert np.array_equal(got[0], first, equal_nan=True)
    assert not np.array_equal(got[0], got[1])
    assert info["counters"]["primaryHits"] >= cnt["primary_hits"]


def test_present_through_node(oracle, tmp_path):
    """renderer.presentFrame(): the RGBA8 of the canvas for the frame renderFrame() returned"""
    node = shutil.which("node")
    out, pix = tmp_path / "frame.f32", tmp_path / "frame.u8"
    subprocess.check_output([node, os.path.join(ROOT, "tools", "render_scene.js"), "cornell", "--out", str(out), "--present", str(pix), "--width", "96",
                             "--height", "64", "--spp", "2", "--bounces", "3", "--filter", "1", "--assets", "/nonexistent"], timeout=300)
    frame = np.fromfile(out, np.float32).reshape(64, 96, 4)
    got = np.fromfile(pix, np.uint8).reshape(64, 96, 4)
    assert np.array_equal(got, oracle.present(frame))
    assert got[..., :3].max() > 0
