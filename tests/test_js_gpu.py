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


def test_frame_loop_with_a_moving_scene_through_node(hip, oracle, scenes, tmp_path):
    """SURVEY 8f N3 / examples/dragon.js:97-110: the application moves the camera every tick and the monkey turns to face it
    (Transform.rotateSpherical), so the transform arrays change every frame.  Three consecutive frames of the renderer's own
    frame loop — engine.renderer.render(): two frames in flight, lights and transforms re-uploaded per frame in stream order,
    pixels handed out of pinned memory — each equal to the oracle's frame for the camera and transforms of that tick; and the
    transform arrays the JavaScript host derives equal the native flx_transforms_pack of the same matrices."""
    import copy
    from flexlight_hip import capi
    node = shutil.which("node")
    w, h, spp, bounces = 320, 180, 2, 3
    prefix = str(tmp_path / "loop")
    info = json.loads(subprocess.check_output(
        [node, os.path.join(ROOT, "tools", "js_loop.js"), os.path.join(ROOT, "tests", "golden", "ref_dragon.flxs.gz"), "--frames", "4", "--move", "1",
         "--width", str(w), "--height", str(h), "--spp", str(spp), "--bounces", str(bounces), "--dump", prefix, "--dump-frames", "3"], timeout=300).decode().splitlines()[-1])
    assert info["frames"] == 4 and info["moving"] is True and info["gpuMsMedian"] > 0
    log = json.load(open(prefix + "log.json"))
    assert len(log) == 3
    sc = scenes("dragon")
    frames = []
    for k, tick in enumerate(log):
        rot, shift = np.asarray(tick["rotation"], np.float32), np.asarray(tick["shift"], np.float32)
        # the same arrays from the native packer: identity, the dragon's 0.5 x identity at (15, 0, 15), the monkey's 2 x R at (5, 1, 12)
        cam = np.asarray(tick["camera"], np.float64)
        d = cam - np.array([5.0, 1.0, 12.0])
        theta = np.sign(d[2]) * np.arccos(d[0] / np.sqrt(d[0] * d[0] + d[2] * d[2])) - np.pi * 0.5
        psi = np.arccos(d[1] / np.sqrt((d * d).sum())) - np.pi * 0.5
        sT, cT, sP, cP = np.sin(theta), np.cos(theta), np.sin(psi), np.cos(psi)
        R = np.array([[cT, 0, sT], [-sT * sP, cP, cT * sP], [-sT * cP, -sP, cT * cP]])
        nrot, nshift = capi.transforms_pack([np.eye(3), 0.5 * np.eye(3), 2.0 * R], [[0, 0, 0], [15, 0, 15], [5, 1, 12]])
        assert np.abs(nrot.reshape(-1) - rot).max() <= 1e-6 and np.array_equal(nshift.reshape(-1), shift)      # (Math.sin vs libm may differ in the last place)
        sck = copy.copy(sc)
        sck.arrays = dict(sc.arrays, rotation=rot, shift=shift)
        p = sc.frame_params(width=w, height=h, samples=spp, max_reflections=bounces, use_filter=0)
        p.camera[:] = tick["camera"]
        p.view_matrix[:] = tick["viewMatrix"]
        want, _, _ = oracle.render(sck, p)
        got = np.fromfile(prefix + "%d.f32" % k, np.float32).reshape(h, w, 4)
        _, mism = assert_parity(got, want, "loop frame %d" % k)
        assert mism == 0, "frame %d" % k
        frames.append(got)
    assert not np.array_equal(frames[0], frames[1]) and not np.array_equal(frames[1], frames[2])


def test_group_of_two_contexts_through_node(hip, scenes, tmp_path):
    """new FlexLight(canvas, { devices: [0, 0] }): the frame split over two contexts and gathered inside the library equals the
    frame of one context, through the whole JavaScript path"""
    node = shutil.which("node")
    w, h, spp, bounces = 96, 64, 2, 3
    base = [node, os.path.join(ROOT, "tools", "render_scene.js"), "cornell", "--width", str(w), "--height", str(h), "--spp", str(spp),
            "--bounces", str(bounces), "--assets", "/nonexistent"]
    for filt in ("0", "1"):
        one, two = tmp_path / ("one%s.f32" % filt), tmp_path / ("two%s.f32" % filt)
        subprocess.check_output(base + ["--out", str(one), "--filter", filt], timeout=300)
        info = json.loads(subprocess.check_output(base + ["--out", str(two), "--filter", filt, "--devices", "0,0"], timeout=300).decode().splitlines()[-1])
        assert info["gpus"] == {"size": 2, "rccl": False}
        assert np.array_equal(np.fromfile(one, np.uint32), np.fromfile(two, np.uint32))


@pytest.mark.parametrize("move_scene", [False, True])
def test_group_frame_loop_through_node(hip, oracle, scenes, tmp_path, move_scene):
    """new FlexLight(canvas, { devices: [0, 0] }).renderer.render(): the renderer's frame loop over a GROUP of contexts — flx_group_frame_begin / _end, every
    context's frame server resolving its strips straight into one pinned image, three frames in flight, nothing waits for a GPU inside a frame (round-3 review,
    item 5; pathtracerWGL2.js:191, 254-303).  Frames of a moving camera equal the oracle's frames for their ticks, with the scene's transforms static (the
    servers' launches live on across the frames: an upload of unchanged lights / transforms is nothing) and with the monkey turning every tick (from the
    first changed upload on the launches take the transforms with every frame: flx_server.hip, VER)."""
    import copy
    node = shutil.which("node")
    w, h, spp, bounces = 320, 176, 2, 3            # (176 rows: strips of whole 8 x 8 tiles for both contexts — frames the server takes)
    prefix = str(tmp_path / "gloop")
    cmd = [node, os.path.join(ROOT, "tools", "js_loop.js"), os.path.join(ROOT, "tests", "golden", "ref_dragon.flxs.gz"), "--frames", "6", "--move", "1" if move_scene else "2",
           "--width", str(w), "--height", str(h), "--spp", str(spp), "--bounces", str(bounces), "--dump", prefix, "--dump-frames", "5", "--devices", "0,0"]
    info = json.loads(subprocess.check_output(cmd, timeout=300).decode().splitlines()[-1])
    assert info["frames"] == 6 and info["devices"] == [0, 0] and info["lanes"] == 3 and info["gpuMsMedian"] > 0
    log = json.load(open(prefix + "log.json"))
    assert len(log) == 5
    sc = scenes("dragon")
    frames = []
    for k, tick in enumerate(log):
        sck = copy.copy(sc)
        sck.arrays = dict(sc.arrays, rotation=np.asarray(tick["rotation"], np.float32), shift=np.asarray(tick["shift"], np.float32))
        p = sc.frame_params(width=w, height=h, samples=spp, max_reflections=bounces, use_filter=0)
        p.camera[:] = tick["camera"]
        p.view_matrix[:] = tick["viewMatrix"]
        want, _, _ = oracle.render(sck, p)
        got = np.fromfile(prefix + "%d.f32" % k, np.float32).reshape(h, w, 4)
        _, mism = assert_parity(got, want, "group loop frame %d" % k)
        assert mism == 0, "frame %d" % k
        frames.append(got)
    assert not np.array_equal(frames[0], frames[1]) and not np.array_equal(frames[3], frames[4])
    if not move_scene:
        assert log[0]["rotation"] == log[4]["rotation"]


def test_group_presents_rgba8_through_node(hip, oracle, scenes, tmp_path):
    """a group whose renderer presents the canvas' RGBA8 (renderer.present8 with devices): the group's frame loop with FLX_FRAME_RGBA8 — every context's frame server
    quantises the tiles it resolves into the pinned image, a quarter of the bytes — hands the loop the bytes flx_present / the oracle store for the float frame of
    the same tick"""
    node = shutil.which("node")
    w, h, spp, bounces = 320, 180, 2, 3
    prefix = str(tmp_path / "g8")
    cmd = [node, os.path.join(ROOT, "tools", "js_loop.js"), os.path.join(ROOT, "tests", "golden", "ref_dragon.flxs.gz"), "--frames", "3", "--move", "2", "--present8", "1",
           "--width", str(w), "--height", str(h), "--spp", str(spp), "--bounces", str(bounces), "--dump", prefix, "--dump-frames", "2", "--devices", "0,0,0"]
    info = json.loads(subprocess.check_output(cmd, timeout=300).decode().splitlines()[-1])
    assert info["present8"] is True and info["devices"] == [0, 0, 0]
    log = json.load(open(prefix + "log.json"))
    sc = scenes("dragon")
    for k, tick in enumerate(log):
        p = sc.frame_params(width=w, height=h, samples=spp, max_reflections=bounces, use_filter=0)
        p.camera[:] = tick["camera"]
        p.view_matrix[:] = tick["viewMatrix"]
        want = oracle.present(oracle.render(sc, p)[0])
        got = np.fromfile(prefix + "%d.f32" % k, np.uint8).reshape(h, w, 4)
        assert np.array_equal(got, want), "frame %d" % k


def test_the_frame_loop_leaves_the_main_thread_in_the_event_loop(tmp_path):
    """SURVEY 8b ("a napi_async_work worker for the non-blocking variant") / pathtracerWGL2.js:300-302 (the reference's loop returns to the event loop every frame):
    PathTracerHIP.render() waits for a frame on a worker thread (frameEndAsync), so an application timer of 1 ms fires on time while whole 1080p frames (6.4 ms each) are
    traced; with the blocking flx_frame_end of round 4 the same timer is late by most of a frame.  Frames are the same bytes either way."""
    node = shutil.which("node")
    assert node
    out = {}
    for mode in ("0", "1"):
        prefix = str(tmp_path / ("lag%s_" % mode))
        cmd = [node, os.path.join(ROOT, "tools", "js_loop.js"), os.path.join(ROOT, "tests", "golden", "ref_dragon.flxs.gz"), "--frames", "40", "--move", "1",
               "--blocking", mode, "--dump", prefix, "--dump-frames", "3"]
        out[mode] = json.loads(subprocess.check_output(cmd, timeout=600).decode().splitlines()[-1])
    a, b = out["0"], out["1"]
    assert a["frameEnd"].startswith("async") and b["frameEnd"] == "blocking"
    assert a["eventLoopLagMs"]["samples"] > 40 and a["eventLoopLagMs"]["median"] < 1.0, a["eventLoopLagMs"]
    assert b["eventLoopLagMs"]["median"] > 2.0 * max(a["eventLoopLagMs"]["median"], 0.25), (a["eventLoopLagMs"], b["eventLoopLagMs"])      # the blocking loop holds the thread for the frame
    assert a["fps"] > 0.9 * b["fps"], (a["fps"], b["fps"])                                         # ... and the worker thread costs the loop next to nothing
    for k in range(3):
        fa = np.fromfile(str(tmp_path / ("lag0_%d.f32" % k)), np.float32)
        fb = np.fromfile(str(tmp_path / ("lag1_%d.f32" % k)), np.float32)
        assert fa.size == 1920 * 1080 * 4 and np.array_equal(fa.view(np.uint32), fb.view(np.uint32)), k
