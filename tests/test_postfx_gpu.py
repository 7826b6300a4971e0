"""the anti-aliasing post passes (modules/fxaa.js, modules/taa.js) on the GPU against the CPU oracle — bit for bit,
like every other pass: both read the RGBA8 texture the frame was drawn into and do float arithmetic on 8-bit texels."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _frames(hip, sc, w, h, n, spp=1, bounces=2):
    """n slightly different frames of a scene (the camera turns a little, as the TAA jitter does)"""
    hip.update_scene(sc)
    out = []
    for k in range(n):
        p = sc.frame_params(width=w, height=h, samples=spp, max_reflections=bounces, use_filter=0)
        p.view_matrix[2] += 0.002 * k
        p.view_matrix[5] -= 0.001 * k
        out.append(hip.render(p)[0])
    return out


@pytest.mark.parametrize("name,w,h", [("cornell", 160, 120), ("dragon", 240, 136), ("theater", 96, 64)])
def test_fxaa_matches_oracle(hip, oracle, scenes, name, w, h):
    frame = _frames(hip, scenes(name), w, h, 1, spp=2, bounces=3)[0]
    got = hip.fxaa(frame)
    want = oracle.fxaa(frame)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert not np.array_equal(got, frame)                     # edges were found and blended
    # a synthetic frame with hard edges, values outside [0, 1], NaN and a transparent background
    rng = np.random.default_rng(3)
    syn = rng.uniform(-0.2, 1.3, (67, 93, 4)).astype(np.float32)
    syn[20:40, 30:60] = [1.0, 0.5, 0.25, 1.0]
    syn[:, :10, 3] = 0.0
    syn[5, 5] = np.nan
    assert np.array_equal(hip.fxaa(syn).view(np.uint32), oracle.fxaa(syn).view(np.uint32))


def test_taa_sequence_matches_oracle(hip, oracle, scenes):
    frames = _frames(hip, scenes("cornell"), 128, 96, 12)
    hip.taa_reset()
    for k, f in enumerate(frames):
        got = hip.taa(f)
        want = oracle.taa(frames[k::-1][:9])                   # newest first; before nine frames the missing ones are zero textures
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), k
    # another size starts a fresh ring
    small = _frames(hip, scenes("cornell"), 64, 48, 2)
    assert np.array_equal(hip.taa(small[0]).view(np.uint32), oracle.taa(small[:1]).view(np.uint32))
    assert np.array_equal(hip.taa(small[1]).view(np.uint32), oracle.taa(small[1::-1]).view(np.uint32))
    hip.taa_reset()
    assert np.array_equal(hip.taa(small[1]).view(np.uint32), oracle.taa(small[1:2]).view(np.uint32))


def test_present_matches_oracle(hip, oracle, scenes):
    """8-bit present: the RGBA8 of the canvas for a rendered frame and for a synthetic one with NaN, infinities, negative and
    > 1 values and the rounding midpoints k / 255 +- half a step."""
    sc = scenes("cornell")
    hip.update_scene(sc)
    frame, _, _ = hip.render(sc.frame_params(width=96, height=64, samples=2, max_reflections=3, use_filter=0))
    rng = np.random.default_rng(7)
    synth = rng.uniform(-0.25, 1.25, (33, 47, 4)).astype(np.float32)
    synth[0, :8, 0] = [np.nan, np.inf, -np.inf, 0.0, -0.0, 1.0, 0.5, 2.0]
    k = np.arange(33 * 47, dtype=np.float32) % 256
    synth[..., 3] = ((k + 0.5) / 255.0).reshape(33, 47)
    synth[..., 2] = np.nextafter(synth[..., 3], np.float32(0))
    for f in (frame, synth):
        got = hip.present(f)
        want = oracle.present(f)
        assert got.dtype == np.uint8 and got.shape == f.shape
        assert np.array_equal(got, want)
    assert hip.present(frame)[..., 3].max() == 255
