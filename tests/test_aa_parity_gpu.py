"""The GPU's FXAA and TAA passes (flx_fxaa, flx_taa) against literal answers computed from their shader texts (tests/golden/aa_kat.json.gz,
tests/analysis/make_aa_kat.py — SURVEY.md 8f N4), without the oracle in between."""
import pytest

from test_oracle_kat import _aa_kat, aa_kat_frame, assert_aa_kat

pytestmark = pytest.mark.gpu


def test_fxaa_literal(hip):
    for k, c in enumerate(_aa_kat()["fxaa"]):
        W, H = c["width"], c["height"]
        assert_aa_kat(hip.fxaa(aa_kat_frame(c["frame"], W, H)), aa_kat_frame(c["out"], W, H), "GPU FXAA, frame %d" % k)


def test_taa_literal(hip):
    """the context keeps the last nine frames: the table's states are one history, fed oldest first"""
    states = _aa_kat()["taa"]
    W, H = states[0]["width"], states[0]["height"]
    longest = max(states, key=lambda c: len(c["frames_newest_first"]))
    # the states hold the newest 1, 4, 9 and 9-of-11 frames of ONE run: rebuild the run oldest first from the two longest
    run = [aa_kat_frame(f, W, H) for f in states[2]["frames_newest_first"]][::-1]
    tail = [aa_kat_frame(f, W, H) for f in states[3]["frames_newest_first"]][::-1]
    run = run + tail[-2:]
    assert len(run) == 11 and longest is states[2] or len(longest["frames_newest_first"]) == 9
    want_after = {len(c["frames_newest_first"]) if i < 3 else 11: aa_kat_frame(c["out"], W, H) for i, c in enumerate(states)}
    hip.taa_reset()
    try:
        for n, fr in enumerate(run, 1):
            got = hip.taa(fr)
            if n in want_after:
                assert_aa_kat(got, want_after[n], "GPU TAA after %d frames" % n)
    finally:
        hip.taa_reset()
