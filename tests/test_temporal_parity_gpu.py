"""GPU temporal frames against literal sequences (tests/golden/temporal_kat.json.gz, tests/analysis/make_temporal_kat.py: every frame traced from the shader text into the
history rings and averaged by the temporal shader the host generates — SURVEY.md 8f N1), without the oracle in between."""
import numpy as np
import pytest

from test_oracle_kat import _temporal_kat_cases, assert_filter_kat, temporal_kat_expectations

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("k", range(2))
def test_temporal_sequence_literal(hip, scenes, k):
    case = _temporal_kat_cases()[k]
    sc, p, want = temporal_kat_expectations(case, scenes)
    hip.update_scene(sc)
    p.temporal_samples = 4
    hip.temporal_reset()
    try:
        for f in range(want.shape[0]):
            p.random_seed = float(f % 4)
            got, _, _ = hip.render(p)
            assert_filter_kat(np.ascontiguousarray(got), np.ascontiguousarray(want[f]), case["hdr"], "GPU, sequence %d frame %d" % (k, f))
    finally:
        hip.temporal_reset()
