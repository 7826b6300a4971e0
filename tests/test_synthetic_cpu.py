"""The synthetic scene generator feeds well-formed arrays and the oracle is deterministic on them (CPU only)."""
import numpy as np

import synth_scene


def test_generator_is_well_formed_and_oracle_is_thread_independent(oracle):
    sc = synth_scene.make(seed=11, n_objects=4, tris_per_object=20, n_transforms=4, n_lights=2, width=48, height=32)
    g = sc.arrays["geometry"].reshape(-1, 12)
    n = g.shape[0]
    assert n % 256 == 0
    for i in np.flatnonzero(g[:, 10] == 1):
        assert i + g[i, 6] < n and g[i, 6] >= 1
        inside = g[i + 1:i + 1 + int(g[i, 6])]
        tris = inside[inside[:, 10] == 2][:, :9].reshape(-1, 3)
        assert (tris >= g[i, 0:3] - 1e-6).all() and (tris <= g[i, 3:6] + 1e-6).all()
    assert set(np.unique(g[:, 10])) <= {0.0, 1.0, 2.0}
    p = sc.frame_params(use_filter=0)
    a, ca, _ = oracle.render(sc, p, threads=1)
    b, cb, _ = oracle.render(sc, p, threads=4)
    assert np.array_equal(a, b, equal_nan=True) and ca == cb
    assert ca["primary_hits"] > 0 and ca["shadow_walks"] > 0 and ca["atlas_texels"] > 0


def test_no_terminator_scene_has_none(oracle):
    sc = synth_scene.make(seed=4, n_objects=3, tris_per_object=50, n_transforms=3, n_lights=1, exact_multiple=True, width=32, height=24)
    g = sc.arrays["geometry"].reshape(-1, 12)
    assert (g[:, 10] != 0).all()
    img, cnt, _ = oracle.render(sc, sc.frame_params(use_filter=0))
    assert cnt["primary_hits"] > 0 and np.isfinite(img[..., 3]).all()
