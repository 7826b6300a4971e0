"""Synthetic scenes for parity tests: random triangle soups flattened into the reference's array formats
(SURVEY.md §8a D1–D6) with features the four BASELINE scenes do not exercise — many transforms with
rotation and non-unit scale, several lights or none, textured materials in multi-cell atlases, translucent
and emissive surfaces, degenerate triangles, entry counts that are an exact multiple of 256 (no terminator).
The arrays only have to be well-formed; both the oracle and the GPU consume the same ones."""
import numpy as np

from flexlight_hip.scene_io import Scene


def _flatten(node, geo, att, transform_of):
    """node = ('group', transform, [children]) | ('tri', transform, verts9, attrs28). DFS pre-order with skip counts."""
    if node[0] == "tri":
        g = np.zeros(12, np.float32)
        g[:9] = node[2]
        g[9] = node[1]
        g[10] = 2
        geo.append(g)
        a = np.array(node[3], np.float32)
        att.append(a)
        v = np.asarray(node[2], np.float32).reshape(3, 3)
        return v.min(0), v.max(0)
    at = len(geo)
    geo.append(None)
    att.append(np.zeros(28, np.float32))
    lo, hi = None, None
    for child in node[2]:
        clo, chi = _flatten(child, geo, att, transform_of)
        lo = clo if lo is None else np.minimum(lo, clo)
        hi = chi if hi is None else np.maximum(hi, chi)
    g = np.zeros(12, np.float32)
    g[0:3], g[3:6] = lo, hi
    g[6] = len(geo) - at - 1
    g[9] = node[1]
    g[10] = 1
    geo[at] = g
    return lo, hi


def _bvh(tris, transform, rng, leaf=4):
    if len(tris) <= leaf:
        return ("group", transform, tris)
    cent = np.array([np.asarray(t[2]).reshape(3, 3).mean(0) for t in tris])
    axis = int(np.argmax(cent.max(0) - cent.min(0)))
    order = np.argsort(cent[:, axis])
    half = len(tris) // 2
    return ("group", transform, [_bvh([tris[i] for i in order[:half]], transform, rng, leaf),
                                 _bvh([tris[i] for i in order[half:]], transform, rng, leaf)])


def _rotation(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def make(seed=0, n_objects=3, tris_per_object=40, n_transforms=3, n_lights=2, textured=True, exact_multiple=False,
         degenerate=0, width=96, height=64, samples=2, bounces=3, axis_aligned_view=False):
    rng = np.random.default_rng(seed)
    # transforms: number 0 is the identity (scene.js:590-593); arrays as Transform.buildWGL2Arrays lays them out
    rotation = np.zeros((n_transforms, 24), np.float32)
    shift = np.zeros((n_transforms, 8), np.float32)
    for t in range(n_transforms):
        if t == 0:
            m, pos = np.eye(3), np.zeros(3)
        else:
            m = _rotation(rng) * rng.choice([0.5, 1.0, 2.0, 1.3])
            pos = rng.uniform(-3, 3, 3)
        inv = np.linalg.inv(m)
        for r in range(3):
            rotation[t, 4 * r:4 * r + 3] = m[r]
            rotation[t, 12 + 4 * r:12 + 4 * r + 3] = inv[r]
        shift[t, 0:3] = pos
        shift[t, 4:7] = -pos
    n_tex = 5 if textured else 0
    objects = []
    for o in range(n_objects):
        transform = o % n_transforms
        centre = rng.uniform(-4, 4, 3) + np.array([0, 0, 10.0])
        tris = []
        for k in range(tris_per_object):
            a = centre + rng.normal(scale=1.5, size=3)
            b = a + rng.normal(scale=1.2, size=3)
            c = a + rng.normal(scale=1.2, size=3)
            if k < degenerate:
                c = b.copy()                                   # zero-area triangle: |det| < BIAS path
            n = np.cross(a - c, a - b)
            n = n / (np.linalg.norm(n) + 1e-30)
            attrs = np.zeros(28, np.float32)
            attrs[0:9] = np.tile(n, 3) + rng.normal(scale=0.05, size=9)          # slightly perturbed vertex normals
            attrs[9:15] = rng.uniform(0, 1, 6)
            attrs[15:18] = [rng.integers(-1, n_tex) if textured else -1 for _ in range(3)]
            attrs[18:21] = rng.uniform(0.2, 1.0, 3)
            attrs[21:24] = [rng.uniform(0, 1), rng.uniform(0, 1), rng.choice([0, 0, 0, 2.0])]
            attrs[24:27] = [rng.choice([0, 0, 1.0]), 0, rng.uniform(1.0, 1.8)]
            tris.append(("tri", transform, np.concatenate([a, b, c]), attrs))
        objects.append(_bvh(tris, transform, rng))
    floor = []
    for quad in ([[-30, -6, -5], [30, -6, -5], [30, -6, 40]], [[30, -6, 40], [-30, -6, 40], [-30, -6, -5]]):
        attrs = np.zeros(28, np.float32)
        attrs[0:9] = np.tile([0, 1, 0], 3)
        attrs[15:18] = -1
        attrs[18:24] = [0.8, 0.8, 0.8, 1, 0, 0]
        attrs[24:27] = [0, 0, 1]
        floor.append(("tri", 0, np.array(quad, np.float64).reshape(-1), attrs))
    root = ("group", 0, floor + objects)
    geo, att = [], []
    _flatten(root, geo, att, None)
    if exact_multiple:                                          # pad with extra floor triangles up to a multiple of 256, no terminator
        while len(geo) % 256 != 0:
            g = geo[1].copy(); g[[1, 4, 7]] -= 0.01 * (len(geo) % 7 + 1)
            geo.append(g); att.append(att[1].copy())
        geo[0][6] = len(geo) - 1
    entries = len(geo)
    padded = ((entries + 255) // 256) * 256
    geometry = np.zeros((padded, 12), np.float32)
    attributes = np.zeros((padded, 28), np.float32)
    geometry[:entries] = np.stack(geo)
    attributes[:entries] = np.stack(att)
    ids = np.flatnonzero(geometry[:, 10] == 2).astype(np.int32)
    lights = np.zeros((n_lights, 6), np.float32)
    for i in range(n_lights):
        lights[i] = [rng.uniform(-8, 8), rng.uniform(5, 12), rng.uniform(0, 15), rng.uniform(200, 800), rng.choice([0.0, 0.4]), 0]
    size = 16
    tw = 2048 // size

    def atlas(n):
        if n == 0:
            return np.zeros(4, np.uint8), [1, 1]
        a = rng.integers(0, 256, (size * n, size * tw, 4), dtype=np.uint8)
        return a.reshape(-1), [size * tw, size * n]
    aa, da = atlas(n_tex)
    ap, dp = atlas(n_tex)
    at, dt = atlas(n_tex)
    fx, fy = (0.0, 0.0) if axis_aligned_view else (0.15, 0.2)
    meta = {
        "name": "synthetic-%d" % seed, "textureLength": entries, "bufferLength": int(ids.size), "entriesPadded": padded,
        "transforms": n_transforms, "lights": n_lights,
        "camera": {"x": 0.5 if not axis_aligned_view else 0.0, "y": 1.0 if not axis_aligned_view else 0.0, "z": -8.0, "fx": fx, "fy": fy, "fov": 1 / np.pi},
        "ambient": [0.05, 0.06, 0.07], "textureWidth": tw,
        "atlas": {"albedo": da, "pbr": dp, "tpo": dt},
        "frame": {"width": width, "height": height, "samplesPerRay": samples, "maxReflections": bounces, "filter": False},
    }
    arrays = {
        "geometry": geometry.reshape(-1), "attributes": attributes.reshape(-1), "ids": ids,
        "rotation": rotation.reshape(-1), "shift": shift.reshape(-1), "lights": lights.reshape(-1),
        "atlasAlbedo": aa, "atlasPbr": ap, "atlasTpo": at,
    }
    return Scene(meta, arrays)
