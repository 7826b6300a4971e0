#!/usr/bin/env python3
"""A deterministic >= 100 000-triangle stand-in for the reference's absent objects/dragon.obj (.MISSING_LARGE_BLOBS:3;
BASELINE.json names "dragon.obj (~100k tris)", examples/dragon.js:37 loads dragon_lp.obj): every triangle of
objects/dragon_lp.obj (43 569 faces) split 1 -> 4 at its edge midpoints = 174 276 triangles on the same surface.  SYNTHETIC
geometry, labelled so everywhere it is used; not reference data.

    python tools/make_dragon_100k.py [reference root] [output dir]     -> <output dir>/objects/dragon_100k.obj

Midpoint positions and normals are shared per edge (one new vertex per vertex pair, one new normal per normal pair: the sum of
the two, normalised), written with six decimals like the Blender export they come from; faces keep their winding:
(a, ab, ca) (ab, b, bc) (ca, bc, c) (ab, bc, ca).  Authoring container only (needs the reference's OBJ); the scene built from
it is committed as tests/golden/ref_dragon_100k.flxs.gz."""
import math
import os
import sys


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else os.environ.get("FLX_REFERENCE", "/root/reference")
    out_dir = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build", "assets")
    v, vn, faces = [], [], []
    with open(os.path.join(ref, "objects", "dragon_lp.obj")) as fh:
        for line in fh:
            t = line.split()
            if not t:
                continue
            if t[0] == "v":
                v.append(tuple(float(x) for x in t[1:4]))
            elif t[0] == "vn":
                vn.append(tuple(float(x) for x in t[1:4]))
            elif t[0] == "f":
                if len(t) != 4:
                    raise SystemExit("dragon_lp.obj: only triangles expected")
                faces.append([tuple(int(k) for k in c.split("//")) for c in t[1:4]])
    mid_v, mid_n = {}, {}

    def vmid(a, b):
        key = (a, b) if a < b else (b, a)
        if key not in mid_v:
            pa, pb = v[key[0] - 1], v[key[1] - 1]
            v.append(tuple((pa[k] + pb[k]) * 0.5 for k in range(3)))
            mid_v[key] = len(v)
        return mid_v[key]

    def nmid(a, b):
        key = (a, b) if a < b else (b, a)
        if key not in mid_n:
            pa, pb = vn[key[0] - 1], vn[key[1] - 1]
            s = [pa[k] + pb[k] for k in range(3)]
            l = math.sqrt(sum(x * x for x in s))
            vn.append(tuple(x / l for x in s) if l > 0 else pa)
            mid_n[key] = len(vn)
        return mid_n[key]

    out = []
    for (a, b, c) in faces:
        ab = (vmid(a[0], b[0]), nmid(a[1], b[1]))
        bc = (vmid(b[0], c[0]), nmid(b[1], c[1]))
        ca = (vmid(c[0], a[0]), nmid(c[1], a[1]))
        out += [(a, ab, ca), (ab, b, bc), (ca, bc, c), (ab, bc, ca)]
    os.makedirs(os.path.join(out_dir, "objects"), exist_ok=True)
    path = os.path.join(out_dir, "objects", "dragon_100k.obj")
    with open(path, "w") as fh:
        fh.write("# SYNTHETIC: objects/dragon_lp.obj of arbobendik/web-ray-tracer, every triangle split 1 -> 4 at its edge midpoints (tools/make_dragon_100k.py)\n")
        fh.write("o dragon\n")
        for p in v:
            fh.write("v %.6f %.6f %.6f\n" % p)
        for n in vn:
            fh.write("vn %.4f %.4f %.4f\n" % n)
        fh.write("s 1\n")
        for f in out:
            fh.write("f %s\n" % " ".join("%d//%d" % c for c in f))
    print("%s: %d vertices, %d normals, %d triangles" % (path, len(v), len(vn), len(out)))


if __name__ == "__main__":
    main()
