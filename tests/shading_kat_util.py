"""Feed one row of tests/golden/shading_kat.json to the CPU oracle (bit patterns in, bit patterns out)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def _f(bits_list):
    return np.array(bits_list, np.uint32).view(np.float32)


_LIB = None


def _lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    import flx_oracle
    l = flx_oracle.lib()
    fp = C.POINTER(C.c_float)
    l.flx_oracle_forward_trace.argtypes = [fp, fp, C.c_float, fp, fp, fp]
    l.flx_oracle_reservoir_sample.argtypes = [fp, C.c_uint32, C.c_float, fp, fp, fp, fp, fp, fp, C.c_float, C.c_int, C.c_int, fp]
    l.flx_oracle_light_trace_bounce.argtypes = [fp, fp, fp, fp, C.c_int, fp, C.c_uint32, fp, C.c_float, C.c_float, fp, fp, fp, fp, C.c_float, fp]
    _LIB = l
    return l


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def oracle_forward_trace(in_bits):
    """[albedo 3, rme 3, lightDir 3, strength, N 3, V 3] -> [r, g, b] as bits"""
    x = _f(in_bits)
    material = np.concatenate([x[0:3], x[3:6], np.zeros(3, np.float32)]).astype(np.float32)
    light, n, v = np.ascontiguousarray(x[6:9]), np.ascontiguousarray(x[10:13]), np.ascontiguousarray(x[13:16])
    out = np.zeros(3, np.float32)
    _lib().flx_oracle_forward_trace(_p(material), _p(light), C.c_float(float(x[9])), _p(n), _p(v), _p(out))
    return [int(b) for b in out.view(np.uint32)]


def oracle_reservoir(row):
    x = _f(row["in"])
    lights = _f([b for lt in row["lights"] for b in lt]) if row["lights"] else np.zeros(6, np.float32)
    lights = np.ascontiguousarray(lights, np.float32)
    material = np.concatenate([x[0:3], x[3:6], np.zeros(3, np.float32)]).astype(np.float32)
    origin, unit_dir, rv = np.ascontiguousarray(x[6:9]), np.ascontiguousarray(x[9:12]), np.ascontiguousarray(x[12:16])
    n, sn = np.ascontiguousarray(x[16:19]), np.ascontiguousarray(x[19:22])
    out = np.zeros(4, np.float32)
    _lib().flx_oracle_reservoir_sample(_p(lights), len(row["lights"]), C.c_float(float(_f([row["random_seed"]])[0])), _p(material), _p(origin), _p(unit_dir),
                                       _p(rv), _p(n), _p(sn), C.c_float(float(x[22])), row["dont_filter"], row["i"], _p(out))
    return [int(b) for b in out.view(np.uint32)]


def oracle_bounce(row):
    """one row of the "bounce" table -> the 17 outputs of flx_oracle_light_trace_bounce as bits"""
    t = row["transform"]
    # scene arrays as the host layer lays them out: per transform 2 x (3 columns of 4 floats) and 2 x 4 floats (forward, inverse)
    rot = np.zeros(24 * (t + 1), np.float32)
    shift = np.zeros(8 * (t + 1), np.float32)
    for k in range(t + 1):
        fwd = _f(row["rotation"]).reshape(3, 3) if k == t else np.eye(3, dtype=np.float32)
        inv = _f(row["rotation_inv"]).reshape(3, 3) if k == t else np.eye(3, dtype=np.float32)
        for c in range(3):
            rot[24 * k + 4 * c: 24 * k + 4 * c + 3] = fwd[c]
            rot[24 * k + 12 + 4 * c: 24 * k + 12 + 4 * c + 3] = inv[c]
        if k == t:
            shift[8 * k: 8 * k + 3] = _f(row["shift"])
            shift[8 * k + 4: 8 * k + 7] = _f(row["shift_inv"])
    geometry = np.ascontiguousarray(_f(row["geometry"]))
    attributes = np.ascontiguousarray(_f(row["attributes"]))
    lights = np.ascontiguousarray(_f([b for lt in row["lights"] for b in lt]))
    ambient, ndc = np.ascontiguousarray(_f(row["ambient"])), np.ascontiguousarray(_f(row["ndc"]))
    camera, dir0, suv = np.ascontiguousarray(_f(row["camera"])), np.ascontiguousarray(_f(row["dir0"])), np.ascontiguousarray(_f(row["suv"]))
    out = np.zeros(17, np.float32)
    _lib().flx_oracle_light_trace_bounce(_p(geometry), _p(attributes), _p(rot), _p(shift), t, _p(lights), len(row["lights"]), _p(ambient),
                                         C.c_float(float(_f([row["random_seed"]])[0])), C.c_float(0.0), _p(ndc), _p(camera), _p(dir0), _p(suv),
                                         C.c_float(float(_f([row["cos_sample_n"]])[0])), _p(out))
    return [int(b) for b in out.view(np.uint32)]
