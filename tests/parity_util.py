"""Shared comparison helpers for the parity tests."""
import numpy as np

# BASELINE.json north_star: radiance within 1e-4 per-channel RMS of the reference path.
RMS_TOL = 1e-4


def per_channel_rms(a, b):
    a = np.asarray(a, np.float64).reshape(-1, a.shape[-1])
    b = np.asarray(b, np.float64).reshape(-1, b.shape[-1])
    both_nan = np.isnan(a) & np.isnan(b)
    d = np.where(both_nan, 0.0, a - b)
    return np.sqrt(np.mean(d * d, axis=0))


def bit_mismatches(a, b):
    """Number of float32 elements whose bit patterns differ (NaN payloads ignored: NaN == NaN)."""
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    same = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
    return int((~same).sum())


def assert_parity(got, want, what=""):
    assert got.shape == want.shape, (got.shape, want.shape)
    nan_got, nan_want = np.isnan(got), np.isnan(want)
    assert (nan_got == nan_want).all(), "%s: NaN pixels differ (%d vs %d)" % (what, nan_got.sum(), nan_want.sum())
    rms = per_channel_rms(got, want)
    assert (rms <= RMS_TOL).all(), "%s: per-channel RMS %s exceeds %g" % (what, rms, RMS_TOL)
    return rms, bit_mismatches(got, want)
