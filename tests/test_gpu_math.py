"""GPU: the device math the kernels rely on is bit-identical to the host (glibc atan2f/asinf, IEEE divide and sqrt)."""
import ctypes as C
import math
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "csrc", "libmath_probe.so")
SRC = os.path.join(HERE, "csrc", "math_probe.hip")


def build_probe():
    if not os.path.exists(SO) or os.path.getmtime(SRC) > os.path.getmtime(SO):
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
                               "-o", SO, SRC])
    return SO


@pytest.mark.gpu
def test_device_math_matches_host_libm():
    import continuous_clustering_amd
    continuous_clustering_amd.load_library()  # binds the process to one HIP runtime first
    lib = C.CDLL(build_probe())
    rng = np.random.default_rng(0)
    n = 1 << 21
    # lidar-like magnitudes with random signs, plus raw bit patterns, plus asin arguments in [-1, 1]
    a = np.concatenate([rng.uniform(-150, 150, n // 4), rng.uniform(-1, 1, n // 4), rng.normal(0, 1e-3, n // 4),
                        rng.integers(0, 2 ** 32, n // 4, dtype=np.uint64).astype(np.uint32).view(np.float32)]).astype(np.float32)
    b = np.concatenate([rng.uniform(-150, 150, n // 2), rng.uniform(-1e-3, 1e-3, n // 4),
                        rng.integers(0, 2 ** 32, n // 4, dtype=np.uint64).astype(np.uint32).view(np.float32)]).astype(np.float32)
    rng.shuffle(b)
    d = np.abs(rng.uniform(0, 4e4, n)).astype(np.float64)
    outs = [np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.float32),
            np.zeros(n, np.float64), np.zeros(n, np.float32)]
    rc = lib.math_probe(n, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p),
                        *[o.ctypes.data_as(C.c_void_p) for o in outs])
    assert rc == 0
    libm = C.CDLL("libm.so.6")
    libm.atan2f.restype = C.c_float
    libm.atan2f.argtypes = [C.c_float, C.c_float]
    libm.asinf.restype = C.c_float
    libm.asinf.argtypes = [C.c_float]

    def same(x, y):
        nx, ny = np.isnan(x), np.isnan(y)
        it = np.uint32 if x.dtype == np.float32 else np.uint64
        return np.array_equal(nx, ny) and np.array_equal(x[~nx].view(it), y[~ny].view(it))

    with np.errstate(all="ignore"):
        assert same(outs[2], (a / b).astype(np.float32)), "f32 divide is not correctly rounded on the device"
        assert same(outs[3], np.sqrt(np.abs(a)).astype(np.float32)), "f32 sqrt"
        assert same(outs[4], np.sqrt(d)), "f64 sqrt"
        assert same(outs[5], np.sqrt((a * a + b * b).astype(np.float32)).astype(np.float32)), "len2 (FMA contraction?)"
    # libm through ctypes is slow: check a 2^17 subsample element-wise
    idx = rng.choice(n, 1 << 17, replace=False)
    ref_atan2 = np.array([libm.atan2f(float(a[i]), float(b[i])) for i in idx], dtype=np.float32)
    ref_asin = np.array([libm.asinf(float(a[i])) for i in idx], dtype=np.float32)
    assert same(outs[0][idx], ref_atan2), "atan2f"
    assert same(outs[1][idx], ref_asin), "asinf"
