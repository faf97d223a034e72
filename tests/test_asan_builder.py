"""Sanitizer run of the product's host-side CPU BVH builder (make -C oracle asan-test): the same bvh_builder.cpp compiled with
-fsanitize=address,undefined must reproduce the committed golden CWBVH / TLAS bytes without a report.  Skipped in a normal run
(PT_ASAN_BUILDER is set by the make target, which also preloads the sanitizer runtimes)."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

LIB = os.environ.get("PT_ASAN_BUILDER")
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

pytestmark = pytest.mark.skipif(not LIB, reason="sanitizer builder not requested (make -C oracle asan-test)")


def test_cwbvh_goldens_under_asan():
    lib = C.CDLL(LIB)
    lib.asan_build_cwbvh.restype = C.c_longlong
    lib.asan_build_cwbvh.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_longlong, C.c_void_p, C.c_longlong, C.POINTER(C.c_longlong)]
    files = sorted(glob.glob(os.path.join(GOLDEN, "cwbvh_*.npz")))
    assert files
    for f in files:
        g = np.load(f)
        v = np.ascontiguousarray(g["vertices"], np.float32)
        n = v.shape[0] // 3
        nodes = np.zeros(g["nodes"].size + 80, np.uint8)
        tris = np.zeros(g["tris"].size, np.uint8)
        tb = C.c_longlong()
        nb = lib.asan_build_cwbvh(v.ctypes.data, n, nodes.ctypes.data, nodes.size, tris.ctypes.data, tris.size, C.byref(tb))
        assert nb == g["nodes"].size and tb.value == g["tris"].size, f
        assert np.array_equal(nodes[:nb], g["nodes"].view(np.uint8).reshape(-1)) and np.array_equal(tris, g["tris"].view(np.uint8).reshape(-1)), f


def test_tlas_goldens_under_asan():
    lib = C.CDLL(LIB)
    lib.asan_build_tlas.restype = C.c_longlong
    lib.asan_build_tlas.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_longlong, C.c_void_p]
    for f in sorted(glob.glob(os.path.join(GOLDEN, "tlas_*.npz"))):
        g = np.load(f)
        inst = np.ascontiguousarray(g["instances"])
        count = inst.size // 192
        nodes = np.zeros(g["nodes"].size + 64, np.uint8)
        idx = np.zeros(count, np.uint32)
        nb = lib.asan_build_tlas(inst.ctypes.data, count, nodes.ctypes.data, nodes.size, idx.ctypes.data)
        assert nb == g["nodes"].size, f
        assert np.array_equal(nodes[:nb], g["nodes"].view(np.uint8).reshape(-1)) and np.array_equal(idx, g["indices"].view(np.uint32).reshape(-1)), f
