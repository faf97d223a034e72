"""The drop-in boundary: the shared library loads, exports every symbol include/ptmi_plugin.h declares,
the ctypes mirrors have the C struct sizes, and the product path fails loudly without a HIP device."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import numpy as np
import pytest

from unity_webgpu_pathtracer_amd import abi, plugin

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ptmi_plugin.h")


def _declared_symbols():
    text = open(HEADER).read()
    return re.findall(r"PT_API\s+[\w\s\*]+?\b(\w+)\s*\(", text)


def test_header_symbols_are_exported():
    lib = plugin.load_library()
    declared = _declared_symbols()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in ptmi_plugin.h but not exported"
    assert sorted(declared) == sorted(plugin.EXPORTED_SYMBOLS)
    # the 12 functions the C# host binds (Assets/Scripts/util/TinyBVH.cs:15-50) + GetBVH (plugin.cpp:23-28)
    for name in ["BuildBVH", "DestroyBVH", "IsBVHReady", "GetBVHPtr", "GetCWBVHNodesSize", "GetCWBVHTrisSize", "GetCWBVHData",
                 "BuildTLAS", "DestroyTLAS", "IsTLASReady", "GetTLASNodesSize", "GetTLASData", "GetBVH"]:
        assert name in declared


def test_library_name_matches_reference_cmake():
    # Plugin/CMakeLists.txt:3 / TinyBVH.cs:8-12: "unity-webgpu-pathtracer-plugin"
    assert os.path.basename(plugin.LIB_PATH) == "libunity-webgpu-pathtracer-plugin.so"
    out = subprocess.check_output(["nm", "-D", "--defined-only", plugin.LIB_PATH]).decode()
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    assert set(plugin.EXPORTED_SYMBOLS) <= exported


def test_struct_sizes_match_c_headers():
    src = r"""
    #include <stdio.h>
    #include "ptmi_plugin.h"
    int main(void) {
      printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(PTSceneDesc), sizeof(PTFrameParams), sizeof(PTStats),
             sizeof(PTTimings), sizeof(PTCwbvhNode), sizeof(PTCwbvhTri), sizeof(PTTriangleAttributes), sizeof(PTMaterialData),
             sizeof(PTLight), sizeof(PTTlasNode), sizeof(PTBlasInstance), sizeof(PTGpuInstance), sizeof(PTPresentParams));
      return 0; }
    """
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "probe.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "probe")
        subprocess.check_call(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), c, "-o", exe])   # plain C: the header is C-clean
        sizes = [int(x) for x in subprocess.check_output([exe]).split()]
    assert sizes[0] == C.sizeof(abi.PTSceneDesc)
    assert sizes[1] == C.sizeof(abi.PTFrameParams)
    assert sizes[2] == C.sizeof(abi.PTStats) == 16 * 8
    assert sizes[3] == C.sizeof(abi.PTTimings)
    assert sizes[4:12] == [80, 48, 128, 128, 64, 64, 192, 144]
    assert sizes[12] == C.sizeof(abi.PTPresentParams) == 36
    assert abi.CWBVH_NODE.itemsize == 80 and abi.TRI_ATTR.itemsize == 128 and abi.BLAS_INSTANCE.itemsize == 192


def test_handle_table_semantics():
    """plugin.cpp:8-21,42-58: first free slot is reused, invalid handles read as 0 / false, destroy is idempotent."""
    T = plugin.TinyBVH
    v = np.zeros((3, 4), np.float32)
    v[1, 0] = 1
    v[2, 1] = 1
    p = v.ctypes.data_as(C.c_void_p)
    a, b = T.BuildBVH(p, 1), T.BuildBVH(p, 1)
    assert a >= 0 and b == a + 1
    assert T.IsBVHReady(a) and T.IsBVHReady(b) and T.GetBVHPtr(a)
    T.DestroyBVH(a)
    assert not T.IsBVHReady(a) and T.GetCWBVHNodesSize(a) == 0 and T.GetCWBVHTrisSize(a) == 0
    assert T.GetCWBVHData(a)[0] is False
    T.DestroyBVH(a)                                  # no-op
    c = T.BuildBVH(p, 1)
    assert c == a                                    # first free slot
    assert T.GetCWBVHNodesSize(c) == 80 and T.GetCWBVHTrisSize(c) == 48
    for h in (b, c):
        T.DestroyBVH(h)
    for bad in (-1, 10 ** 6):
        assert not T.IsBVHReady(bad) and T.GetCWBVHNodesSize(bad) == 0 and not T.IsTLASReady(bad) and T.GetTLASNodesSize(bad) == 0
        T.DestroyBVH(bad)
        T.DestroyTLAS(bad)
    # degenerate input: the reference exit(1)s (tiny_bvh.h:1615-1618, 2253); the drop-in returns -1
    assert T.BuildBVH(p, 0) == -1 and T.BuildBVH(None, 1) == -1 and T.BuildTLAS(None, 0) == -1


def test_no_cpu_fallback_without_gpu():
    """Without a HIP device the render path must fail loudly (PT_ERR_NO_DEVICE), never fall back to a CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: covered by the gpu tests")
    lib = plugin.load_library()
    ctx = C.c_void_p()
    rc = lib.PTCreate(0, C.byref(ctx))
    assert rc == abi.PT_ERR_NO_DEVICE and not ctx.value
    assert b"no CPU fallback" in lib.PTGetLastError()
    from unity_webgpu_pathtracer_amd import scenes
    from unity_webgpu_pathtracer_amd.pathtracer import PathTracer
    with pytest.raises(plugin.PluginError):
        PathTracer(scenes.cornell_box())


def test_product_never_imports_oracle():
    """Nothing under unity_webgpu_pathtracer_amd/ may import, link or load anything under oracle/."""
    pkg = os.path.join(ROOT, "unity_webgpu_pathtracer_amd")
    for base, _, files in os.walk(pkg):
        if "_obj" in base or "__pycache__" in base:
            continue
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", "Makefile")):
                text = open(os.path.join(base, f), errors="ignore").read()
                for needle in ("pyoracle", "liboracle", "pt_oracle", "oracle/", "_ref/", "ref_driver"):
                    assert needle not in text, f"{f} mentions {needle}"
    ldd = subprocess.check_output(["ldd", plugin.LIB_PATH]).decode()
    assert "oracle" not in ldd and "tinybvh" not in ldd
