"""ctypes binding of libunity-webgpu-pathtracer-plugin.so — the reference-side view of the drop-in.

`TinyBVH` mirrors Assets/Scripts/util/TinyBVH.cs:15-50 name for name (the [DllImport] block of the C# host),
so parity tests read like calls made by BVHScene.cs.  The PT* render functions are bound on the same
library handle (include/ptmi_plugin.h, Part 2).

The library is the product: if it is missing it is built with hipcc (csrc/Makefile); if that fails the
import fails loudly.  There is no Python or CPU fallback for any of these entry points.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = "libunity-webgpu-pathtracer-plugin.so"      # Plugin/CMakeLists.txt:3 / TinyBVH.cs:8-12
LIB_PATH = os.environ.get("PT_PLUGIN") or os.path.join(_HERE, "lib", LIB_NAME)   # PT_PLUGIN: explicit build to load

_lib = None


def build_library(verbose: bool = False):
    """Compile every HIP/C++ source of the plugin for gfx950 (hipcc cross-compiles without a GPU)."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(["make", "-C", os.path.join(_HERE, "csrc")], stdout=out)
    return LIB_PATH


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build_library()
    # PyTorch-ROCm ships its own libamdhip64 under the same SONAME as /opt/rocm's.  Whichever loads first serves the whole
    # process; torch only finds its GPUs through its own copy, so a host that hands torch buffers to PTRenderPassTo must
    # let torch load first.  (The plugin itself is happy with either copy; a host without torch is unaffected.)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    vp, i32, u32p = C.c_void_p, C.c_int, C.POINTER(C.c_uint32)
    sig = {
        # Part 1 (TinyBVH.cs)
        "BuildBVH": (i32, [vp, i32]), "DestroyBVH": (None, [i32]), "IsBVHReady": (i32, [i32]),
        "GetBVHPtr": (vp, [i32]), "GetBVH": (vp, [i32]),
        "GetCWBVHNodesSize": (i32, [i32]), "GetCWBVHTrisSize": (i32, [i32]),
        "GetCWBVHData": (i32, [i32, C.POINTER(vp), C.POINTER(vp)]),
        "PTBuildBVHDevice": (i32, [i32, vp, i32]), "PTGetBVHBuildError": (C.c_char_p, []), "PTGetBVHBuildMs": (C.c_double, [i32]),
        "BuildTLAS": (i32, [vp, i32]), "DestroyTLAS": (None, [i32]), "IsTLASReady": (i32, [i32]),
        "GetTLASNodesSize": (i32, [i32]), "GetTLASData": (i32, [i32, C.POINTER(vp), C.POINTER(vp)]),
        # Part 2 (render)
        "PTCreate": (i32, [i32, C.POINTER(vp)]), "PTDestroy": (i32, [vp]),
        "PTSetScene": (i32, [vp, C.POINTER(abi.PTSceneDesc)]),
        "PTSetTileOwnership": (i32, [vp, i32, i32]),
        "PTRenderPass": (i32, [vp, C.POINTER(abi.PTFrameParams)]),
        "PTFlipFrames": (i32, [vp]), "PTResetFrames": (i32, [vp]),
        "PTRenderPassTo": (i32, [vp, C.POINTER(abi.PTFrameParams), vp, vp]),
        "PTRenderPassBatchTo": (i32, [vp, C.POINTER(abi.PTFrameParams), i32, vp, vp]),
        "PTRenderPassBatch": (i32, [vp, C.POINTER(abi.PTFrameParams), i32]),
        "PTGroupRenderPassBatch": (i32, [vp, C.POINTER(abi.PTFrameParams), i32]),
        "PTSynchronize": (i32, [vp]), "PTReadback": (i32, [vp, vp, C.c_uint64]),
        "PTGetFramePointer": (vp, [vp, i32]), "PTGetStream": (vp, [vp]),
        "PTSetStatsLevel": (i32, [vp, i32]), "PTGetStats": (i32, [vp, C.POINTER(abi.PTStats)]), "PTResetStats": (i32, [vp]),
        "PTSetProfiling": (i32, [vp, i32]), "PTGetTimings": (i32, [vp, C.POINTER(abi.PTTimings)]), "PTResetTimings": (i32, [vp]),
        "PTProcessMeshes": (i32, [vp, C.POINTER(abi.PTMeshDesc), C.c_uint32, C.c_uint32, vp, vp]),
        "PTCopyTextureData": (i32, [vp, C.POINTER(abi.PTTextureDesc), C.c_uint32, vp, C.c_uint64]),
        "PTPresent": (i32, [vp, C.POINTER(abi.PTPresentParams), vp, vp]),
        "PTPresentToHost": (i32, [vp, C.POINTER(abi.PTPresentParams), vp, C.c_uint64]),
        "PTSetSchedule": (i32, [vp, i32]), "PTGetSchedule": (i32, [vp]), "PTSetWavefrontIterations": (i32, [vp, i32]),
        "PTSetPassesInFlight": (i32, [vp, i32]), "PTGetPassesInFlight": (i32, [vp]), "PTSetSubFrames": (i32, [vp, i32]),
        "PTGetOwnedTileSlots": (i32, [vp, C.POINTER(abi.PTFrameParams), C.POINTER(C.c_uint64)]),
        "PTPackOwnedTiles": (i32, [vp, C.POINTER(abi.PTFrameParams), vp, vp]),
        "PTUnpackTiles": (i32, [vp, C.POINTER(abi.PTFrameParams), i32, i32, vp, vp]),
        "PTCreateMulti": (i32, [C.POINTER(i32), i32, C.POINTER(vp)]), "PTGroupDestroy": (i32, [vp]), "PTGroupSize": (i32, [vp]),
        "PTGroupGetContext": (vp, [vp, i32]), "PTGroupSetScene": (i32, [vp, C.POINTER(abi.PTSceneDesc)]),
        "PTGroupRenderPass": (i32, [vp, C.POINTER(abi.PTFrameParams)]), "PTGroupFlipFrames": (i32, [vp]), "PTGroupResetFrames": (i32, [vp]),
        "PTGroupSynchronize": (i32, [vp]), "PTGroupReadback": (i32, [vp, vp, C.c_uint64]), "PTGroupGetAssembledFrame": (vp, [vp]),
        "PTGroupGetStats": (i32, [vp, C.POINTER(abi.PTStats)]), "PTGroupResetStats": (i32, [vp]),
        "PTGetLastError": (C.c_char_p, []), "PTGetVersion": (i32, []),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)          # AttributeError here = the library does not export what the header declares
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


EXPORTED_SYMBOLS = [
    "BuildBVH", "DestroyBVH", "IsBVHReady", "GetBVHPtr", "GetBVH", "GetCWBVHNodesSize", "GetCWBVHTrisSize", "GetCWBVHData",
    "PTBuildBVHDevice", "PTGetBVHBuildError", "PTGetBVHBuildMs",
    "BuildTLAS", "DestroyTLAS", "IsTLASReady", "GetTLASNodesSize", "GetTLASData",
    "PTCreate", "PTDestroy", "PTSetScene", "PTSetTileOwnership", "PTRenderPass", "PTFlipFrames", "PTResetFrames",
    "PTRenderPassTo", "PTRenderPassBatchTo", "PTRenderPassBatch", "PTGroupRenderPassBatch", "PTSynchronize", "PTReadback", "PTGetFramePointer", "PTGetStream", "PTSetStatsLevel", "PTGetStats",
    "PTResetStats", "PTSetProfiling", "PTGetTimings", "PTResetTimings", "PTProcessMeshes", "PTCopyTextureData", "PTPresent", "PTPresentToHost", "PTSetSchedule", "PTGetSchedule", "PTSetWavefrontIterations", "PTSetPassesInFlight", "PTGetPassesInFlight", "PTSetSubFrames",
    "PTGetOwnedTileSlots", "PTPackOwnedTiles", "PTUnpackTiles", "PTCreateMulti", "PTGroupDestroy", "PTGroupSize", "PTGroupGetContext",
    "PTGroupSetScene", "PTGroupRenderPass", "PTGroupFlipFrames", "PTGroupResetFrames", "PTGroupSynchronize", "PTGroupReadback",
    "PTGroupGetAssembledFrame", "PTGroupGetStats", "PTGroupResetStats",
    "PTGetLastError", "PTGetVersion",
]


class PluginError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"plugin error {code}: {message}")
        self.code = code


def check(rc):
    if rc != abi.PT_OK:
        raise PluginError(rc, load_library().PTGetLastError().decode())
    return rc


class TinyBVH:
    """Static mirror of the C# `TinyBVH` class (Assets/Scripts/util/TinyBVH.cs): same names, same meaning."""

    @staticmethod
    def BuildBVH(verticesPtr, count):
        return load_library().BuildBVH(verticesPtr, count)

    @staticmethod
    def DestroyBVH(index):
        load_library().DestroyBVH(index)

    @staticmethod
    def IsBVHReady(index):
        return bool(load_library().IsBVHReady(index))

    @staticmethod
    def GetBVHPtr(index):
        return load_library().GetBVHPtr(index)

    @staticmethod
    def GetCWBVHNodesSize(index):
        return load_library().GetCWBVHNodesSize(index)

    @staticmethod
    def GetCWBVHTrisSize(index):
        return load_library().GetCWBVHTrisSize(index)

    @staticmethod
    def GetCWBVHData(index):
        """-> (ok, bvhNodes IntPtr, bvhTris IntPtr), like the C# `out IntPtr` pair."""
        n, t = C.c_void_p(), C.c_void_p()
        ok = load_library().GetCWBVHData(index, C.byref(n), C.byref(t))
        return bool(ok), n.value, t.value

    @staticmethod
    def BuildTLAS(instances, instanceCount):
        return load_library().BuildTLAS(instances, instanceCount)

    @staticmethod
    def DestroyTLAS(index):
        load_library().DestroyTLAS(index)

    @staticmethod
    def IsTLASReady(index):
        return bool(load_library().IsTLASReady(index))

    @staticmethod
    def GetTLASNodesSize(index):
        return load_library().GetTLASNodesSize(index)

    @staticmethod
    def GetTLASData(index):
        n, i = C.c_void_p(), C.c_void_p()
        ok = load_library().GetTLASData(index, C.byref(n), C.byref(i))
        return bool(ok), n.value, i.value


def build_cwbvh(vertices: np.ndarray, device: int = None, timing: dict = None):
    """What BVHScene.OnCompleteReadback does with the plugin (BVHScene.cs:629-659): BuildBVH, read the sizes,
    fetch the borrowed pointers, copy the bytes out (Utilities.UploadFromPointer), DestroyBVH.
    device = None: BuildBVH (CPU, byte-identical to the reference plugin); device = k: PTBuildBVHDevice on HIP device k
    (same format, a different tree).  timing (optional dict) receives {"build_ms": ...}.
    Returns (nodes uint8[], tris uint8[])."""
    v = np.ascontiguousarray(vertices, dtype=np.float32)
    assert v.ndim == 2 and v.shape[1] == 4 and v.shape[0] % 3 == 0
    lib = load_library()
    if device is None:
        h = TinyBVH.BuildBVH(v.ctypes.data_as(C.c_void_p), v.shape[0] // 3)
        if h < 0:
            raise PluginError(h, "BuildBVH failed")
    else:
        h = lib.PTBuildBVHDevice(device, v.ctypes.data_as(C.c_void_p), v.shape[0] // 3)
        if h < 0:
            raise PluginError(h, "PTBuildBVHDevice failed: " + lib.PTGetBVHBuildError().decode())
    if timing is not None:
        timing["build_ms"] = float(lib.PTGetBVHBuildMs(h))
    try:
        nb, tb = TinyBVH.GetCWBVHNodesSize(h), TinyBVH.GetCWBVHTrisSize(h)
        ok, pn, pt = TinyBVH.GetCWBVHData(h)
        assert ok
        nodes = np.ctypeslib.as_array(C.cast(pn, C.POINTER(C.c_uint8)), shape=(nb,)).copy()
        tris = np.ctypeslib.as_array(C.cast(pt, C.POINTER(C.c_uint8)), shape=(tb,)).copy()
    finally:
        TinyBVH.DestroyBVH(h)
    return nodes, tris


def build_tlas(instances: np.ndarray):
    inst = np.ascontiguousarray(instances)
    assert inst.dtype == abi.BLAS_INSTANCE
    n = inst.shape[0]
    h = TinyBVH.BuildTLAS(inst.ctypes.data_as(C.c_void_p), n)
    if h < 0:
        raise PluginError(h, "BuildTLAS failed")
    try:
        nb = TinyBVH.GetTLASNodesSize(h)
        ok, pn, pi = TinyBVH.GetTLASData(h)
        assert ok
        nodes = np.ctypeslib.as_array(C.cast(pn, C.POINTER(C.c_uint8)), shape=(nb,)).copy()
        idx = np.ctypeslib.as_array(C.cast(pi, C.POINTER(C.c_uint32)), shape=(n,)).copy()
    finally:
        TinyBVH.DestroyTLAS(h)
    return nodes, idx
