"""ctypes bindings of the CHECKERS — test infrastructure only.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
It loads
  * oracle/_build/liboracle.so                       the CPU restatement of the shader (pt_oracle.cpp)
  * oracle/_ref/libunity-webgpu-pathtracer-plugin-ref.so   the REFERENCE plugin compiled in place
  * oracle/_ref/libref_tinybvh.so                    the reference's tinybvh CPU walkers (ref_driver.cpp)
The product package (unity_webgpu_pathtracer_amd) never imports anything from here.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from unity_webgpu_pathtracer_amd import abi  # noqa: E402  (struct definitions of the public ABI only)

ORACLE_SO = os.path.join(_HERE, "_build", "liboracle.so")
REF_PLUGIN_SO = os.path.join(_HERE, "_ref", "libunity-webgpu-pathtracer-plugin-ref.so")
REF_TINYBVH_SO = os.path.join(_HERE, "_ref", "libref_tinybvh.so")
REF_CWBVH_SO = os.path.join(_HERE, "_ref", "libref_cwbvh.so")


class OracleRay(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("direction", C.c_float * 3), ("tmax", C.c_float), ("kind", C.c_float)]


ORACLE_RAY_DTYPE = np.dtype([("origin", "<f4", 3), ("direction", "<f4", 3), ("tmax", "<f4"), ("kind", "<f4")])


class OracleOptions(C.Structure):
    _fields_ = [("x0", C.c_int32), ("y0", C.c_int32), ("x1", C.c_int32), ("y1", C.c_int32),
                ("threads", C.c_int32), ("shadowAnyHit", C.c_int32),
                ("tileRank", C.c_int32), ("tileWorld", C.c_int32),
                ("rayDump", C.c_void_p), ("rayDumpCapacity", C.c_uint64), ("rayDumpCount", C.c_uint64)]


def build(ref: bool = True, quiet: bool = True):
    """(Re)build the checkers with oracle/Makefile.  The reference part is skipped when /root/reference is absent."""
    out = subprocess.DEVNULL if quiet else None
    subprocess.check_call(["make", "-C", _HERE], stdout=out)
    if ref:
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=out)


_oracle = None


def load_oracle():
    global _oracle
    if _oracle is None:
        path = os.environ.get("PT_ORACLE_LIB") or ORACLE_SO       # PT_ORACLE_LIB: the sanitizer build (make -C oracle asan-test)
        if not os.path.exists(path):
            build(ref=False)
        lib = C.CDLL(path)
        lib.oracle_render.restype = C.c_int
        lib.oracle_render.argtypes = [C.POINTER(abi.PTSceneDesc), C.POINTER(abi.PTFrameParams), C.c_void_p,
                                      C.c_void_p, C.POINTER(abi.PTStats), C.POINTER(OracleOptions)]
        lib.oracle_trace.restype = C.c_int
        lib.oracle_trace.argtypes = [C.POINTER(abi.PTSceneDesc), C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.oracle_trace_uv.restype = C.c_int
        lib.oracle_trace_uv.argtypes = [C.POINTER(abi.PTSceneDesc), C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        lib.oracle_random_float.restype = C.c_float
        lib.oracle_random_float.argtypes = [C.POINTER(C.c_uint32)]
        lib.oracle_math.restype = C.c_float
        lib.oracle_math.argtypes = [C.c_int, C.c_float, C.c_float]
        lib.oracle_eval_brdf.restype = None
        lib.oracle_eval_brdf.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
        lib.oracle_sample_brdf.restype = None
        lib.oracle_sample_brdf.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.POINTER(C.c_uint32), C.c_void_p, C.c_void_p]
        lib.oracle_eval_brdf_batch.restype = None
        lib.oracle_eval_brdf_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_uint64, C.c_void_p]
        lib.oracle_sample_brdf_batch.restype = None
        lib.oracle_sample_brdf_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.POINTER(C.c_uint32), C.c_uint64, C.c_void_p, C.c_void_p]
        lib.oracle_process_mesh.restype = C.c_int
        lib.oracle_process_mesh.argtypes = [C.POINTER(abi.PTMeshDesc), C.c_void_p, C.c_void_p]
        lib.oracle_copy_texture_data.restype = C.c_int
        lib.oracle_copy_texture_data.argtypes = [C.POINTER(abi.PTTextureDesc), C.c_uint32, C.c_void_p]
        lib.oracle_present.restype = C.c_int
        lib.oracle_present.argtypes = [C.POINTER(abi.PTPresentParams), C.c_void_p, C.c_void_p]
        lib.oracle_env_probe.restype = C.c_float
        lib.oracle_env_probe.argtypes = [C.POINTER(abi.PTSceneDesc), C.POINTER(abi.PTFrameParams), C.c_int, C.c_void_p, C.c_uint64, C.c_void_p]
        _oracle = lib
    return _oracle


_ref_plugin = None


def load_ref_plugin():
    """The reference plugin compiled from /root/reference/Assets/Plugins/Web/plugin.cpp (None if not built)."""
    global _ref_plugin
    if _ref_plugin is None:
        if not os.path.exists(REF_PLUGIN_SO):
            return None
        lib = C.CDLL(REF_PLUGIN_SO)
        lib.BuildBVH.restype = C.c_int
        lib.BuildBVH.argtypes = [C.c_void_p, C.c_int]
        lib.DestroyBVH.argtypes = [C.c_int]
        lib.IsBVHReady.restype = C.c_bool
        lib.IsBVHReady.argtypes = [C.c_int]
        lib.GetCWBVHNodesSize.restype = C.c_int
        lib.GetCWBVHNodesSize.argtypes = [C.c_int]
        lib.GetCWBVHTrisSize.restype = C.c_int
        lib.GetCWBVHTrisSize.argtypes = [C.c_int]
        lib.GetCWBVHData.restype = C.c_bool
        lib.GetCWBVHData.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
        lib.BuildTLAS.restype = C.c_int
        lib.BuildTLAS.argtypes = [C.c_void_p, C.c_int]
        lib.DestroyTLAS.argtypes = [C.c_int]
        lib.IsTLASReady.restype = C.c_bool
        lib.IsTLASReady.argtypes = [C.c_int]
        lib.GetTLASNodesSize.restype = C.c_int
        lib.GetTLASNodesSize.argtypes = [C.c_int]
        lib.GetTLASData.restype = C.c_bool
        lib.GetTLASData.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
        _ref_plugin = lib
    return _ref_plugin


def ref_build_cwbvh(vertices: np.ndarray):
    """CWBVH (nodes, tris) as uint8 arrays, produced by the REFERENCE plugin (BuildBVH, plugin.cpp:35-40)."""
    lib = load_ref_plugin()
    assert lib is not None, "reference plugin not built (make -C oracle ref)"
    v = np.ascontiguousarray(vertices, dtype=np.float32)
    h = lib.BuildBVH(v.ctypes.data_as(C.c_void_p), v.shape[0] // 3)
    assert h >= 0 and lib.IsBVHReady(h)
    nb, tb = lib.GetCWBVHNodesSize(h), lib.GetCWBVHTrisSize(h)
    pn, pt = C.c_void_p(), C.c_void_p()
    assert lib.GetCWBVHData(h, C.byref(pn), C.byref(pt))
    nodes = np.ctypeslib.as_array(C.cast(pn, C.POINTER(C.c_uint8)), shape=(nb,)).copy()
    tris = np.ctypeslib.as_array(C.cast(pt, C.POINTER(C.c_uint8)), shape=(tb,)).copy()
    lib.DestroyBVH(h)
    return nodes, tris


def ref_build_tlas(instances: np.ndarray):
    """(nodes uint8, indices uint32) from the REFERENCE BuildTLAS (plugin.cpp:111-118)."""
    lib = load_ref_plugin()
    assert lib is not None
    inst = np.ascontiguousarray(instances)
    n = inst.shape[0]
    h = lib.BuildTLAS(inst.ctypes.data_as(C.c_void_p), n)
    assert h >= 0 and lib.IsTLASReady(h)
    nb = lib.GetTLASNodesSize(h)
    pn, pi = C.c_void_p(), C.c_void_p()
    assert lib.GetTLASData(h, C.byref(pn), C.byref(pi))
    nodes = np.ctypeslib.as_array(C.cast(pn, C.POINTER(C.c_uint8)), shape=(nb,)).copy()
    idx = np.ctypeslib.as_array(C.cast(pi, C.POINTER(C.c_uint32)), shape=(n,)).copy()
    lib.DestroyTLAS(h)
    return nodes, idx


_ref_tinybvh = None


def load_ref_tinybvh():
    global _ref_tinybvh
    if _ref_tinybvh is None:
        if not os.path.exists(REF_TINYBVH_SO):
            return None
        lib = C.CDLL(REF_TINYBVH_SO)
        lib.ref_bvh_build.restype = C.c_void_p
        lib.ref_bvh_build.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double)]
        lib.ref_bvh_destroy.argtypes = [C.c_void_p]
        lib.ref_bvh_used_nodes.restype = C.c_int
        lib.ref_bvh_used_nodes.argtypes = [C.c_void_p]
        lib.ref_bvh_trace.restype = C.c_double
        lib.ref_bvh_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p]
        _ref_tinybvh = lib
    return _ref_tinybvh


_ref_cwbvh = None


def load_ref_cwbvh():
    """oracle/ref_driver_cwbvh.cpp: the reference's own CWBVH walker, BVH8_CWBVH::Intersect (tiny_bvh.h:7006-7114)."""
    global _ref_cwbvh
    if _ref_cwbvh is None:
        if not os.path.exists(REF_CWBVH_SO):
            return None
        lib = C.CDLL(REF_CWBVH_SO)
        lib.ref_cwbvh_build.restype = C.c_void_p
        lib.ref_cwbvh_build.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double)]
        lib.ref_cwbvh_destroy.argtypes = [C.c_void_p]
        lib.ref_cwbvh_data.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_longlong), C.POINTER(C.c_void_p), C.POINTER(C.c_longlong)]
        lib.ref_cwbvh_trace.restype = C.c_double
        lib.ref_cwbvh_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p]
        _ref_cwbvh = lib
    return _ref_cwbvh


def ref_cwbvh_trace(vertices: np.ndarray, raw_rays: np.ndarray, threads=1):
    """Builds the reference CWBVH (scalar pipeline) over `vertices` (3T x 4 float32) and walks it with the reference's
    BVH8_CWBVH::Intersect for every ray of raw_rays (n x 8 float32: o, d, tmax, kind).
    Returns (out[n,4] = t, u, v, prim bits; node bytes; tri bytes)."""
    lib = load_ref_cwbvh()
    verts = np.ascontiguousarray(vertices, np.float32)
    raw = np.ascontiguousarray(raw_rays, np.float32)
    h = lib.ref_cwbvh_build(verts.ctypes.data_as(C.c_void_p), verts.shape[0] // 3, None)
    pn, bn, ptr, bt = C.c_void_p(), C.c_longlong(), C.c_void_p(), C.c_longlong()
    lib.ref_cwbvh_data(h, C.byref(pn), C.byref(bn), C.byref(ptr), C.byref(bt))
    nodes = np.frombuffer((C.c_uint8 * bn.value).from_address(pn.value), np.uint8).copy()
    tris = np.frombuffer((C.c_uint8 * bt.value).from_address(ptr.value), np.uint8).copy()
    out = np.zeros((raw.shape[0], 4), np.float32)
    lib.ref_cwbvh_trace(h, raw.ctypes.data_as(C.c_void_p), raw.shape[0], threads, out.ctypes.data_as(C.c_void_p))
    lib.ref_cwbvh_destroy(h)
    return out, nodes, tris


class SceneBuffers:
    """Keeps the numpy arrays alive behind a PTSceneDesc."""

    def __init__(self, scene, nodes, tris, tlas_data=None, tlas_index_offset=0, gpu_instances=None):
        self.scene = scene
        self.nodes = np.ascontiguousarray(nodes).view(np.uint8)
        self.tris = np.ascontiguousarray(tris).view(np.uint8)
        self.attrs = np.ascontiguousarray(scene.tri_attrs)
        self.materials = np.ascontiguousarray(scene.materials, dtype=np.float32)
        self.lights = np.ascontiguousarray(scene.lights, dtype=np.float32)
        self.tex = np.ascontiguousarray(scene.texture_data, dtype=np.uint32)
        d = abi.PTSceneDesc()
        d.bvhNodes = self.nodes.ctypes.data
        d.bvhNodesBytes = self.nodes.nbytes
        d.bvhTris = self.tris.ctypes.data
        d.bvhTrisBytes = self.tris.nbytes
        d.triAttrs = self.attrs.ctypes.data
        d.triAttrsBytes = self.attrs.nbytes
        d.materials = self.materials.ctypes.data
        d.materialCount = self.materials.shape[0]
        d.lights = self.lights.ctypes.data if self.lights.size else None
        d.lightCount = self.lights.shape[0]
        d.textureData = self.tex.ctypes.data if self.tex.size else None
        d.textureDataUints = self.tex.size
        d.features = scene.features
        if tlas_data is not None:
            self.tlas = np.ascontiguousarray(tlas_data, dtype=np.float32)
            self.inst = np.ascontiguousarray(gpu_instances)
            d.tlasData = self.tlas.ctypes.data
            d.tlasDataFloats = self.tlas.size
            d.tlasIndexOffset = tlas_index_offset
            d.instanceCount = self.inst.shape[0]
            d.gpuInstances = self.inst.ctypes.data
        if scene.environment_texture is not None:
            self.env = np.ascontiguousarray(scene.environment_texture, dtype=np.float32)
            d.envTexture = self.env.ctypes.data
            d.envHeight, d.envWidth = self.env.shape[:2]
        self.desc = d


def buffers_from_bvhscene(bvh_scene):
    """SceneBuffers for the oracle from the arrays the product-side BVHScene mirror assembled (plain data hand-over)."""
    return SceneBuffers(bvh_scene.scene, bvh_scene.bvh_nodes, bvh_scene.bvh_tris, bvh_scene.tlas_data,
                        bvh_scene.tlas_index_offset, bvh_scene.gpu_instances)


def render(buffers: SceneBuffers, params, accumulated=None, threads=None, window=None, shadow_any_hit=False,
           tile_rank=0, tile_world=1, ray_dump_capacity=0):
    """One pass through the oracle.  Returns (frame (H, W, 4) float32, PTStats[, rays])."""
    lib = load_oracle()
    W, H = params.OutputWidth, params.OutputHeight
    out = np.zeros((H, W, 4), dtype=np.float32)
    opt = OracleOptions()
    if window is not None:
        opt.x0, opt.y0, opt.x1, opt.y1 = window
    opt.threads = threads if threads is not None else min(8, os.cpu_count() or 1)
    opt.shadowAnyHit = 1 if shadow_any_hit else 0
    opt.tileRank, opt.tileWorld = tile_rank, tile_world
    dump = None
    if ray_dump_capacity:
        dump = np.zeros(ray_dump_capacity, dtype=ORACLE_RAY_DTYPE)
        opt.rayDump = dump.ctypes.data
        opt.rayDumpCapacity = ray_dump_capacity
    acc_p = None
    if accumulated is not None:
        accumulated = np.ascontiguousarray(accumulated, dtype=np.float32)
        acc_p = accumulated.ctypes.data
    st = abi.PTStats()
    rc = lib.oracle_render(C.byref(buffers.desc), C.byref(params), acc_p, out.ctypes.data, C.byref(st), C.byref(opt))
    assert rc == 0, rc
    if ray_dump_capacity:
        return out, st, dump[: opt.rayDumpCount]
    return out, st


def trace(buffers: SceneBuffers, rays: np.ndarray):
    """Restated CWBVH traversal over explicit rays -> (t, prim, steps)."""
    lib = load_oracle()
    rays = np.ascontiguousarray(rays, dtype=ORACLE_RAY_DTYPE)
    n = rays.shape[0]
    t = np.zeros(n, dtype=np.float32)
    prim = np.zeros(n, dtype=np.uint32)
    steps = np.zeros(n, dtype=np.uint32)
    rc = lib.oracle_trace(C.byref(buffers.desc), rays.ctypes.data, n, t.ctypes.data, prim.ctypes.data, steps.ctypes.data)
    assert rc == 0
    return t, prim, steps


def trace_uv(buffers: SceneBuffers, rays: np.ndarray):
    """Restated CWBVH traversal over explicit rays -> (out[n,4] = t, u, v, prim bits; nodeVisits; triTests)."""
    lib = load_oracle()
    rays = np.ascontiguousarray(rays, dtype=ORACLE_RAY_DTYPE)
    n = rays.shape[0]
    out = np.zeros((n, 4), dtype=np.float32)
    nv, tt = C.c_uint64(), C.c_uint64()
    rc = lib.oracle_trace_uv(C.byref(buffers.desc), rays.ctypes.data, n, out.ctypes.data, C.byref(nv), C.byref(tt))
    assert rc == 0
    return out, nv.value, tt.value


def env_probe(buffers: SceneBuffers, params, what: int, values: np.ndarray):
    """oracle_env_probe: what 0 CDF[i], 1 BinarySearch(value), 2 EvalEnvMap(dir), 3 SampleEnvMap(rng state bits).
    Returns (array, EnvironmentCdfSum)."""
    lib = load_oracle()
    per_in = {0: 1, 1: 1, 2: 3, 3: 1}[what]
    per_out = {0: 1, 1: 2, 2: 4, 3: 8}[what]
    if what == 3:
        vin = np.ascontiguousarray(values, dtype=np.uint32).view(np.float32)
    else:
        vin = np.ascontiguousarray(values, dtype=np.float32)
    n = vin.size // per_in
    out = np.zeros((n, per_out), dtype=np.float32)
    total = lib.oracle_env_probe(C.byref(buffers.desc), C.byref(params), what, vin.ctypes.data, n, out.ctypes.data)
    return out, float(total)


def present(params, frame: np.ndarray) -> np.ndarray:
    """oracle_present: Presentation.shader over a linear (H, W, 4) float32 frame."""
    src = np.ascontiguousarray(frame, dtype=np.float32)
    dst = np.empty_like(src)
    rc = load_oracle().oracle_present(C.byref(params), src.ctypes.data, dst.ctypes.data)
    assert rc == 0
    return dst


def process_meshes(descs, total_triangles: int):
    """oracle_process_mesh over a list of abi.PTMeshDesc -> (vertex positions (T*3, 4) f32, attributes (T,) TRI_ATTR)."""
    lib = load_oracle()
    pos = np.zeros((total_triangles * 3, 4), dtype=np.float32)
    attrs = np.zeros(total_triangles, dtype=abi.TRI_ATTR)
    for d in descs:
        assert lib.oracle_process_mesh(C.byref(d), pos.ctypes.data, attrs.ctypes.data) == 0
    return pos, attrs


def copy_texture_data(images):
    """oracle_copy_texture_data over [(h, w, 4) float32 image, has_alpha] pairs -> TextureData uint32."""
    lib = load_oracle()
    arr = (abi.PTTextureDesc * len(images))()
    keep = []
    total = 4 * len(images)
    for i, (img, has_alpha) in enumerate(images):
        a = np.ascontiguousarray(img, dtype=np.float32)
        keep.append(a)
        arr[i].texels, arr[i].height, arr[i].width, arr[i].hasAlpha = a.ctypes.data, a.shape[0], a.shape[1], 1 if has_alpha else 0
        total += a.shape[0] * a.shape[1]
    out = np.zeros(total, dtype=np.uint32)
    assert lib.oracle_copy_texture_data(arr, len(images), out.ctypes.data) == 0
    return out
