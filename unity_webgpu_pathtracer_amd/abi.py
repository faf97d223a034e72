"""ctypes mirror of include/ptmi_plugin.h and include/ptmi_layouts.h.

These are the blittable structs that cross the drop-in boundary; the field order and
widths are checked against the C headers by tests/test_abi.py (sizeof via a compiled probe).
Reference for every field: see the comments in include/ptmi_plugin.h.
"""
import ctypes as C

import numpy as np

# ---------------------------------------------------------------------------------------
# numpy dtypes of the scene buffers (include/ptmi_layouts.h)
# ---------------------------------------------------------------------------------------
FLOAT4 = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("w", "<f4")])

CWBVH_NODE = np.dtype([
    ("lo", "<f4", 3), ("ex", "u1"), ("ey", "u1"), ("ez", "u1"), ("imask", "u1"),
    ("childBaseIndex", "<u4"), ("triBaseIndex", "<u4"), ("meta", "u1", 8),
    ("qlox", "u1", 8), ("qloy", "u1", 8), ("qloz", "u1", 8),
    ("qhix", "u1", 8), ("qhiy", "u1", 8), ("qhiz", "u1", 8),
])
assert CWBVH_NODE.itemsize == 80

TRI_ATTR = np.dtype([
    ("normal0", "<f4", 3), ("pad0", "<f4"), ("normal1", "<f4", 3), ("pad1", "<f4"),
    ("normal2", "<f4", 3), ("pad2", "<f4"), ("tangent0", "<f4", 3), ("pad3", "<f4"),
    ("tangent1", "<f4", 3), ("pad4", "<f4"), ("tangent2", "<f4", 3), ("pad5", "<f4"),
    ("uv0", "<f4", 2), ("uv1", "<f4", 2), ("uv2", "<f4", 2), ("materialIndex", "<u4"), ("pad6", "<f4"),
])
assert TRI_ATTR.itemsize == 128

MATERIAL_FLOATS = 32   # BVHScene.cs kMaterialSize
LIGHT_FLOATS = 16      # PathTracer.cs LightStructSize

TLAS_NODE = np.dtype([
    ("lmin", "<f4", 3), ("left", "<u4"), ("lmax", "<f4", 3), ("right", "<u4"),
    ("rmin", "<f4", 3), ("triCount", "<u4"), ("rmax", "<f4", 3), ("firstTri", "<u4"),
])
assert TLAS_NODE.itemsize == 64

GPU_INSTANCE = np.dtype([
    ("localToWorld", "<f4", 16), ("worldToLocal", "<f4", 16),
    ("bvhOffset", "<i4"), ("triOffset", "<i4"), ("triAttributeOffset", "<i4"), ("materialIndex", "<i4"),
])
assert GPU_INSTANCE.itemsize == 144

BLAS_INSTANCE = np.dtype([
    ("localToWorld", "<f4", 16), ("worldToLocal", "<f4", 16),
    ("aabbMin", "<f4", 3), ("blasIndex", "<u4"), ("aabbMax", "<f4", 3), ("mask", "<u4"),
    ("pad", "<u4", 8),
])
assert BLAS_INSTANCE.itemsize == 192

PT_FEATURE_HAS_LIGHTS = 0x1
PT_FEATURE_HAS_TEXTURES = 0x2
PT_FEATURE_HAS_TLAS = 0x4
PT_FEATURE_HAS_ENVIRONMENT_TEXTURE = 0x8

PT_OK = 0
PT_ERR_INVALID_ARG = -1
PT_ERR_NO_DEVICE = -2
PT_ERR_HIP = -3
PT_ERR_NO_SCENE = -4
PT_ERR_UNSUPPORTED = -5


class _Sized(C.Structure):
    """PTSceneDesc / PTFrameParams start with structSize = sizeof(struct) (include/ptmi_plugin.h, "Versioning")."""
    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        self.structSize = C.sizeof(type(self))


class PTSceneDesc(_Sized):
    _fields_ = [
        ("structSize", C.c_uint32), ("_pad3", C.c_uint32),
        ("bvhNodes", C.c_void_p), ("bvhNodesBytes", C.c_uint64),
        ("bvhTris", C.c_void_p), ("bvhTrisBytes", C.c_uint64),
        ("triAttrs", C.c_void_p), ("triAttrsBytes", C.c_uint64),
        ("materials", C.c_void_p), ("materialCount", C.c_uint32), ("_pad0", C.c_uint32),
        ("lights", C.c_void_p), ("lightCount", C.c_uint32), ("_pad1", C.c_uint32),
        ("textureData", C.c_void_p), ("textureDataUints", C.c_uint64),
        ("features", C.c_uint32), ("_pad2", C.c_uint32),
        ("tlasData", C.c_void_p), ("tlasDataFloats", C.c_uint64),
        ("tlasIndexOffset", C.c_uint32), ("instanceCount", C.c_uint32),
        ("gpuInstances", C.c_void_p),
        ("envTexture", C.c_void_p), ("envWidth", C.c_uint32), ("envHeight", C.c_uint32),
    ]


class PTFrameParams(_Sized):
    _fields_ = [
        ("structSize", C.c_uint32), ("_pad0", C.c_uint32),
        ("CamInvProj", C.c_float * 16),
        ("CamToWorld", C.c_float * 16),
        ("RngSeedRoot", C.c_uint32),
        ("MaxRayBounces", C.c_uint32),
        ("SamplesPerPass", C.c_int32),
        ("OutputWidth", C.c_uint32),
        ("OutputHeight", C.c_uint32),
        ("CurrentSample", C.c_uint32),
        ("EnvironmentMode", C.c_int32),
        ("EnvironmentIntensity", C.c_float),
        ("EnvironmentColor", C.c_float * 4),
        ("EnvironmentMapRotation", C.c_float),
        ("FocalLength", C.c_float),
        ("Aperture", C.c_float),
        ("UseFireflyFilter", C.c_int32),
        ("MaxFireflyLuminance", C.c_float),
        ("UseRussianRoulette", C.c_int32),
        ("DispatchGroupsX", C.c_uint32),
        ("DispatchGroupsY", C.c_uint32),
    ]

    def copy(self):
        other = PTFrameParams()
        C.memmove(C.byref(other), C.byref(self), C.sizeof(PTFrameParams))
        return other


class PTStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "paths", "closestHitRays", "shadowRays", "nodeVisits", "triTests", "attrFetches",
        "materialFetches", "lightFetches", "texelFetches", "texDescriptorFetches",
        "pixelsWritten", "pixelsRead", "maxStackDepth", "stackOverflows", "tlasNodeVisits", "instanceVisits")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}

    @property
    def rays(self):
        return int(self.closestHitRays + self.shadowRays)

    def algorithmic_bytes(self):
        """bytes the reference kernel reads/writes for this work."""
        return int(80 * self.nodeVisits + 48 * self.triTests + 128 * self.attrFetches
                   + 128 * self.materialFetches + 64 * self.lightFetches + 4 * self.texelFetches
                   + 16 * self.texDescriptorFetches + 16 * (self.pixelsWritten + self.pixelsRead)
                   + 64 * self.tlasNodeVisits + 144 * self.instanceVisits)


class PTPresentParams(C.Structure):
    """Presentation.shader uniforms (PathTracer.cs:255-264); defaults = the inspector defaults (PathTracer.cs:41-48)."""
    _fields_ = [("OutputWidth", C.c_uint32), ("OutputHeight", C.c_uint32), ("Mode", C.c_int32), ("sRGB", C.c_int32),
                ("Exposure", C.c_float), ("Brightness", C.c_float), ("Contrast", C.c_float), ("Saturation", C.c_float),
                ("Vignette", C.c_float)]


TONEMAP_NONE, TONEMAP_ACES, TONEMAP_FILMIC, TONEMAP_REINHARD, TONEMAP_LOTTES = range(5)      # PathTracer.cs:8-14


PT_MESH_HAS_32_BIT_INDICES, PT_MESH_HAS_NORMALS, PT_MESH_HAS_TANGENTS, PT_MESH_HAS_UVS = 0x1, 0x2, 0x4, 0x8


class PTMeshDesc(C.Structure):
    """One Dispatch of MeshProcessing.compute (BVHScene.cs:489-553)."""
    _fields_ = [("vertexBuffer", C.c_void_p), ("vertexBufferBytes", C.c_uint64),
                ("indexBuffer", C.c_void_p), ("indexBufferBytes", C.c_uint64),
                ("VertexStride", C.c_uint32), ("PositionOffset", C.c_uint32), ("NormalOffset", C.c_uint32),
                ("TangentOffset", C.c_uint32), ("UVOffset", C.c_uint32), ("MaterialIndex", C.c_uint32),
                ("TriangleCount", C.c_uint32), ("OutputTriangleStart", C.c_uint32),
                ("LocalToWorld", C.c_float * 16), ("WorldToLocal", C.c_float * 16),
                ("flags", C.c_uint32), ("_pad", C.c_uint32)]


class PTTextureDesc(C.Structure):
    _fields_ = [("texels", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32), ("hasAlpha", C.c_int32), ("_pad", C.c_uint32)]


class PTTimings(C.Structure):
    _fields_ = [("passes", C.c_uint64), ("kernelMsTotal", C.c_double),
                ("kernelMsLast", C.c_double), ("kernelLaunches", C.c_uint64)]


def as_void_p(arr):
    """Borrowed host pointer of a C-contiguous numpy array (None -> NULL)."""
    if arr is None:
        return None
    assert arr.flags["C_CONTIGUOUS"]
    return arr.ctypes.data_as(C.c_void_p)
