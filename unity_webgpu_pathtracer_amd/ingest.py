"""Scene ingestion: what BVHScene.cs does between "a Unity scene" and "the path tracer's buffers".

  Mesh            one Unity mesh as the reference sees it: an interleaved fp32 vertex stream (position [, normal] [, tangent]
                  [, uv]) + an optional 16/32-bit index buffer + the renderer's localToWorld and material index
  process_meshes  BVHScene.ProcessMeshes + readback (BVHScene.cs:429-560): MeshProcessing.compute on the MI355X through
                  PTProcessMeshes -> BuildBVH input (3 float4 per triangle) + TriangleAttributes
  copy_texture_data   the texture loop of BVHScene.cs:386-417 (CopyTextureData.compute) through PTCopyTextureData
  load_obj        a Wavefront OBJ reader producing `Mesh` objects (the reference itself loads glTF through UnityGLTF, a Unity
                  package; OBJ is the format the public Sponza / bunny assets ship in)
  scene_from_meshes   the whole ingestion: meshes + materials + textures -> scenes.Scene

Nothing here computes on the host: transforms, normal matrices, index decoding and RGBA8 packing run in the HIP kernels.
"""
import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import abi, plugin, scenes


@dataclass
class Mesh:
    positions: np.ndarray                       # (V, 3) float32
    normals: np.ndarray = None                  # (V, 3) or None
    tangents: np.ndarray = None                 # (V, 3) or None  (Unity stores float4; xyz is what the shader reads)
    uvs: np.ndarray = None                      # (V, 2) or None
    indices: np.ndarray = None                  # (T*3,) uint16 / uint32, or None (non-indexed: 3 consecutive vertices per triangle)
    local_to_world: np.ndarray = field(default_factory=lambda: np.eye(4, dtype=np.float64))
    material_index: int = 0
    name: str = ""

    @property
    def triangle_count(self):                   # Utilities.GetTriangleCount
        return (len(self.indices) if self.indices is not None else len(self.positions)) // 3

    def vertex_stream(self):
        """Interleaves the attributes the way a Unity vertex stream 0 is laid out; returns (bytes array, stride, offsets)."""
        cols, offsets, off = [], {}, 0
        for key, arr, n in (("position", self.positions, 3), ("normal", self.normals, 3), ("tangent", self.tangents, 4), ("uv", self.uvs, 2)):
            if arr is None:
                offsets[key] = 0
                continue
            a = np.asarray(arr, dtype=np.float32)
            if key == "tangent" and a.shape[1] == 3:
                a = np.concatenate([a, np.ones((a.shape[0], 1), np.float32)], axis=1)      # w = handedness
            cols.append(a[:, :n])
            offsets[key] = off
            off += 4 * n
        vb = np.ascontiguousarray(np.concatenate(cols, axis=1), dtype=np.float32)
        return vb.view(np.uint8).reshape(-1), off, offsets


def _unity_matrix(m):
    """(4, 4) row-major math matrix -> 16 floats in Unity Matrix4x4 memory order (column-major)."""
    return np.asarray(m, dtype=np.float64).T.reshape(16).astype(np.float32)


def mesh_descs(meshes):
    """The per-mesh uniforms of BVHScene.cs:520-548.  Returns (ctypes array of PTMeshDesc, total triangles, keep-alive list)."""
    arr = (abi.PTMeshDesc * len(meshes))()
    keep, start = [], 0
    for i, m in enumerate(meshes):
        vb, stride, offs = m.vertex_stream()
        keep.append(vb)
        d = arr[i]
        d.vertexBuffer, d.vertexBufferBytes = vb.ctypes.data, vb.nbytes
        flags = 0
        if m.indices is not None:
            ib = np.ascontiguousarray(m.indices)
            assert ib.dtype in (np.uint16, np.uint32)
            keep.append(ib)
            d.indexBuffer, d.indexBufferBytes = ib.ctypes.data, ib.nbytes
            if ib.dtype == np.uint32:
                flags |= abi.PT_MESH_HAS_32_BIT_INDICES
        if m.normals is not None:
            flags |= abi.PT_MESH_HAS_NORMALS
        if m.tangents is not None:
            flags |= abi.PT_MESH_HAS_TANGENTS
        if m.uvs is not None:
            flags |= abi.PT_MESH_HAS_UVS
        d.VertexStride, d.PositionOffset, d.NormalOffset = stride, offs["position"], offs["normal"]
        d.TangentOffset, d.UVOffset = offs["tangent"], offs["uv"]
        d.MaterialIndex, d.TriangleCount, d.OutputTriangleStart = m.material_index, m.triangle_count, start
        l2w = np.asarray(m.local_to_world, dtype=np.float64)
        d.LocalToWorld[:] = _unity_matrix(l2w).tolist()
        d.WorldToLocal[:] = _unity_matrix(np.linalg.inv(l2w)).tolist()              # renderer.worldToLocalMatrix
        d.flags = flags
        start += m.triangle_count
    return arr, start, keep


def process_meshes(ctx, meshes):
    """BVHScene.ProcessMeshes on the GPU.  Returns (vertices (T*3, 4) float32 w = 0, tri_attrs (T,) abi.TRI_ATTR)."""
    arr, total, keep = mesh_descs(meshes)
    pos = np.zeros((total * 3, 4), dtype=np.float32)
    attrs = np.zeros(total, dtype=abi.TRI_ATTR)
    plugin.check(plugin.load_library().PTProcessMeshes(ctx, arr, len(meshes), total, pos.ctypes.data_as(C.c_void_p),
                                                       attrs.ctypes.data_as(C.c_void_p)))
    del keep
    return pos, attrs


def copy_texture_data(ctx, images):
    """images: list of ((h, w, 4) float32 in [0, 1], has_alpha).  Returns TextureData uint32 (descriptors, then RGBA8 texels)."""
    arr = (abi.PTTextureDesc * len(images))()
    keep, total = [], 4 * len(images)
    for i, (img, has_alpha) in enumerate(images):
        a = np.ascontiguousarray(img, dtype=np.float32)
        keep.append(a)
        arr[i].texels, arr[i].height, arr[i].width, arr[i].hasAlpha = a.ctypes.data, a.shape[0], a.shape[1], 1 if has_alpha else 0
        total += a.shape[0] * a.shape[1]
    out = np.zeros(total, dtype=np.uint32)
    plugin.check(plugin.load_library().PTCopyTextureData(ctx, arr, len(images), out.ctypes.data_as(C.c_void_p), out.size))
    return out


# ---------------------------------------------------------------------------------------
# Wavefront OBJ
# ---------------------------------------------------------------------------------------
def load_obj(path, index_dtype=None):
    """Reads v / vn / vt / f / usemtl / o / g.  Polygons are fanned into triangles; every distinct (v, vt, vn) triple becomes
    one vertex (Unity's importer does the same split).  Returns (list of Mesh, one per material group, list of material names)."""
    v, vn, vt = [], [], []
    groups, order, current = {}, [], "default"
    with open(path) as f:
        for line in f:
            p = line.split()
            if not p or p[0].startswith("#"):
                continue
            if p[0] == "v":
                v.append([float(x) for x in p[1:4]])
            elif p[0] == "vn":
                vn.append([float(x) for x in p[1:4]])
            elif p[0] == "vt":
                vt.append([float(x) for x in p[1:3]])
            elif p[0] == "usemtl":
                current = p[1] if len(p) > 1 else "default"
            elif p[0] == "f":
                if current not in groups:
                    groups[current] = []
                    order.append(current)
                corners = []
                for c in p[1:]:
                    a = (c.split("/") + ["", ""])[:3]
                    iv = int(a[0])
                    it = int(a[1]) if a[1] else 0
                    inn = int(a[2]) if a[2] else 0
                    corners.append((iv - 1 if iv > 0 else len(v) + iv,
                                    (it - 1 if it > 0 else len(vt) + it) if it else -1,
                                    (inn - 1 if inn > 0 else len(vn) + inn) if inn else -1))
                for k in range(1, len(corners) - 1):
                    groups[current].append((corners[0], corners[k], corners[k + 1]))
    v, vn, vt = np.asarray(v, np.float32).reshape(-1, 3), np.asarray(vn, np.float32).reshape(-1, 3), np.asarray(vt, np.float32).reshape(-1, 2)
    meshes = []
    for mi, name in enumerate(order):
        remap, pos, nrm, uv, idx = {}, [], [], [], []
        has_n = all(c[2] >= 0 for tri in groups[name] for c in tri)
        has_t = all(c[1] >= 0 for tri in groups[name] for c in tri)
        for tri in groups[name]:
            for c in tri:
                if c not in remap:
                    remap[c] = len(pos)
                    pos.append(v[c[0]])
                    if has_n:
                        nrm.append(vn[c[2]])
                    if has_t:
                        uv.append(vt[c[1]])
                idx.append(remap[c])
        dt = index_dtype or (np.uint16 if len(pos) < 65536 else np.uint32)
        meshes.append(Mesh(positions=np.asarray(pos, np.float32), normals=np.asarray(nrm, np.float32) if has_n else None,
                           uvs=np.asarray(uv, np.float32) if has_t else None, indices=np.asarray(idx, dtype=dt),
                           material_index=mi, name=name))
    return meshes, order


def scene_from_meshes(ctx, meshes, materials, camera, images=None, lights=None, **scene_kw) -> scenes.Scene:
    """meshes -> Scene, every buffer produced by the ingestion kernels."""
    verts, attrs = process_meshes(ctx, meshes)
    tex = copy_texture_data(ctx, images) if images else np.zeros(0, dtype=np.uint32)
    lights = np.zeros((0, 16), np.float32) if lights is None else lights
    return scenes.Scene(scene_kw.pop("name", "ingested"), verts, attrs, np.asarray(materials, np.float32), lights, tex, camera, **scene_kw)
