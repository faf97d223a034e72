"""Scene ingestion: what BVHScene.cs does between "a Unity scene" and "the path tracer's buffers".

  Mesh            one Unity mesh as the reference sees it: an interleaved fp32 vertex stream (position [, normal] [, tangent]
                  [, uv]) + an optional 16/32-bit index buffer + the renderer's localToWorld and material index
  process_meshes  BVHScene.ProcessMeshes + readback (BVHScene.cs:429-560): MeshProcessing.compute on the MI355X through
                  PTProcessMeshes -> BuildBVH input (3 float4 per triangle) + TriangleAttributes
  copy_texture_data   the texture loop of BVHScene.cs:386-417 (CopyTextureData.compute) through PTCopyTextureData
  load_obj        a Wavefront OBJ reader producing `Mesh` objects (the format the public Sponza / bunny assets ship in)
  load_glb        a binary glTF 2.0 (.glb) reader: the reference loads its models through UnityGLTF (Packages/manifest.json), and
                  the one real model in its tree is Assets/Examples/Models/DamagedHelmet.glb.  Node hierarchy (matrix / TRS),
                  triangle primitives with POSITION / NORMAL / TANGENT / TEXCOORD_0 and 8/16/32-bit indices, strided and
                  normalised accessors, metallic-roughness materials, embedded PNG / JPEG images
  write_glb       the inverse for tests (a GLB the test writes itself)
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


# ---------------------------------------------------------------------------------------
# binary glTF 2.0
# ---------------------------------------------------------------------------------------
_GLTF_COMPONENT = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5125: np.uint32, 5126: np.float32}
_GLTF_WIDTH = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT4": 16}


def _glb_chunks(data):
    import json
    import struct
    if len(data) < 20 or data[:4] != b"glTF":
        raise ValueError("not a binary glTF file")
    version, length = struct.unpack_from("<II", data, 4)
    if version != 2 or length > len(data):
        raise ValueError("unsupported glTF version / truncated file")
    off, doc, blob = 12, None, b""
    while off + 8 <= length:
        clen, ctype = struct.unpack_from("<II", data, off)
        chunk = data[off + 8:off + 8 + clen]
        if ctype == 0x4E4F534A:
            doc = json.loads(chunk.decode("utf-8"))
        elif ctype == 0x004E4942 and not blob:
            blob = chunk
        off += 8 + clen + (-clen % 4)
    if doc is None:
        raise ValueError("GLB without a JSON chunk")
    return doc, blob


def _accessor(doc, blob, index):
    """One accessor as a (count, width) array of its component type converted per the glTF rules (normalised ints -> float)."""
    a = doc["accessors"][index]
    dt, width, count = np.dtype(_GLTF_COMPONENT[a["componentType"]]), _GLTF_WIDTH[a["type"]], a["count"]
    if "bufferView" not in a:
        return np.zeros((count, width), dt)
    bv = doc["bufferViews"][a["bufferView"]]
    if bv.get("buffer", 0) != 0:
        raise ValueError("only the GLB-embedded buffer is supported")
    start = bv.get("byteOffset", 0) + a.get("byteOffset", 0)
    stride = bv.get("byteStride", 0) or dt.itemsize * width
    need = start + stride * (count - 1) + dt.itemsize * width if count else start
    if need > len(blob) or start + 0 > bv.get("byteOffset", 0) + bv["byteLength"]:
        raise ValueError("accessor reaches outside the binary chunk")
    out = np.lib.stride_tricks.as_strided(np.frombuffer(blob, dt, width, start) if count else np.zeros(width, dt), shape=(count, width),
                                          strides=(stride, dt.itemsize), writeable=False).copy() if count else np.zeros((0, width), dt)
    if a.get("normalized") and dt.kind in "iu":
        scale = float(np.iinfo(dt).max)
        out = np.maximum(out.astype(np.float32) / scale, -1.0) if dt.kind == "i" else out.astype(np.float32) / scale
    return out


def _node_matrix(node):
    if "matrix" in node:
        return np.asarray(node["matrix"], np.float64).reshape(4, 4).T          # glTF stores matrices column-major
    t = np.asarray(node.get("translation", (0, 0, 0)), np.float64)
    x, y, z, w = node.get("rotation", (0, 0, 0, 1))
    sc = np.asarray(node.get("scale", (1, 1, 1)), np.float64)
    r = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]], np.float64)
    m = np.eye(4)
    m[:3, :3] = r * sc[None, :]
    m[:3, 3] = t
    return m


def load_glb(path, unity_handedness=True, load_images=True, flip_axis=2, flip_v=False):
    """Reads a .glb into (meshes, materials, images):
      meshes     one `Mesh` per triangle primitive of every node of the default scene, local_to_world = the node's world matrix
      materials  one dict per glTF material: base_color (linear rgba), metallic, roughness, emissive, alpha_mode (0 opaque,
                 1 mask, 2 blend), alpha_cutoff, tex_base / tex_mr / tex_emission / tex_occlusion / tex_normal = image index or -1
      images     (h, w, 4) float32 arrays in [0, 1] (PNG / JPEG decoded with Pillow when it is installed, else None entries)
    unity_handedness: glTF is right-handed, Unity left-handed, so the importer mirrors ONE axis of positions / normals / tangents
    (and of the node transforms) and reverses the winding; BVHScene then sees the mirrored meshes.
    PARITY UNPINNED -- an assumption, not a fact about the reference: its importer is the package org.khronos.unitygltf 2.14.1
    (Packages/manifest.json), whose source is not in the reference tree and cannot be run here.  `flip_axis` (0 = x, 1 = y,
    2 = z) selects the mirrored axis: 2 is what early UnityGLTF releases did (CoordinateSpaceConversionScale (1, 1, -1)); later
    2.x releases are believed to mirror x instead ((-1, 1, 1)), which differs from the z flip by a half turn about y -- the model
    faces the other way for the same camera.  The same releases also negate tangent.w and flip the V texture coordinate
    (v -> 1 - v, with images stored bottom row first); `flip_v` applies the latter, tangent.w is not carried at all (the hot path
    reads no tangents: the normal-map code of util/material.hlsl:114-133 is commented out).  Neither choice changes what the
    kernels compute for given buffers.  Primitive modes other than TRIANGLES raise."""
    with open(path, "rb") as f:
        doc, blob = _glb_chunks(f.read())
    if flip_axis not in (0, 1, 2):
        raise ValueError("flip_axis must be 0, 1 or 2")
    mirror = np.ones(3, np.float32)
    mirror[flip_axis] = -1.0
    flip = np.diag([float(mirror[0]), float(mirror[1]), float(mirror[2]), 1.0]) if unity_handedness else np.eye(4)
    meshes = []

    def visit(index, parent):
        node = doc["nodes"][index]
        world = parent @ _node_matrix(node)
        if "mesh" in node:
            for prim in doc["meshes"][node["mesh"]]["primitives"]:
                if prim.get("mode", 4) != 4:
                    raise ValueError("only TRIANGLES primitives are supported")
                att = prim["attributes"]
                pos = _accessor(doc, blob, att["POSITION"]).astype(np.float32)
                nrm = _accessor(doc, blob, att["NORMAL"]).astype(np.float32) if "NORMAL" in att else None
                tan = _accessor(doc, blob, att["TANGENT"]).astype(np.float32)[:, :3] if "TANGENT" in att else None
                uv = _accessor(doc, blob, att["TEXCOORD_0"]).astype(np.float32) if "TEXCOORD_0" in att else None
                idx = _accessor(doc, blob, prim["indices"]).reshape(-1) if "indices" in prim else None
                if idx is not None and (idx.size % 3 or (idx.size and int(idx.max()) >= len(pos))):
                    raise ValueError("index accessor does not describe triangles of this primitive")
                if unity_handedness:
                    pos = pos * mirror
                    nrm = nrm * mirror if nrm is not None else None
                    tan = tan * mirror if tan is not None else None
                    if idx is not None:
                        idx = idx.reshape(-1, 3)[:, ::-1].reshape(-1)
                    else:
                        order = np.arange(len(pos)).reshape(-1, 3)[:, ::-1].reshape(-1)
                        pos, nrm, tan, uv = pos[order], (nrm[order] if nrm is not None else None), (tan[order] if tan is not None else None), (uv[order] if uv is not None else None)
                if flip_v and uv is not None:
                    uv = uv * np.array([1, -1], np.float32) + np.array([0, 1], np.float32)
                if idx is not None:
                    idx = idx.astype(np.uint16 if len(pos) < 65536 else np.uint32)
                meshes.append(Mesh(positions=np.ascontiguousarray(pos), normals=nrm, tangents=tan, uvs=uv, indices=idx,
                                   local_to_world=flip @ world @ flip, material_index=prim.get("material", 0), name=node.get("name", "")))
        for child in node.get("children", []):
            visit(child, world)

    scene_nodes = doc["scenes"][doc.get("scene", 0)]["nodes"] if doc.get("scenes") else list(range(len(doc.get("nodes", []))))
    for n in scene_nodes:
        visit(n, np.eye(4))

    def tex_image(ref):
        if ref is None:
            return -1
        return doc["textures"][ref["index"]].get("source", -1)
    materials = []
    for m in doc.get("materials", []):
        pbr = m.get("pbrMetallicRoughness", {})
        materials.append(dict(
            name=m.get("name", ""), base_color=tuple(pbr.get("baseColorFactor", (1, 1, 1, 1))), metallic=float(pbr.get("metallicFactor", 1.0)),
            roughness=float(pbr.get("roughnessFactor", 1.0)), emissive=tuple(m.get("emissiveFactor", (0, 0, 0))),
            alpha_mode={"OPAQUE": 0, "MASK": 1, "BLEND": 2}[m.get("alphaMode", "OPAQUE")], alpha_cutoff=float(m.get("alphaCutoff", 0.5)),
            tex_base=tex_image(pbr.get("baseColorTexture")), tex_mr=tex_image(pbr.get("metallicRoughnessTexture")),
            tex_emission=tex_image(m.get("emissiveTexture")), tex_occlusion=tex_image(m.get("occlusionTexture")), tex_normal=tex_image(m.get("normalTexture"))))
    images = []
    for im in doc.get("images", []) if load_images else []:
        arr = None
        if "bufferView" in im:
            bv = doc["bufferViews"][im["bufferView"]]
            raw = blob[bv.get("byteOffset", 0):bv.get("byteOffset", 0) + bv["byteLength"]]
            try:
                import io
                from PIL import Image
                arr = np.asarray(Image.open(io.BytesIO(raw)).convert("RGBA"), dtype=np.float32) / 255.0
            except ImportError:
                arr = None
        images.append(arr)
    return meshes, materials, images


def pack_gltf_materials(materials, image_to_texture=None):
    """glTF material dicts (load_glb) -> (M, 32) MaterialData rows with the packing rules of BVHScene.cs:236-282.  glTF factors are
    linear while BVHScene applies pow(c, 2.2) to a Unity (gamma) colour, so the factor goes in as c^(1/2.2).  image_to_texture
    maps an image index to its slot in the packed TextureData (None: the images' own order)."""
    rows = []
    for m in materials:
        def slot(i):
            return -1 if i is None or i < 0 else (image_to_texture[i] if image_to_texture is not None else i)
        r, g, b, a = m["base_color"]
        rows.append(scenes.pack_material(color=(r ** (1 / 2.2), g ** (1 / 2.2), b ** (1 / 2.2), a), metallic=m["metallic"], roughness=m["roughness"],
                                         emission=m["emissive"], alpha_mode=m["alpha_mode"], alpha_cutoff=m["alpha_cutoff"],
                                         tex_base=slot(m["tex_base"]), tex_mr=slot(m["tex_mr"]), tex_emission=slot(m["tex_emission"]),
                                         tex_occlusion=slot(m["tex_occlusion"]), tex_normal=slot(m["tex_normal"])))
    return np.stack(rows) if rows else np.stack([scenes.pack_material()])


def write_glb(path, meshes, materials=None, node_matrices=None, interleave=False):
    """A small GLB writer for tests: one node + mesh + primitive per `Mesh` (positions, optional normals / uvs, optional indices),
    float32 attributes, optionally interleaved into one strided bufferView; `node_matrices` (4x4 each) become node matrices."""
    import json
    import struct
    blob, views, accessors, gl_meshes, nodes = bytearray(), [], [], [], []

    def add_view(raw, stride=None):
        while len(blob) % 4:
            blob.append(0)
        v = {"buffer": 0, "byteOffset": len(blob), "byteLength": len(raw)}
        if stride:
            v["byteStride"] = stride
        blob.extend(raw)
        views.append(v)
        return len(views) - 1

    for k, m in enumerate(meshes):
        att = {}
        cols = [("POSITION", np.asarray(m.positions, np.float32), "VEC3")]
        if m.normals is not None:
            cols.append(("NORMAL", np.asarray(m.normals, np.float32), "VEC3"))
        if m.uvs is not None:
            cols.append(("TEXCOORD_0", np.asarray(m.uvs, np.float32), "VEC2"))
        if interleave:
            packed = np.ascontiguousarray(np.concatenate([c[1] for c in cols], axis=1), np.float32)
            view, off = add_view(packed.tobytes(), stride=packed.shape[1] * 4), 0
            for name, arr, typ in cols:
                acc = {"bufferView": view, "byteOffset": off, "componentType": 5126, "count": len(arr), "type": typ}
                if name == "POSITION":
                    acc["min"], acc["max"] = arr.min(axis=0).tolist(), arr.max(axis=0).tolist()
                accessors.append(acc)
                att[name] = len(accessors) - 1
                off += arr.shape[1] * 4
        else:
            for name, arr, typ in cols:
                acc = {"bufferView": add_view(np.ascontiguousarray(arr).tobytes()), "componentType": 5126, "count": len(arr), "type": typ}
                if name == "POSITION":
                    acc["min"], acc["max"] = arr.min(axis=0).tolist(), arr.max(axis=0).tolist()
                accessors.append(acc)
                att[name] = len(accessors) - 1
        prim = {"attributes": att, "material": int(m.material_index)}
        if m.indices is not None:
            idx = np.ascontiguousarray(m.indices)
            ct = {np.dtype(np.uint8): 5121, np.dtype(np.uint16): 5123, np.dtype(np.uint32): 5125}[idx.dtype]
            accessors.append({"bufferView": add_view(idx.tobytes()), "componentType": ct, "count": int(idx.size), "type": "SCALAR"})
            prim["indices"] = len(accessors) - 1
        gl_meshes.append({"primitives": [prim]})
        node = {"mesh": k, "name": m.name or f"node{k}"}
        if node_matrices is not None:
            node["matrix"] = np.asarray(node_matrices[k], np.float64).T.reshape(16).tolist()
        nodes.append(node)
    doc = {"asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": list(range(len(nodes)))}], "nodes": nodes, "meshes": gl_meshes,
           "accessors": accessors, "bufferViews": views, "buffers": [{"byteLength": len(blob)}]}
    if materials:
        doc["materials"] = materials
    js = json.dumps(doc).encode("utf-8")
    js += b" " * (-len(js) % 4)
    while len(blob) % 4:
        blob.append(0)
    total = 12 + 8 + len(js) + 8 + len(blob)
    with open(path, "wb") as f:
        f.write(struct.pack("<4sII", b"glTF", 2, total))
        f.write(struct.pack("<II", len(js), 0x4E4F534A) + js)
        f.write(struct.pack("<II", len(blob), 0x004E4942) + bytes(blob))


def scene_from_meshes(ctx, meshes, materials, camera, images=None, lights=None, **scene_kw) -> scenes.Scene:
    """meshes -> Scene, every buffer produced by the ingestion kernels."""
    verts, attrs = process_meshes(ctx, meshes)
    tex = copy_texture_data(ctx, images) if images else np.zeros(0, dtype=np.uint32)
    lights = np.zeros((0, 16), np.float32) if lights is None else lights
    return scenes.Scene(scene_kw.pop("name", "ingested"), verts, attrs, np.asarray(materials, np.float32), lights, tex, camera, **scene_kw)
