"""Synthetic scene generators that emit the reference's exact buffer layouts.

No real scene assets exist in the reference snapshot (all .unity/.obj/.fbx/.png are git-LFS
stubs), so every benchmark / parity scene is generated here, seeded and CPU-side.  The packing
rules restate what the reference's scene packer does (the C# itself cannot run here):

  * world-space float4 positions, w = 0, three per triangle   MeshProcessing.compute:100-102
  * 128-byte TriangleAttributes (normals, tangents, uvs, materialIndex)  MeshProcessing.compute:104-138
  * 32-float MaterialData, sRGB->linear pow(c, 2.2) on base colour, opacity = a*(1-transmission),
    roughness = 1 - _Glossiness, default ior 1.1, texture slots -1 when absent   BVHScene.cs:178-282
  * 16-float Light records, rectangle: position -= (u+v)/2, u = right*sizeX, v = up*sizeY,
    type = Unity LightType as int bits                                          PathTracer.cs:324-348,410-460
  * TextureData: K x {w, h, offset, 0} then RGBA8 texels (a<<24|b<<16|g<<8|r)   CopyTextureData.compute:21-35
  * camera: Unity cameraToWorldMatrix (GL convention, camera looks down -Z) and
    projectionMatrix.inverse, 16 floats in Matrix4x4 memory order (column-major) PathTracer.cs:230-231

The scenes are the configs of BASELINE.json: cornell (C1/C2), bunny (C3, "bunny-class" displaced
icosphere in the Cornell box), sponza (C4/C5, "Sponza-class" procedural atrium), plus `zoo`, a small
scene that exercises every material lobe, light type, alpha mode, texture slot and the thin lens.
"""
from dataclasses import dataclass, field

import numpy as np

from . import abi

F32 = np.float32


# ---------------------------------------------------------------------------------------
# containers
# ---------------------------------------------------------------------------------------
@dataclass
class Camera:
    eye: tuple
    target: tuple
    up: tuple = (0.0, 1.0, 0.0)
    vfov_deg: float = 40.0
    near: float = 0.1
    far: float = 100.0
    aperture: float = 0.0
    focal_length: float = 0.0


@dataclass
class Scene:
    name: str
    vertices: np.ndarray          # (T*3, 4) float32, w = 0        -> BuildBVH input
    tri_attrs: np.ndarray         # (T,) abi.TRI_ATTR
    materials: np.ndarray         # (M, 32) float32
    lights: np.ndarray            # (L, 16) float32
    texture_data: np.ndarray      # (n,) uint32 (descriptors + texels); empty when no textures
    camera: Camera
    environment_mode: int = 0
    environment_color: tuple = (0.0, 0.0, 0.0, 1.0)
    environment_intensity: float = 1.0
    extras: dict = field(default_factory=dict)
    # HAS_TLAS scenes (BVHScene.cs:600-758): `vertices` / `tri_attrs` hold the unique meshes back to back in LOCAL space,
    # mesh_ranges[i] = (first triangle, triangle count) of mesh i, instances[j] = (mesh index, localToWorld 4x4, material index)
    mesh_ranges: list = None
    instances: list = None
    # HAS_ENVIRONMENT_TEXTURE (PathTracer.cs:32,119-137): (H, W, 4) float32, rows in readback order; None = colour / basic sky
    environment_texture: np.ndarray = None
    environment_map_rotation: float = 0.0

    @property
    def use_tlas(self):
        return self.instances is not None

    @property
    def tri_count(self):
        return self.vertices.shape[0] // 3

    @property
    def features(self):
        f = 0
        if len(self.lights) > 0:
            f |= abi.PT_FEATURE_HAS_LIGHTS
        if self.texture_data.size > 0:
            f |= abi.PT_FEATURE_HAS_TEXTURES
        if self.use_tlas:
            f |= abi.PT_FEATURE_HAS_TLAS
        if self.environment_texture is not None:
            f |= abi.PT_FEATURE_HAS_ENVIRONMENT_TEXTURE
        return f


# ---------------------------------------------------------------------------------------
# camera matrices (Unity conventions)
# ---------------------------------------------------------------------------------------
def _normalize(v):
    v = np.asarray(v, dtype=np.float64)
    return v / np.linalg.norm(v)


def camera_matrices(cam: Camera, aspect: float):
    """Returns (CamInvProj[16], CamToWorld[16]) float32 in Unity Matrix4x4 memory order
    (element (r, c) at index c*4 + r)."""
    eye = np.asarray(cam.eye, dtype=np.float64)
    fwd = _normalize(np.asarray(cam.target, dtype=np.float64) - eye)
    # Unity is left-handed: right = up x forward
    right = _normalize(np.cross(np.asarray(cam.up, dtype=np.float64), fwd))
    up = np.cross(fwd, right)
    c2w = np.eye(4)
    c2w[:3, 0] = right
    c2w[:3, 1] = up
    c2w[:3, 2] = -fwd          # cameraToWorldMatrix: camera space looks down -Z (GL convention)
    c2w[:3, 3] = eye
    t = np.tan(np.radians(cam.vfov_deg) * 0.5)
    n, f = cam.near, cam.far
    proj = np.zeros((4, 4))
    proj[0, 0] = 1.0 / (t * aspect)
    proj[1, 1] = 1.0 / t
    proj[2, 2] = -(f + n) / (f - n)
    proj[2, 3] = -2.0 * f * n / (f - n)
    proj[3, 2] = -1.0
    inv = np.linalg.inv(proj)
    return (np.ascontiguousarray(inv.T.reshape(16), dtype=F32),
            np.ascontiguousarray(c2w.T.reshape(16), dtype=F32))


def frame_params(scene: Scene, width: int, height: int, spp: int = 1, current_sample: int = 0,
                 seed: int = 0x12345678, max_bounces: int = 4, russian_roulette: bool = True,
                 firefly: bool = False, max_firefly_luminance: float = 10.0,
                 dispatch_groups=(0, 0)) -> abi.PTFrameParams:
    """The uniform block of PathTracer.cs:230-249 for one pass."""
    p = abi.PTFrameParams()
    inv, c2w = camera_matrices(scene.camera, width / height)
    p.CamInvProj[:] = inv.tolist()
    p.CamToWorld[:] = c2w.tolist()
    p.RngSeedRoot = seed & 0xFFFFFFFF
    p.MaxRayBounces = max(max_bounces, 1)          # PathTracer.cs:234
    p.SamplesPerPass = max(1, spp)                 # PathTracer.cs:235
    p.OutputWidth = width
    p.OutputHeight = height
    p.CurrentSample = current_sample
    p.EnvironmentMode = scene.environment_mode
    p.EnvironmentIntensity = scene.environment_intensity
    p.EnvironmentColor[:] = [float(c) for c in scene.environment_color]
    p.EnvironmentMapRotation = scene.environment_map_rotation
    p.FocalLength = scene.camera.focal_length
    p.Aperture = scene.camera.aperture
    p.UseFireflyFilter = 1 if firefly else 0
    p.MaxFireflyLuminance = max_firefly_luminance
    p.UseRussianRoulette = 1 if russian_roulette else 0
    p.DispatchGroupsX, p.DispatchGroupsY = dispatch_groups
    return p


# ---------------------------------------------------------------------------------------
# material / light / texture packing
# ---------------------------------------------------------------------------------------
def pack_material(color=(0.8, 0.8, 0.8, 1.0), transmission=0.0, emission=(0.0, 0.0, 0.0), metallic=0.0,
                  glossiness=None, roughness=0.0, ior=1.1, normal_scale=1.0, alpha_mode=0, alpha_cutoff=0.5,
                  anisotropic=0.0, specular=0.0, specular_tint=0.0, sheen=0.0, sheen_tint=0.0,
                  subsurface=0.0, clearcoat=0.0, clearcoat_gloss=0.0,
                  tex_base=-1, tex_mr=-1, tex_normal=-1, tex_emission=-1, tex_occlusion=-1,
                  uv_scale=(1.0, 1.0), uv_offset=(0.0, 0.0)):
    """BVHScene.cs:236-282.  `glossiness` (Unity Standard) overrides roughness = 1 - glossiness."""
    m = np.zeros(32, dtype=F32)
    r, g, b, a = [F32(c) for c in color]
    opacity = F32(a) * (F32(1.0) - F32(transmission))
    m[0:3] = np.power(np.array([r, g, b], dtype=F32), F32(2.2))     # Mathf.Pow(c, 2.2f)
    m[3] = opacity
    m[4:7] = emission
    m[7] = alpha_cutoff
    m[8] = metallic
    m[9] = (1.0 - glossiness) if glossiness is not None else roughness
    m[10] = normal_scale
    m[11] = ior
    m[12] = float(alpha_mode)
    m[13] = anisotropic
    m[14] = specular
    m[15] = specular_tint
    m[16] = sheen
    m[17] = sheen_tint
    m[18] = subsurface
    m[19] = clearcoat
    m[20] = clearcoat_gloss
    m[21] = F32(1.0) - opacity
    m[22] = tex_base
    m[23] = tex_mr
    m[24] = tex_normal
    m[25] = tex_emission
    m[26] = tex_occlusion
    m[27] = -1.0
    m[28:30] = uv_scale
    m[30:32] = uv_offset
    return m


def _type_bits(t):
    return np.array([t], dtype=np.uint32).view(F32)[0]


def pack_rect_light(center, right, up, size, color, intensity=1.0, rng=100.0):
    """PathTracer.cs:324-348 (GetLightData, area light branch) + :410-460."""
    l = np.zeros(16, dtype=F32)
    u = np.asarray(right, dtype=F32) * F32(size[0])
    v = np.asarray(up, dtype=F32) * F32(size[1])
    pos = np.asarray(center, dtype=F32) - (u + v) * F32(0.5)
    l[0:3] = pos
    l[3] = _type_bits(3)
    l[4:7] = np.asarray(color, dtype=F32) * F32(intensity)
    l[7] = rng
    l[8:11] = u
    l[11] = F32(size[0]) * F32(size[1])
    l[12:15] = v
    return l


def pack_point_light(position, color, intensity=1.0, rng=10.0):
    l = np.zeros(16, dtype=F32)
    l[0:3] = position
    l[3] = _type_bits(2)
    l[4:7] = np.asarray(color, dtype=F32) * F32(intensity)
    l[7] = rng
    # GetLightData's else-branch also runs for point lights: u = right*areaSize.x, v = up*areaSize.y
    # with areaSize = 0 for a point light, so u = v = 0 and area = 0 (PathTracer.cs:343-348).
    return l


def pack_spot_light(position, forward, spot_angle_deg, inner_angle_deg, color, intensity=1.0, rng=10.0):
    l = np.zeros(16, dtype=F32)
    l[0:3] = position
    l[3] = _type_bits(0)
    l[4:7] = np.asarray(color, dtype=F32) * F32(intensity)
    l[7] = rng
    l[8:11] = _normalize(forward).astype(F32)
    l[12] = np.cos(np.radians(spot_angle_deg) * 0.5)
    l[13] = np.cos(np.radians(inner_angle_deg) * 0.5)
    return l


def pack_textures(images):
    """images: list of (h, w, 4) float32 in [0,1] -> TextureData uint32 (CopyTextureData.compute:21-35)."""
    if not images:
        return np.zeros(0, dtype=np.uint32)
    k = len(images)
    offset = 4 * k
    desc = []
    chunks = []
    for img in images:
        h, w, _ = img.shape
        desc += [w, h, offset, 0]
        q = (img.astype(F32) * F32(255.0)).astype(np.uint32) & 0xFF      # (uint)(texel * 255.0f) & 0xff
        texel = (q[..., 3] << 24) | (q[..., 2] << 16) | (q[..., 1] << 8) | q[..., 0]
        chunks.append(texel.reshape(-1).astype(np.uint32))
        offset += w * h
    return np.concatenate([np.array(desc, dtype=np.uint32)] + chunks)


# ---------------------------------------------------------------------------------------
# procedural helpers
# ---------------------------------------------------------------------------------------
def _hash_u32(x):
    x = x.astype(np.uint32)
    x ^= x >> np.uint32(16)
    x = x * np.uint32(0x7FEB352D)
    x ^= x >> np.uint32(15)
    x = x * np.uint32(0x846CA68B)
    x ^= x >> np.uint32(16)
    return x


def _lattice(ix, iy, iz, seed):
    h = _hash_u32(ix.astype(np.uint32) * np.uint32(73856093) ^ iy.astype(np.uint32) * np.uint32(19349663)
                  ^ iz.astype(np.uint32) * np.uint32(83492791) ^ np.uint32(seed * 2654435761 & 0xFFFFFFFF))
    return h.astype(np.float64) / 4294967295.0


def value_noise3(p, seed=7, octaves=3, frequency=2.0):
    """Smooth lattice value noise in [-1, 1], summed over octaves (amplitude halves)."""
    p = np.asarray(p, dtype=np.float64)
    total = np.zeros(p.shape[:-1])
    amp, freq, norm = 1.0, frequency, 0.0
    for o in range(octaves):
        q = p * freq + 19.19 * o
        i = np.floor(q).astype(np.int64)
        f = q - i
        s = f * f * (3.0 - 2.0 * f)
        acc = 0.0
        for dx in (0, 1):
            wx = s[..., 0] if dx else 1.0 - s[..., 0]
            for dy in (0, 1):
                wy = s[..., 1] if dy else 1.0 - s[..., 1]
                for dz in (0, 1):
                    wz = s[..., 2] if dz else 1.0 - s[..., 2]
                    acc = acc + wx * wy * wz * _lattice(i[..., 0] + dx, i[..., 1] + dy, i[..., 2] + dz, seed + o)
        total += amp * (acc * 2.0 - 1.0)
        norm += amp
        amp *= 0.5
        freq *= 2.0
    return total / norm


class SoupBuilder:
    """Accumulates a world-space triangle soup with per-vertex normal / tangent / uv and a material id."""

    def __init__(self):
        self.P, self.N, self.T, self.UV, self.M = [], [], [], [], []

    def add(self, P, N, T, UV, material):
        P = np.asarray(P, dtype=np.float64).reshape(-1, 3, 3)
        n = P.shape[0]
        self.P.append(P)
        self.N.append(np.asarray(N, dtype=np.float64).reshape(n, 3, 3))
        self.T.append(np.asarray(T, dtype=np.float64).reshape(n, 3, 3))
        self.UV.append(np.asarray(UV, dtype=np.float64).reshape(n, 3, 2))
        self.M.append(np.full(n, material, dtype=np.uint32))

    def grid(self, fn, nu, nv, material):
        """Tessellate a parametric surface fn(u, v) -> (pos, normal, tangent, uv) over [0,1]^2."""
        u = np.linspace(0.0, 1.0, nu + 1)
        v = np.linspace(0.0, 1.0, nv + 1)
        U, V = np.meshgrid(u, v, indexing="ij")
        pos, nrm, tan, uv = fn(U, V)
        idx = lambda a, i, j: a[i:i + nu, j:j + nv]

        def corner(a, di, dj):
            return a[di:di + nu, dj:dj + nv].reshape(-1, a.shape[-1])
        tris = [((0, 0), (1, 0), (1, 1)), ((0, 0), (1, 1), (0, 1))]
        for tri in tris:
            P = np.stack([corner(pos, *c) for c in tri], axis=1)
            N = np.stack([corner(nrm, *c) for c in tri], axis=1)
            T = np.stack([corner(tan, *c) for c in tri], axis=1)
            UV = np.stack([corner(uv, *c) for c in tri], axis=1)
            self.add(P, N, T, UV, material)

    def quad(self, origin, eu, ev, normal, nu, nv, material, uv_scale=(1.0, 1.0)):
        origin, eu, ev = (np.asarray(a, dtype=np.float64) for a in (origin, eu, ev))
        normal = _normalize(normal)
        tangent = _normalize(eu)

        def fn(U, V):
            pos = origin + U[..., None] * eu + V[..., None] * ev
            nrm = np.broadcast_to(normal, pos.shape)
            tan = np.broadcast_to(tangent, pos.shape)
            uv = np.stack([U * uv_scale[0], V * uv_scale[1]], axis=-1)
            return pos, nrm, tan, uv
        self.grid(fn, nu, nv, material)

    def cylinder(self, base, radius, height, segments, rings, material, uv_scale=(1.0, 1.0), flare=0.0):
        base = np.asarray(base, dtype=np.float64)

        def fn(U, V):
            ang = U * 2.0 * np.pi
            r = radius * (1.0 + flare * (np.cos(V * 2.0 * np.pi) * 0.5 + 0.5) * (np.abs(V - 0.5) > 0.42))
            pos = base + np.stack([r * np.cos(ang), V * height, r * np.sin(ang)], axis=-1)
            nrm = np.stack([np.cos(ang), np.zeros_like(ang), np.sin(ang)], axis=-1)
            tan = np.stack([-np.sin(ang), np.zeros_like(ang), np.cos(ang)], axis=-1)
            uv = np.stack([U * uv_scale[0], V * uv_scale[1]], axis=-1)
            return pos, nrm, tan, uv
        self.grid(fn, segments, rings, material)

    def uv_sphere(self, center, radius, segments, rings, material, uv_scale=(1.0, 1.0)):
        center = np.asarray(center, dtype=np.float64)

        def fn(U, V):
            ang = U * 2.0 * np.pi
            th = (V * 0.998 + 0.001) * np.pi          # avoid degenerate pole triangles
            d = np.stack([np.sin(th) * np.cos(ang), np.cos(th), np.sin(th) * np.sin(ang)], axis=-1)
            tan = np.stack([-np.sin(ang), np.zeros_like(ang), np.cos(ang)], axis=-1)
            uv = np.stack([U * uv_scale[0], V * uv_scale[1]], axis=-1)
            return center + radius * d, d, tan, uv
        self.grid(fn, segments, rings, material)

    def arch(self, c0, c1, tube_radius, nu, nv, material):
        """Half-circular tube from c0 to c1 (same height), rising in +y."""
        c0, c1 = np.asarray(c0, dtype=np.float64), np.asarray(c1, dtype=np.float64)
        mid = 0.5 * (c0 + c1)
        half = 0.5 * np.linalg.norm(c1 - c0)
        ax = (c1 - c0) / (2.0 * half)
        upv = np.array([0.0, 1.0, 0.0])
        side = np.cross(ax, upv)

        def fn(U, V):
            a = U * np.pi
            centre = mid + (-np.cos(a))[..., None] * half * ax + np.sin(a)[..., None] * half * upv
            radial = (-np.cos(a))[..., None] * ax + np.sin(a)[..., None] * upv
            tang = np.sin(a)[..., None] * ax + np.cos(a)[..., None] * upv
            b = V * 2.0 * np.pi
            nrm = np.cos(b)[..., None] * radial + np.sin(b)[..., None] * side
            pos = centre + tube_radius * nrm
            uv = np.stack([U * 4.0, V], axis=-1)
            return pos, nrm, tang, uv
        self.grid(fn, nu, nv, material)

    def curtain(self, top_left, width_vec, drop, n, material, phase=0.0, folds=5.0, depth=0.25, uv_scale=(2.0, 2.0)):
        top_left, width_vec = np.asarray(top_left, dtype=np.float64), np.asarray(width_vec, dtype=np.float64)
        wlen = np.linalg.norm(width_vec)
        wdir = width_vec / wlen
        out = np.cross(wdir, np.array([0.0, 1.0, 0.0]))

        def fn(U, V):
            wave = depth * np.sin(U * folds * 2.0 * np.pi + phase) * (0.3 + 0.7 * V)
            dwave = depth * folds * 2.0 * np.pi / wlen * np.cos(U * folds * 2.0 * np.pi + phase) * (0.3 + 0.7 * V)
            sag = 0.15 * np.sin(U * np.pi)
            pos = (top_left + U[..., None] * width_vec + (-(V * drop) - sag * V)[..., None] * np.array([0.0, 1.0, 0.0])
                   + wave[..., None] * out)
            nrm = out - dwave[..., None] * wdir
            nrm = nrm / np.linalg.norm(nrm, axis=-1, keepdims=True)
            tan = wdir + dwave[..., None] * out
            tan = tan / np.linalg.norm(tan, axis=-1, keepdims=True)
            uv = np.stack([U * uv_scale[0], V * uv_scale[1]], axis=-1)
            return pos, nrm, tan, uv
        self.grid(fn, n, n, material)

    def finish(self):
        P = np.concatenate(self.P).astype(F32)
        N = np.concatenate(self.N)
        T = np.concatenate(self.T)
        UV = np.concatenate(self.UV).astype(F32)
        M = np.concatenate(self.M)
        n = P.shape[0]
        N = (N / np.maximum(np.linalg.norm(N, axis=-1, keepdims=True), 1e-30)).astype(F32)
        T = (T / np.maximum(np.linalg.norm(T, axis=-1, keepdims=True), 1e-30)).astype(F32)
        verts = np.zeros((n * 3, 4), dtype=F32)
        verts[:, :3] = P.reshape(-1, 3)
        attrs = np.zeros(n, dtype=abi.TRI_ATTR)
        for k in range(3):
            attrs[f"normal{k}"] = N[:, k]
            attrs[f"tangent{k}"] = T[:, k]
            attrs[f"uv{k}"] = UV[:, k]
        attrs["materialIndex"] = M
        return verts, attrs


# ---------------------------------------------------------------------------------------
# textures
# ---------------------------------------------------------------------------------------
def tex_checker(size, c0, c1, cells=8):
    y, x = np.mgrid[0:size, 0:size]
    m = (((x * cells) // size + (y * cells) // size) & 1).astype(bool)
    img = np.where(m[..., None], np.array(c1, dtype=F32), np.array(c0, dtype=F32)).astype(F32)
    return img


def tex_brick(size, brick, mortar, rows=8, seed=13):
    y, x = np.mgrid[0:size, 0:size]
    fy = y * rows / size
    row = np.floor(fy).astype(np.int64)
    fx = x * (rows / 2) / size + 0.5 * (row & 1)
    edge = (np.minimum(fy - row, 1 - (fy - row)) < 0.06) | (np.minimum(fx - np.floor(fx), 1 - (fx - np.floor(fx))) < 0.03)
    tint = _lattice(np.floor(fx).astype(np.int64), row, np.zeros_like(row), seed)[..., None] * 0.3 + 0.7
    img = np.where(edge[..., None], np.array(mortar, dtype=np.float64), np.array(brick, dtype=np.float64) * tint)
    img[..., 3] = 1.0
    return np.clip(img, 0, 1).astype(F32)


def tex_noise(size, c0, c1, seed=13, frequency=6.0):
    y, x = np.mgrid[0:size, 0:size]
    p = np.stack([x / size, y / size, np.zeros_like(x, dtype=np.float64)], axis=-1)
    n = value_noise3(p, seed=seed, octaves=3, frequency=frequency) * 0.5 + 0.5
    img = np.array(c0, dtype=np.float64) * (1 - n[..., None]) + np.array(c1, dtype=np.float64) * n[..., None]
    return np.clip(img, 0, 1).astype(F32)


def tex_cutout(size, color, holes=6):
    y, x = np.mgrid[0:size, 0:size]
    fx = (x * holes / size) % 1.0 - 0.5
    fy = (y * holes / size) % 1.0 - 0.5
    inside = (fx * fx + fy * fy) < 0.11
    img = np.zeros((size, size, 4), dtype=F32)
    img[..., :3] = np.array(color[:3], dtype=F32)
    img[..., 3] = np.where(inside, 0.0, 1.0)
    return img


# ---------------------------------------------------------------------------------------
# C1 / C2: Cornell box
# ---------------------------------------------------------------------------------------
_WHITE = (0.7169812, 0.7169812, 0.7169812, 1.0)        # Assets/Examples/Materials/White.mat
_RED = (1.0, 0.0, 0.0, 1.0)                            # Red.mat
_GREEN = (0.14809364, 0.9622642, 0.13163051, 1.0)      # Green.mat


def _cornell_shell(sb, mats, tess=1):
    mats.append(pack_material(color=_WHITE, roughness=1.0))     # 0 white
    mats.append(pack_material(color=_RED, roughness=1.0))       # 1 red
    mats.append(pack_material(color=_GREEN, roughness=1.0))     # 2 green
    sb.quad((-1, 0, -1), (2, 0, 0), (0, 0, 2), (0, 1, 0), tess, tess, 0)       # floor
    sb.quad((-1, 0, 1), (2, 0, 0), (0, 2, 0), (0, 0, -1), tess, tess, 0)       # back wall z = +1
    sb.quad((-1, 0, -1), (0, 0, 2), (0, 2, 0), (1, 0, 0), tess, tess, 1)       # left wall (red)
    sb.quad((1, 0, -1), (0, 0, 2), (0, 2, 0), (-1, 0, 0), tess, tess, 2)       # right wall (green)


def _cornell_light():
    return pack_rect_light(center=(0.0, 1.98, 0.0), right=(1, 0, 0), up=(0, 0, 1), size=(0.5, 0.5),
                           color=(17.0, 12.0, 4.0), intensity=1.0, rng=100.0)


def cornell_box() -> Scene:
    """C1/C2: 8 triangles, one rectangle light, black environment."""
    sb, mats = SoupBuilder(), []
    _cornell_shell(sb, mats)
    verts, attrs = sb.finish()
    assert verts.shape[0] == 24
    return Scene("cornell", verts, attrs, np.stack(mats), np.stack([_cornell_light()]),
                 np.zeros(0, dtype=np.uint32),
                 Camera(eye=(0.0, 1.0, -3.4), target=(0.0, 1.0, 0.0), vfov_deg=40.0),
                 environment_mode=0, environment_color=(0.0, 0.0, 0.0, 1.0), environment_intensity=1.0)


# ---------------------------------------------------------------------------------------
# C3: "bunny-class" displaced geodesic icosphere inside the Cornell box
# ---------------------------------------------------------------------------------------
def _icosahedron():
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = np.array([(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t),
                  (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)], dtype=np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array([(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2),
                  (10, 7, 6), (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5),
                  (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)], dtype=np.int64)
    return v, f


def displaced_icosphere(sb, center, radius, frequency, material, amplitude=0.08, noise_seed=7):
    v, f = _icosahedron()
    n = frequency
    tris_dirs = []
    i, j = np.meshgrid(np.arange(n + 1), np.arange(n + 1), indexing="ij")
    for a, b, c in f:
        A, B, Cc = v[a], v[b], v[c]
        pts = (A[None, None] * (n - i - j)[..., None] + B[None, None] * i[..., None] + Cc[None, None] * j[..., None]) / n
        up = [(ii, jj) for ii in range(n) for jj in range(n - ii)]
        ii = np.array([p[0] for p in up])
        jj = np.array([p[1] for p in up])
        tris_dirs.append(np.stack([pts[ii, jj], pts[ii + 1, jj], pts[ii, jj + 1]], axis=1))
        dn = [(ii_, jj_) for ii_ in range(n - 1) for jj_ in range(n - 1 - ii_)]
        if dn:
            ii = np.array([p[0] for p in dn])
            jj = np.array([p[1] for p in dn])
            tris_dirs.append(np.stack([pts[ii + 1, jj], pts[ii + 1, jj + 1], pts[ii, jj + 1]], axis=1))
    D = np.concatenate(tris_dirs)                      # (T, 3, 3) un-normalised directions
    D = D / np.linalg.norm(D, axis=-1, keepdims=True)

    def surf(d):
        return d * (radius * (1.0 + amplitude / radius * value_noise3(d * 1.7, seed=noise_seed, octaves=3, frequency=2.0)))[..., None]
    P = surf(D)
    # smooth normals by central differences of the displaced surface along two tangent directions
    ref = np.where(np.abs(D[..., 1:2]) < 0.9, np.array([0.0, 1.0, 0.0]), np.array([1.0, 0.0, 0.0]))
    t1 = np.cross(ref, D)
    t1 /= np.linalg.norm(t1, axis=-1, keepdims=True)
    t2 = np.cross(D, t1)
    eps = 1e-3

    def nd(d):
        return d / np.linalg.norm(d, axis=-1, keepdims=True)
    du = surf(nd(D + eps * t1)) - surf(nd(D - eps * t1))
    dv = surf(nd(D + eps * t2)) - surf(nd(D - eps * t2))
    N = np.cross(du, dv)
    N /= np.linalg.norm(N, axis=-1, keepdims=True)
    N = np.where((np.sum(N * D, axis=-1, keepdims=True) < 0), -N, N)
    UV = np.stack([np.arctan2(D[..., 2], D[..., 0]) / (2 * np.pi) + 0.5, np.arccos(np.clip(D[..., 1], -1, 1)) / np.pi], axis=-1)
    sb.add(P + np.asarray(center, dtype=np.float64), N, du / np.linalg.norm(du, axis=-1, keepdims=True), UV, material)
    return D.shape[0]


def bunny_box(frequency: int = 59) -> Scene:
    """C3: Cornell box + frequency-59 displaced geodesic icosphere (20*59^2 = 69,620 triangles)."""
    sb, mats = SoupBuilder(), []
    _cornell_shell(sb, mats)
    mats.append(pack_material(color=(0.5, 0.5, 0.5, 1.0), roughness=0.3))      # 3 glossy grey
    n = displaced_icosphere(sb, (0.0, 0.7, 0.0), 0.6, frequency, 3, amplitude=0.08, noise_seed=7)
    assert n == 20 * frequency * frequency
    verts, attrs = sb.finish()
    return Scene("bunny", verts, attrs, np.stack(mats), np.stack([_cornell_light()]), np.zeros(0, dtype=np.uint32),
                 Camera(eye=(0.0, 1.0, -3.4), target=(0.0, 1.0, 0.0), vfov_deg=40.0),
                 environment_mode=0, environment_color=(0.0, 0.0, 0.0, 1.0))


# ---------------------------------------------------------------------------------------
# C4 / C5: "Sponza-class" procedural atrium
# ---------------------------------------------------------------------------------------
def sponza_atrium(tex_size: int = 1024, detail: float = 1.0) -> Scene:
    """~250,000 triangles (detail = 1), 24 materials, 24 RGBA8 textures, 2 rectangle lights, uniform sky.

    `detail` scales every tessellation count (tests use 0.15 -> ~6k triangles); `tex_size` is the edge
    of each of the 24 textures (1024 -> 100.7 MB of TextureData, the size SURVEY.md §8a R9 quotes).
    """
    d = lambda n: max(2, int(round(n * detail)))
    sb = SoupBuilder()
    rng = np.random.RandomState(11)

    # ---- textures (seed 13) -------------------------------------------------------------
    S = tex_size
    images = [
        tex_checker(S, (0.85, 0.82, 0.75, 1), (0.35, 0.33, 0.3, 1), 8),            # 0 floor tiles
        tex_brick(S, (0.62, 0.30, 0.22, 1), (0.75, 0.73, 0.68, 1), 12, seed=13),   # 1 wall brick
        tex_noise(S, (0.55, 0.52, 0.47, 1), (0.85, 0.82, 0.78, 1), seed=14),       # 2 stone
        tex_noise(S, (0.0, 0.85, 0.1, 1), (0.0, 0.35, 0.9, 1), seed=15),           # 3 metallic-roughness map (g = rough, b = metal)
        tex_checker(S, (0.7, 0.1, 0.1, 1), (0.9, 0.8, 0.2, 1), 16),                # 4 curtain red/gold
        tex_cutout(S, (0.2, 0.5, 0.25, 1), 6),                                     # 5 alpha cut-out (MASK)
        tex_noise(S, (0.5, 0.5, 0.5, 1), (1.0, 1.0, 1.0, 1), seed=16, frequency=12.0),  # 6 occlusion
        tex_checker(S, (0.0, 0.0, 0.0, 1), (0.9, 0.6, 0.2, 1), 4),                 # 7 emission pattern
    ]
    for k in range(8, 24):
        c0 = rng.uniform(0.15, 0.6, 3).tolist() + [1.0]
        c1 = rng.uniform(0.4, 0.95, 3).tolist() + [1.0]
        kind = k % 3
        images.append(tex_checker(S, c0, c1, 4 + 2 * (k % 5)) if kind == 0 else
                      tex_brick(S, c0, c1, 6 + (k % 7), seed=13 + k) if kind == 1 else
                      tex_noise(S, c0, c1, seed=13 + k, frequency=4.0 + (k % 4) * 2))
    texture_data = pack_textures(images)

    # ---- materials ----------------------------------------------------------------------
    W = (1.0, 1.0, 1.0, 1.0)
    mats = [
        pack_material(color=W, roughness=0.6, tex_base=0, uv_scale=(1.0, 1.0)),                          # 0 floor
        pack_material(color=W, roughness=0.9, tex_base=1),                                               # 1 brick walls
        pack_material(color=W, roughness=0.8, tex_base=2, tex_occlusion=6),                              # 2 stone columns (+AO)
        pack_material(color=(0.9, 0.9, 0.9, 1), roughness=0.5, metallic=1.0, tex_mr=3),                  # 3 textured rough metal
        pack_material(color=W, roughness=1.0, tex_base=4, sheen=0.8, sheen_tint=0.5),                    # 4 curtain (sheen)
        pack_material(color=W, roughness=0.9, tex_base=5, alpha_mode=2, alpha_cutoff=0.5),               # 5 MASK cut-out curtain
        pack_material(color=(0.8, 0.75, 0.3, 1), roughness=0.25, metallic=1.0),                          # 6 gold
        pack_material(color=(0.95, 0.95, 0.97, 1), roughness=0.05, metallic=1.0),                        # 7 mirror-ish
        pack_material(color=(0.7, 0.1, 0.1, 1), roughness=0.4, clearcoat=1.0, clearcoat_gloss=0.9),      # 8 clearcoat red
        pack_material(color=(0.9, 0.95, 1.0, 0.15), roughness=0.05, ior=1.5),                            # 9 glass (opacity 0.15 -> specTrans)
        pack_material(color=(0.3, 0.5, 0.8, 0.6), roughness=0.7, alpha_mode=1),                          # 10 BLEND
        pack_material(color=(0.6, 0.6, 0.6, 1), roughness=0.35, metallic=0.8, anisotropic=0.8),          # 11 anisotropic metal
        pack_material(color=(0.8, 0.6, 0.5, 1), roughness=0.7, subsurface=0.8),                          # 12 subsurface
        pack_material(color=(0.1, 0.1, 0.1, 1), roughness=0.9, emission=(1.0, 1.0, 1.0), tex_emission=7),  # 13 emissive pattern
        pack_material(color=(0.5, 0.5, 0.5, 1), roughness=0.5, specular_tint=1.0, ior=1.8),              # 14 tinted dielectric
    ]
    for k in range(15, 24):
        mats.append(pack_material(color=W, roughness=float(rng.uniform(0.2, 1.0)), metallic=float(k % 2) * 0.5,
                                  tex_base=8 + (k - 15), uv_scale=(2.0, 2.0), uv_offset=(0.25, 0.5)))
    mats = np.stack(mats)

    # ---- geometry (seed 11): 30 x 12 x 14 m atrium ----------------------------------------
    L, H, Wd = 30.0, 12.0, 14.0
    x0, z0 = -L / 2, -Wd / 2
    sb.quad((x0, 0, z0), (L, 0, 0), (0, 0, Wd), (0, 1, 0), d(120), d(60), 0, uv_scale=(15.0, 7.0))      # floor
    sb.quad((x0, 0, z0), (L, 0, 0), (0, H, 0), (0, 0, 1), d(60), d(24), 1, uv_scale=(10.0, 4.0))        # wall z-
    sb.quad((x0, 0, -z0), (L, 0, 0), (0, H, 0), (0, 0, -1), d(60), d(24), 1, uv_scale=(10.0, 4.0))      # wall z+
    sb.quad((x0, 0, z0), (0, 0, Wd), (0, H, 0), (1, 0, 0), d(28), d(24), 15, uv_scale=(5.0, 4.0))       # end wall x-
    sb.quad((-x0, 0, z0), (0, 0, Wd), (0, H, 0), (-1, 0, 0), d(28), d(24), 16, uv_scale=(5.0, 4.0))     # end wall x+
    # galleries (balcony slabs) along both long walls at y = 6
    sb.quad((x0, 6.0, z0), (L, 0, 0), (0, 0, 2.6), (0, -1, 0), d(60), d(8), 17, uv_scale=(10.0, 1.0))
    sb.quad((x0, 6.0, -z0 - 2.6), (L, 0, 0), (0, 0, 2.6), (0, -1, 0), d(60), d(8), 17, uv_scale=(10.0, 1.0))
    # two rows of 8 columns
    col_x = np.linspace(x0 + 2.5, -x0 - 2.5, 8)
    for zi, zc in enumerate((-4.2, 4.2)):
        for xi, xc in enumerate(col_x):
            mat = 2 if (xi + zi) % 3 else 18 + (xi % 3)
            sb.cylinder((xc, 0.0, zc), 0.45, 6.0, d(64), d(48), mat, uv_scale=(2.0, 6.0), flare=0.25)
        for xi in range(7):
            sb.arch((col_x[xi], 6.0, zc), (col_x[xi + 1], 6.0, zc), 0.28, d(32), d(16), 2 if xi % 2 else 21)
    # 12 sine-draped curtains hanging between columns under the galleries
    for k in range(12):
        zc = -4.9 if k < 6 else 4.9
        xi = k % 6
        mat = 5 if k % 4 == 1 else (10 if k % 6 == 3 else 4)
        sb.curtain((col_x[xi] + 0.55, 5.9, zc), (col_x[xi + 1] - col_x[xi] - 1.1, 0.0, 0.0), 4.6, d(64), mat,
                   phase=float(rng.uniform(0, 6.28)), folds=4.0 + (k % 3), depth=0.22)
    # six "exhibit" spheres on the floor, one per special material
    for k, mat in enumerate((6, 7, 8, 9, 11, 12)):
        sb.uv_sphere((x0 + 6.0 + 3.6 * k, 0.9, (-1.2 if k % 2 else 1.2)), 0.9, d(48), d(24), mat)
    # emissive sign + tinted dielectric plinth + textured metal plate
    sb.quad((-1.5, 7.5, -z0 - 0.05), (3.0, 0, 0), (0, 1.5, 0), (0, 0, -1), d(8), d(4), 13, uv_scale=(2.0, 1.0))
    sb.quad((2.0, 0.02, -3.0), (3.0, 0, 0), (0, 0, 6.0), (0, 1, 0), d(12), d(24), 3, uv_scale=(1.0, 2.0))
    sb.quad((-6.0, 0.02, -3.0), (3.0, 0, 0), (0, 0, 6.0), (0, 1, 0), d(12), d(24), 14, uv_scale=(1.0, 2.0))
    verts, attrs = sb.finish()

    lights = np.stack([
        pack_rect_light(center=(0.0, 11.5, 0.0), right=(1, 0, 0), up=(0, 0, 1), size=(4.0, 2.0),
                        color=(30.0, 28.0, 25.0), rng=60.0),
        pack_rect_light(center=(x0 + 0.2, 4.0, 0.0), right=(0, 0, 1), up=(0, 1, 0), size=(3.0, 1.5),
                        color=(12.0, 14.0, 18.0), rng=60.0),
    ])
    return Scene("sponza", verts, attrs, mats, lights, texture_data,
                 Camera(eye=(x0 + 1.5, 2.2, 0.6), target=(4.0, 3.4, -0.4), vfov_deg=60.0),
                 environment_mode=0, environment_color=(0.6, 0.7, 1.0, 1.0), environment_intensity=1.0,
                 extras={"tex_size": tex_size, "detail": detail})


# ---------------------------------------------------------------------------------------
# zoo: every lobe / light type / alpha mode / texture slot / thin lens, small enough for the oracle
# ---------------------------------------------------------------------------------------
def material_zoo(tex_size: int = 32, basic_sky: bool = False, env_map=None) -> Scene:
    sb = SoupBuilder()
    S = tex_size
    images = [
        tex_checker(S, (0.9, 0.9, 0.9, 1), (0.2, 0.2, 0.25, 1), 8),
        tex_noise(S, (0.0, 0.9, 0.0, 1), (0.0, 0.2, 1.0, 1), seed=21),
        tex_cutout(S, (0.9, 0.6, 0.2, 1), 4),
        tex_noise(S, (0.4, 0.4, 0.4, 1), (1, 1, 1, 1), seed=22, frequency=8.0),
        tex_checker(S, (0, 0, 0, 1), (0.2, 0.9, 0.4, 1), 4),
    ]
    texture_data = pack_textures(images)
    mats = np.stack([
        pack_material(color=(1, 1, 1, 1), roughness=0.8, tex_base=0, uv_scale=(4.0, 4.0), uv_offset=(0.5, -0.25)),  # 0 floor
        pack_material(color=(0.8, 0.75, 0.3, 1), roughness=0.2, metallic=1.0),                      # 1 gold
        pack_material(color=(0.9, 0.95, 1.0, 0.1), roughness=0.02, ior=1.5),                        # 2 glass
        pack_material(color=(0.7, 0.1, 0.1, 1), roughness=0.5, clearcoat=1.0, clearcoat_gloss=0.8), # 3 clearcoat
        pack_material(color=(0.5, 0.5, 0.9, 1), roughness=1.0, sheen=1.0, sheen_tint=0.7, subsurface=0.5),  # 4 sheen + sss
        pack_material(color=(0.9, 0.9, 0.9, 1), roughness=0.6, metallic=1.0, tex_mr=1, anisotropic=0.7),    # 5 textured aniso metal
        pack_material(color=(1, 1, 1, 1), roughness=0.9, tex_base=2, alpha_mode=2, alpha_cutoff=0.5),       # 6 MASK
        pack_material(color=(0.2, 0.6, 0.9, 0.5), roughness=0.6, alpha_mode=1),                             # 7 BLEND
        pack_material(color=(0.6, 0.6, 0.6, 1), roughness=0.7, tex_occlusion=3, specular_tint=0.8, ior=1.7),  # 8 AO + tint
        pack_material(color=(0.05, 0.05, 0.05, 1), roughness=0.9, emission=(2, 2, 2), tex_emission=4),        # 9 emissive
        pack_material(color=(0.9, 0.9, 0.9, 0.0), roughness=0.3, ior=1.33),                                   # 10 fully transmissive, rough
    ])
    sb.quad((-4, 0, -4), (8, 0, 0), (0, 0, 8), (0, 1, 0), 8, 8, 0)
    sb.quad((-4, 0, 4), (8, 0, 0), (0, 5, 0), (0, 0, -1), 4, 4, 8, uv_scale=(2.0, 2.0))
    centers = [(-2.4, 0.6, 0.5), (-1.2, 0.6, -0.6), (0.0, 0.6, 0.5), (1.2, 0.6, -0.6), (2.4, 0.6, 0.5), (0.0, 0.6, 2.0)]
    for c, m in zip(centers, (1, 2, 3, 4, 5, 10)):
        sb.uv_sphere(c, 0.55, 20, 10, m)
    sb.quad((-3.0, 0.2, -1.6), (2.0, 0, 0), (0, 1.6, 0.3), (0, 0.2, -1), 3, 3, 6, uv_scale=(1.0, 1.0))
    sb.quad((1.0, 0.2, -1.6), (2.0, 0, 0), (0, 1.6, 0.3), (0, 0.2, -1), 3, 3, 7)
    sb.quad((-1.0, 2.2, 3.9), (2.0, 0, 0), (0, 1.0, 0), (0, 0, -1), 2, 2, 9, uv_scale=(2.0, 1.0))
    verts, attrs = sb.finish()
    lights = np.stack([
        pack_rect_light(center=(0.0, 4.5, 0.0), right=(1, 0, 0), up=(0, 0, 1), size=(1.5, 1.0), color=(20, 18, 15), rng=30.0),
        pack_point_light((-3.0, 2.5, -2.0), (1.0, 0.8, 0.6), intensity=6.0, rng=12.0),
        pack_spot_light((3.0, 3.5, -2.5), (-0.6, -1.0, 0.5), 50.0, 30.0, (0.6, 0.8, 1.0), intensity=9.0, rng=15.0),
    ])
    cam = Camera(eye=(0.0, 2.2, -5.5), target=(0.0, 0.7, 0.5), vfov_deg=45.0, aperture=0.08, focal_length=6.0)
    if env_map:
        return Scene("zoo_env", verts, attrs, mats, lights, texture_data, cam, environment_mode=0,
                     environment_color=(0.0, 0.0, 0.0, 1.0), environment_intensity=0.9,
                     environment_texture=sky_environment(*env_map), environment_map_rotation=0.15)
    if basic_sky:
        return Scene("zoo_sky", verts, attrs, mats, lights, texture_data, cam, environment_mode=1,
                     environment_color=(0.0, 0.0, 0.0, 1.0), environment_intensity=0.8)
    return Scene("zoo", verts, attrs, mats, lights, texture_data, cam, environment_mode=0,
                 environment_color=(0.5, 0.6, 0.8, 1.0), environment_intensity=0.7)


def sky_environment(width: int = 128, height: int = 64, sun_dir=(0.4, 0.55, -0.3), seed: int = 5) -> np.ndarray:
    """A synthetic HDR lat-long environment (H, W, 4) float32: sky gradient, warm horizon band, dark ground, a small very
    bright sun and a few noise clouds.  Stands in for PathTracer.environmentTexture (PathTracer.cs:32); rows are in the
    order the path tracer consumes them (include/ptmi_plugin.h, PTSceneDesc.envTexture): row r is at v = 1 - (r + 0.5)/H."""
    r = (np.arange(height, dtype=np.float64) + 0.5) / height
    c = (np.arange(width, dtype=np.float64) + 0.5) / width
    v = 1.0 - r[:, None]                                   # uv.y of the row
    u = np.broadcast_to(c[None, :], (height, width))
    # inverse of EvalEnvMap's mapping (sky.hlsl:46-48): theta = (1 - v) * pi, atan2(z, x) = u * 2pi - pi
    theta = (1.0 - v) * np.pi
    phi = u * 2.0 * np.pi - np.pi
    d = np.stack([np.sin(theta) * np.cos(phi), np.cos(theta) * np.ones_like(phi), np.sin(theta) * np.sin(phi)], axis=-1)
    up = np.clip(d[..., 1], 0.0, 1.0)
    sky = (1.0 - up)[..., None] * np.array([0.9, 0.75, 0.6]) + up[..., None] * np.array([0.15, 0.35, 0.9])
    ground = np.array([0.08, 0.07, 0.06])
    img = np.where((d[..., 1] >= 0.0)[..., None], sky, ground)
    s = np.asarray(sun_dir, dtype=np.float64)
    s = s / np.linalg.norm(s)
    cosang = np.clip((d * s).sum(-1), -1.0, 1.0)
    img = img + (np.exp((cosang - 1.0) * 900.0) * 400.0)[..., None] * np.array([1.0, 0.9, 0.7])
    clouds = value_noise3(np.stack([u * 6.0, np.broadcast_to(v, u.shape) * 3.0, np.zeros_like(u)], axis=-1).reshape(-1, 3),
                          seed=seed, octaves=3, frequency=1.5).reshape(height, width)
    img = img + (np.clip(clouds, 0.0, 1.0) * (d[..., 1] > 0.05))[..., None] * 0.6
    out = np.ones((height, width, 4), dtype=F32)
    out[..., :3] = img.astype(F32)
    return out


def furnace(albedo: float = 0.8) -> Scene:
    """White-furnace check (SURVEY.md §8c iii): closed diffuse box is not needed -- a single diffuse
    sphere under a uniform white sky with no analytic lights must converge to a constant."""
    sb = SoupBuilder()
    mats = np.stack([pack_material(color=(albedo ** (1 / 2.2),) * 3 + (1.0,), roughness=1.0)])
    sb.uv_sphere((0, 0, 0), 1.0, 24, 12, 0)
    verts, attrs = sb.finish()
    return Scene("furnace", verts, attrs, mats, np.zeros((0, 16), dtype=F32), np.zeros(0, dtype=np.uint32),
                 Camera(eye=(0, 0, -4.0), target=(0, 0, 0), vfov_deg=30.0),
                 environment_mode=0, environment_color=(1.0, 1.0, 1.0, 1.0), environment_intensity=1.0)


# ---------------------------------------------------------------------------------------
# instanced scene for the HAS_TLAS variant (util/tlas.hlsl)
# ---------------------------------------------------------------------------------------
def _trs(translate=(0, 0, 0), rotate_y_deg=0.0, rotate_x_deg=0.0, scale=(1, 1, 1)):
    ry, rx = np.radians(rotate_y_deg), np.radians(rotate_x_deg)
    Ry = np.array([[np.cos(ry), 0, np.sin(ry), 0], [0, 1, 0, 0], [-np.sin(ry), 0, np.cos(ry), 0], [0, 0, 0, 1]])
    Rx = np.array([[1, 0, 0, 0], [0, np.cos(rx), -np.sin(rx), 0], [0, np.sin(rx), np.cos(rx), 0], [0, 0, 0, 1]])
    S = np.diag([scale[0], scale[1], scale[2], 1.0])
    T = np.eye(4)
    T[:3, 3] = translate
    return T @ Ry @ Rx @ S


def instanced_scene(count: int = 14, detail: int = 12, seed: int = 17) -> Scene:
    """A floor plus `count` instances of three small meshes (sphere, box, cone) with rotations, translations and
    non-uniform scales -- the last on purpose: the reference compares instance-LOCAL hit parameters against the
    world-space distance of the best hit so far (util/tlas.hlsl:47,216-217), and the restatement must keep that."""
    rng = np.random.RandomState(seed)
    meshes = []
    sb = SoupBuilder()
    sb.uv_sphere((0, 0, 0), 0.5, detail * 2, detail, 0)
    meshes.append(sb.finish())
    sb = SoupBuilder()
    for o, eu, ev, n in (((-.5, -.5, -.5), (1, 0, 0), (0, 1, 0), (0, 0, -1)), ((-.5, -.5, .5), (1, 0, 0), (0, 1, 0), (0, 0, 1)),
                         ((-.5, -.5, -.5), (0, 0, 1), (0, 1, 0), (-1, 0, 0)), ((.5, -.5, -.5), (0, 0, 1), (0, 1, 0), (1, 0, 0)),
                         ((-.5, -.5, -.5), (1, 0, 0), (0, 0, 1), (0, -1, 0)), ((-.5, .5, -.5), (1, 0, 0), (0, 0, 1), (0, 1, 0))):
        sb.quad(o, eu, ev, n, 2, 2, 0)
    meshes.append(sb.finish())
    sb = SoupBuilder()

    def cone(U, V):
        ang = U * 2 * np.pi
        r = 0.5 * (1.0 - V)
        pos = np.stack([r * np.cos(ang), V - 0.5, r * np.sin(ang)], axis=-1)
        nrm = np.stack([np.cos(ang), np.full_like(ang, 0.5), np.sin(ang)], axis=-1)
        tan = np.stack([-np.sin(ang), np.zeros_like(ang), np.cos(ang)], axis=-1)
        return pos, nrm, tan, np.stack([U, V], axis=-1)
    sb.grid(cone, detail * 2, detail // 2, 0)
    meshes.append(sb.finish())
    sb = SoupBuilder()
    sb.quad((-6, 0, -6), (12, 0, 0), (0, 0, 12), (0, 1, 0), 6, 6, 0, uv_scale=(6, 6))
    meshes.append(sb.finish())

    verts = np.concatenate([m[0] for m in meshes])
    attrs = np.concatenate([m[1] for m in meshes])
    ranges, start = [], 0
    for m in meshes:
        n = m[0].shape[0] // 3
        ranges.append((start, n))
        start += n
    mats = np.stack([
        pack_material(color=(0.8, 0.8, 0.8, 1), roughness=0.9),                                   # 0 floor
        pack_material(color=(0.8, 0.2, 0.2, 1), roughness=0.4),
        pack_material(color=(0.8, 0.75, 0.3, 1), roughness=0.25, metallic=1.0),
        pack_material(color=(0.2, 0.4, 0.8, 1), roughness=0.6, clearcoat=1.0, clearcoat_gloss=0.8),
        pack_material(color=(0.9, 0.95, 1.0, 0.2), roughness=0.05, ior=1.45),
    ])
    instances = [(3, _trs(), 0)]
    for k in range(count):
        mesh = k % 3
        pos = (rng.uniform(-3.5, 3.5), rng.uniform(0.5, 1.6), rng.uniform(-3.0, 3.5))
        sc = (rng.uniform(0.6, 1.5), rng.uniform(0.6, 1.5), rng.uniform(0.6, 1.5)) if k % 2 else (1.0, 1.0, 1.0)
        instances.append((mesh, _trs(pos, rng.uniform(0, 360), rng.uniform(-30, 30), sc), 1 + k % 4))
    lights = np.stack([pack_rect_light(center=(0.0, 5.0, 0.0), right=(1, 0, 0), up=(0, 0, 1), size=(2.0, 2.0),
                                       color=(14, 13, 12), rng=40.0)])
    return Scene("instanced", verts, attrs, mats, lights, np.zeros(0, dtype=np.uint32),
                 Camera(eye=(0.0, 3.0, -7.5), target=(0.0, 0.8, 0.0), vfov_deg=45.0),
                 environment_mode=0, environment_color=(0.45, 0.55, 0.75, 1.0), environment_intensity=0.8,
                 mesh_ranges=ranges, instances=instances)


def instance_world_bounds(local_verts, local_to_world):
    """renderer.bounds (BVHScene.cs:690): world AABB of the transformed local AABB's eight corners."""
    lo, hi = local_verts[:, :3].min(axis=0).astype(np.float64), local_verts[:, :3].max(axis=0).astype(np.float64)
    corners = np.array([[x, y, z, 1.0] for x in (lo[0], hi[0]) for y in (lo[1], hi[1]) for z in (lo[2], hi[2])])
    w = (local_to_world @ corners.T).T[:, :3]
    return w.min(axis=0).astype(F32), w.max(axis=0).astype(F32)


def bake_instances(scene: Scene) -> Scene:
    """The same geometry as a flat world-space triangle soup (what the non-TLAS path of BVHScene.cs would upload)."""
    P, A = [], []
    for mesh, l2w, material in scene.instances:
        t0, n = scene.mesh_ranges[mesh]
        v = scene.vertices[t0 * 3:(t0 + n) * 3].astype(np.float64)
        h = np.concatenate([v[:, :3], np.ones((v.shape[0], 1))], axis=1)
        w = (l2w @ h.T).T
        out = np.zeros((v.shape[0], 4), F32)
        out[:, :3] = w[:, :3]
        P.append(out)
        a = scene.tri_attrs[t0:t0 + n].copy()
        nt = np.linalg.inv(l2w)[:3, :3].T
        for k in range(3):
            nn = (nt @ a[f"normal{k}"].astype(np.float64).T).T
            a[f"normal{k}"] = (nn / np.linalg.norm(nn, axis=1, keepdims=True)).astype(F32)
        a["materialIndex"] = material
        A.append(a)
    return Scene(scene.name + "_baked", np.concatenate(P), np.concatenate(A), scene.materials, scene.lights, scene.texture_data,
                 scene.camera, scene.environment_mode, scene.environment_color, scene.environment_intensity)


SCENES = {
    "cornell": cornell_box,
    "bunny": bunny_box,
    "sponza": sponza_atrium,
    "zoo": material_zoo,
    "zoo_env": lambda **kw: material_zoo(env_map=kw.pop("env_map", (128, 64)), **kw),
    "furnace": furnace,
    "instanced": instanced_scene,
}


def make_scene(name: str, **kw) -> Scene:
    return SCENES[name](**kw)
