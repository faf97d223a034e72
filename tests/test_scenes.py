"""The synthetic scene generators emit the reference's buffer layouts."""
import numpy as np

from unity_webgpu_pathtracer_amd import abi, scenes
from unity_webgpu_pathtracer_amd.scenes import camera_matrices, frame_params, pack_material, pack_rect_light, pack_textures


def test_vertex_and_attribute_layout():
    s = scenes.cornell_box()
    assert s.vertices.dtype == np.float32 and s.vertices.shape == (24, 4)
    assert (s.vertices[:, 3] == 0).all()                      # MeshProcessing.compute:100-102 writes w = 0
    assert s.tri_attrs.dtype == abi.TRI_ATTR and s.tri_attrs.nbytes == 8 * 128
    n = np.stack([s.tri_attrs[f"normal{k}"] for k in range(3)])
    assert np.allclose(np.linalg.norm(n, axis=-1), 1.0, atol=1e-6)
    assert set(s.tri_attrs["materialIndex"].tolist()) == {0, 1, 2}


def test_material_packing_order():
    """BVHScene.cs:241-282."""
    m = pack_material(color=(0.5, 0.25, 1.0, 0.8), transmission=0.5, emission=(1, 2, 3), metallic=0.3, glossiness=0.25,
                      ior=1.4, alpha_mode=2, alpha_cutoff=0.4, anisotropic=0.1, specular=0.2, specular_tint=0.3, sheen=0.4,
                      sheen_tint=0.5, subsurface=0.6, clearcoat=0.7, clearcoat_gloss=0.8, tex_base=3, tex_mr=4,
                      tex_emission=5, tex_occlusion=6, uv_scale=(2, 3), uv_offset=(0.1, 0.2))
    assert m.shape == (32,) and m.dtype == np.float32
    assert np.allclose(m[0:3], np.power(np.float32([0.5, 0.25, 1.0]), np.float32(2.2)))
    assert np.isclose(m[3], 0.8 * 0.5) and np.isclose(m[21], 1 - 0.4)
    assert m[4:8].tolist() == [1, 2, 3, np.float32(0.4)]
    assert np.allclose(m[8:12], [0.3, 0.75, 1.0, 1.4])          # roughness = 1 - _Glossiness
    assert np.allclose(m[12:20], [2, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7]) and np.isclose(m[20], 0.8)
    assert m[22:28].tolist() == [3, 4, -1, 5, 6, -1]
    assert np.allclose(m[28:32], [2, 3, 0.1, 0.2])
    d = pack_material()
    assert np.isclose(d[11], 1.1) and (d[22:28] == -1).all()   # default ior 1.1, no textures


def test_rect_light_packing():
    """PathTracer.cs:343-348,442-460: position -= (u+v)/2, type = LightType.Rectangle (3) as int bits."""
    l = pack_rect_light(center=(0, 1.98, 0), right=(1, 0, 0), up=(0, 0, 1), size=(0.5, 0.5), color=(17, 12, 4))
    assert np.allclose(l[0:3], [-0.25, 1.98, -0.25])
    assert l[3:4].view(np.uint32)[0] == 3
    assert np.allclose(l[8:11], [0.5, 0, 0]) and np.isclose(l[11], 0.25) and np.allclose(l[12:15], [0, 0, 0.5])
    n = np.cross(l[8:11], l[12:15])
    assert n[1] < 0                                              # faces -y


def test_texture_packing():
    """CopyTextureData.compute:21-35: descriptors {w, h, offset, 0}, texel = a<<24 | b<<16 | g<<8 | r."""
    a = np.zeros((2, 3, 4), np.float32)
    a[..., 0], a[..., 1], a[..., 2], a[..., 3] = 1.0, 0.5, 0.25, 1.0
    b = np.ones((1, 1, 4), np.float32)
    t = pack_textures([a, b])
    assert t.dtype == np.uint32 and t.size == 8 + 6 + 1
    assert t[0:4].tolist() == [3, 2, 8, 0] and t[4:8].tolist() == [1, 1, 14, 0]
    assert t[8] == (255 << 24) | (63 << 16) | (127 << 8) | 255
    assert t[14] == 0xFFFFFFFF


def test_camera_matrices_unity_convention():
    s = scenes.cornell_box()
    inv, c2w = camera_matrices(s.camera, 1.0)
    M = c2w.reshape(4, 4).T                                     # element (r, c) at c*4 + r
    assert np.allclose(M[:3, 3], [0, 1, -3.4])
    assert np.allclose(M[:3, 2], [0, 0, -1])                    # camera looks down -Z (GL convention) towards world +z
    P = inv.reshape(4, 4).T
    d = P @ np.array([0, 0, 0, 1.0])
    assert d[2] < 0
    p = frame_params(s, 1920, 1080, spp=8, current_sample=16, seed=0x1_0000_0001)
    assert p.RngSeedRoot == 1 and p.SamplesPerPass == 8 and p.CurrentSample == 16 and p.MaxRayBounces == 4
    assert frame_params(s, 8, 8, spp=0, max_bounces=0).SamplesPerPass == 1     # host clamps (PathTracer.cs:234-235)
    assert frame_params(s, 8, 8, spp=0, max_bounces=0).MaxRayBounces == 1


def test_scene_features():
    assert scenes.cornell_box().features == abi.PT_FEATURE_HAS_LIGHTS
    assert scenes.furnace().features == 0
    z = scenes.material_zoo()
    assert z.features == abi.PT_FEATURE_HAS_LIGHTS | abi.PT_FEATURE_HAS_TEXTURES
    types = sorted(z.lights[:, 3].view(np.uint32).tolist())
    assert types == [0, 2, 3]                                    # spot, point, rectangle
