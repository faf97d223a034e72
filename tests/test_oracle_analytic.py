"""End-to-end ANALYTIC pins of the shading half of the CPU oracle.

The reference holds no fixtures and its HLSL cannot run here, so these rows cannot be pinned by anything the reference
produced.  What can be done is to render small constructed scenes whose image has a CLOSED FORM that follows from the
geometry and from what the shader documents (camera model, light shapes, texture addressing) -- computed below in float64
with numpy, independently of oracle/pt_oracle.cpp -- and to compare.  Each test names the reference lines it pins.  The
MI355X kernels inherit every one of them through bit-equality with the oracle (tests/test_gpu_parity.py).
"""
import numpy as np
import pytest

from unity_webgpu_pathtracer_amd import plugin, scenes
from unity_webgpu_pathtracer_amd.scenes import Camera, Scene, SoupBuilder, pack_material

F32 = np.float32
NO_LIGHTS = np.zeros((0, 16), dtype=F32)
NO_TEX = np.zeros(0, dtype=np.uint32)


def _render(oracle, s, w, h, **kw):
    nodes, tris = plugin.build_cwbvh(s.vertices)
    b = oracle.SceneBuffers(s, nodes, tris)
    img, _ = oracle.render(b, scenes.frame_params(s, w, h, **kw))
    return img.astype(np.float64)


def _hidden_geometry(sb, material=0):
    """A BVH needs a triangle: one far behind every camera used here."""
    sb.quad((-0.01, -0.01, -500.0), (0.02, 0, 0), (0, 0.02, 0), (0, 0, 1), 1, 1, material)


def _camera_basis(cam):
    """Unity conventions, float64: left-handed, right = up x forward (BVHScene / Camera.cameraToWorldMatrix)."""
    eye = np.asarray(cam.eye, np.float64)
    fwd = np.asarray(cam.target, np.float64) - eye
    fwd /= np.linalg.norm(fwd)
    right = np.cross(np.asarray(cam.up, np.float64), fwd)
    right /= np.linalg.norm(right)
    return eye, right, np.cross(fwd, right), fwd


def _pixel_directions(cam, w, h):
    """World direction through the CENTRE of every pixel of a pinhole camera with vertical field of view vfov:
    pixel (x, y) -> ndc = ((x + 0.5) / w * 2 - 1, (y + 0.5) / h * 2 - 1), y up (PathTracer.compute:52-72, util/camera.hlsl:13-24)."""
    eye, right, up, fwd = _camera_basis(cam)
    t = np.tan(np.radians(cam.vfov_deg) / 2)
    x = ((np.arange(w) + 0.5) / w * 2 - 1) * t * (w / h)
    y = ((np.arange(h) + 0.5) / h * 2 - 1) * t
    X, Y = np.meshgrid(x, y)
    d = fwd[None, None, :] + X[..., None] * right + Y[..., None] * up
    return eye, d / np.linalg.norm(d, axis=-1, keepdims=True)


def _project(cam, w, h, p):
    """World point -> continuous pixel coordinates (pixel centres at integer + 0.5)."""
    eye, right, up, fwd = _camera_basis(cam)
    v = np.asarray(p, np.float64) - eye
    z = v @ fwd
    t = np.tan(np.radians(cam.vfov_deg) / 2)
    return ((v @ right) / z / (t * w / h) + 1) / 2 * w, ((v @ up) / z / t + 1) / 2 * h, z


# ---------------------------------------------------------------------------------------
# R15: the basic sky gradient (util/sky.hlsl:100-112) seen by camera rays (rayDepth 0: intensity 1, util/sky.hlsl:94-96)
# ---------------------------------------------------------------------------------------
def test_basic_sky_gradient_closed_form(oracle):
    sb = SoupBuilder()
    _hidden_geometry(sb)
    verts, attrs = sb.finish()
    cam = Camera(eye=(0, 1, 0), target=(0, 1.6, 1), vfov_deg=70.0)
    s = Scene("sky", verts, attrs, np.stack([pack_material()]), NO_LIGHTS, NO_TEX, cam, environment_mode=1, environment_intensity=3.0)
    w, h = 48, 36
    img = _render(oracle, s, w, h, spp=64, seed=5, max_bounces=1)
    _, d = _pixel_directions(cam, w, h)
    a = np.clip(0.5 * (d[..., 1] + 1.0), 0.0, 1.0)
    blue = np.array([0.5 ** 2.2, 0.7 ** 2.2, 1.0])
    want = (1.0 - a)[..., None] * np.ones(3) + a[..., None] * blue
    assert np.abs(img[..., :3] - want).max() < 4e-3            # the gradient moves ~1e-2 per pixel here; 64 jittered samples average to the centre
    assert (img[..., 3] == 1.0).all()


# ---------------------------------------------------------------------------------------
# R2 + R7: a rectangle light seen directly.  Its image is the perspective projection of the rectangle (camera model), every
# covered pixel holds the light's emission exactly once (util/pathtrace.hlsl:42-49: no falloff, no cosine), the back face is
# invisible (util/intersect.hlsl:29-54: dot(normal, dir) < 0), and the pixel-filter jitter is symmetric (PathTracer.compute:31-38,66-72)
# ---------------------------------------------------------------------------------------
def _light_scene(center, right, up, size, cam, emission=(5.0, 3.0, 2.0)):
    sb = SoupBuilder()
    _hidden_geometry(sb)
    verts, attrs = sb.finish()
    light = scenes.pack_rect_light(center, right, up, size, emission)
    return Scene("light", verts, attrs, np.stack([pack_material()]), np.stack([light]), NO_TEX, cam,
                 environment_mode=0, environment_color=(0.0, 0.0, 0.0, 1.0))


@pytest.mark.parametrize("offset", [(0.0, 0.0), (0.9, -0.5), (-1.3, 0.7)])
def test_rectangle_light_projects_where_the_camera_model_says(oracle, offset):
    cam = Camera(eye=(0.5, 1.0, -6.0), target=(0.3, 1.2, 0.0), vfov_deg=40.0)
    eye, right, up, fwd = _camera_basis(cam)
    centre = eye + 6.0 * fwd + offset[0] * right + offset[1] * up
    size = (0.5, 0.3)
    # pack_rect_light: normal = normalize(cross(u, v)); (right, -up) makes it face the camera (against fwd)
    s = _light_scene(centre, right, -up, size, cam)
    w, h = 96, 72
    img = _render(oracle, s, w, h, spp=64, seed=9, max_bounces=1)
    lum = img[..., 0]
    assert lum.max() <= 5.0 + 1e-5 and np.allclose(img[..., 1] * 5.0, img[..., 0] * 3.0, atol=1e-5)        # emission, never scaled
    px, py, z = _project(cam, w, h, centre)
    ys, xs = np.mgrid[0:h, 0:w]
    m = lum.sum()
    cx, cy = (lum * (xs + 0.5)).sum() / m, (lum * (ys + 0.5)).sum() / m
    assert abs(cx - px) < 0.08 and abs(cy - py) < 0.08, (cx, px, cy, py)
    # flux: emission x the rectangle's area in pixels (the plane is parallel to the image plane)
    pix = 2 * z * np.tan(np.radians(cam.vfov_deg) / 2) / h
    assert abs(m / 5.0 - size[0] * size[1] / pix ** 2) / (size[0] * size[1] / pix ** 2) < 0.02
    # the back of the light is not seen
    back = _render(oracle, _light_scene(centre, right, up, size, cam), w, h, spp=4, seed=9, max_bounces=1)
    assert back[..., :3].max() == 0.0


# ---------------------------------------------------------------------------------------
# R2: thin lens (util/camera.hlsl:26-40).  A small light at depth z with the focus at ft is imaged as a disc of radius
# R |1 - ft / z| on the focal plane, R = Aperture / 2, uniformly filled (ConcentricSampleDisk): variance R^2 |1 - ft/z|^2 / 4 per axis
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("depth", [4.0, 6.0, 10.0])
def test_thin_lens_circle_of_confusion(oracle, depth):
    ft, aperture = 6.0, 0.3
    cam = Camera(eye=(0, 0, 0), target=(0, 0, 1), vfov_deg=30.0, aperture=aperture, focal_length=ft)
    pin = Camera(eye=(0, 0, 0), target=(0, 0, 1), vfov_deg=30.0)
    size = (0.1, 0.1)
    w, h = 96, 96
    ys, xs = np.mgrid[0:h, 0:w]

    def second_moment(c):
        s = _light_scene((0.0, 0.0, depth), (1, 0, 0), (0, -1, 0), size, c, emission=(1.0, 1.0, 1.0))
        lum = _render(oracle, s, w, h, spp=1024, seed=21, max_bounces=1)[..., 0]
        m = lum.sum()
        cx, cy = (lum * (xs + 0.5)).sum() / m, (lum * (ys + 0.5)).sum() / m
        return (lum * ((xs + 0.5 - cx) ** 2 + (ys + 0.5 - cy) ** 2)).sum() / m, cx, cy

    v_lens, cx, cy = second_moment(cam)
    v_pin, _, _ = second_moment(pin)                       # rectangle size + pixel filter + pixel box: common to both
    assert abs(cx - w / 2) < 0.1 and abs(cy - h / 2) < 0.1
    pix_at_depth = 2 * depth * np.tan(np.radians(15.0)) / h                 # world size of a pixel at the light's depth
    r = aperture / 2 * abs(1 - depth / ft) / pix_at_depth                   # blur radius in pixels, seen from the light's plane
    want = 2 * r * r / 4                                                     # x + y variance of a uniform disc
    got = v_lens - v_pin
    # ~2,500 camera samples land on the light per render: the second moments carry ~3 % of sampling noise each
    if depth == ft:
        assert abs(got) < 0.08, got                                          # in focus: as sharp as the pinhole
    else:
        assert abs(got - want) / want < 0.08, (got, want)


# ---------------------------------------------------------------------------------------
# R8 + R9: texture addressing and bilinear filtering (util/texture.hlsl:25-76) through the emission slot
# (util/material.hlsl:8-21: the texel REPLACES the emission), on a quad whose uv the attributes interpolate (util/bvh.hlsl:201-215)
# ---------------------------------------------------------------------------------------
def test_emission_texture_is_bilinear_in_uv(oracle):
    rng = np.random.RandomState(3)
    tw, th = 5, 4
    texel = rng.randint(0, 256, size=(th, tw, 4)).astype(np.uint32)
    img_tex = (texel.astype(F32) + F32(0.25)) / F32(255.0)              # pack_textures truncates (uint)(x * 255): + 0.25 keeps every byte
    tex = scenes.pack_textures([img_tex])
    assert np.array_equal(tex[4:] & 0xFF, texel[..., 0].reshape(-1))     # row-major from row 0, red in the low byte (CopyTextureData.compute:21-35)
    sb = SoupBuilder()
    ext = 2.0
    sb.quad((-1.0, -1.0, 4.0), (ext, 0, 0), (0, ext, 0), (0, 0, -1), 1, 1, 0)
    verts, attrs = sb.finish()
    mat = pack_material(color=(0, 0, 0, 1), roughness=1.0, emission=(1.0, 1.0, 1.0), tex_emission=0)
    cam = Camera(eye=(0, 0, 0), target=(0, 0, 1), vfov_deg=24.0)
    s = Scene("tex", verts, attrs, np.stack([mat]), NO_LIGHTS, tex, cam, environment_mode=0, environment_color=(0.0, 0.0, 0.0, 1.0))
    w, h = 40, 40
    img = _render(oracle, s, w, h, spp=32, seed=2, max_bounces=1)
    eye, d = _pixel_directions(cam, w, h)
    hit = eye + d * (4.0 / d[..., 2])[..., None]
    u, v = (hit[..., 0] + 1.0) / ext, (hit[..., 1] + 1.0) / ext
    assert u.min() > 0 and u.max() < 1 and v.min() > 0 and v.max() < 1
    tu, tv = u * (tw - 1), v * (th - 1)                                   # util/texture.hlsl:50-51
    tx, ty = np.floor(tu).astype(int), np.floor(tv).astype(int)
    fu, fv = (tu - tx)[..., None], (tv - ty)[..., None]
    c = texel.astype(np.float64) / 255.0
    x1, y1 = np.minimum(tx + 1, tw - 1), np.minimum(ty + 1, th - 1)
    want = (c[ty, tx] * (1 - fu) + c[ty, x1] * fu) * (1 - fv) + (c[y1, tx] * (1 - fu) + c[y1, x1] * fu) * fv
    # the footprint of a pixel is ~1/13 of a texel and the filter is piecewise linear: the jittered mean sits at the centre value
    # except in pixels that straddle a texel boundary (a kink): allow those the kink's worth
    err = np.abs(img[..., :3] - want[..., :3])
    assert np.median(err) < 2e-3 and err.max() < 4e-2, (np.median(err), err.max())


# ---------------------------------------------------------------------------------------
# R10 / R11: spot cone and range falloff (util/light.hlsl:60-93) as RATIOS between renders that differ in one light parameter:
# the BSDF, the geometry and the (absent) 1 / d^2 cancel, what is left is the factor itself
# ---------------------------------------------------------------------------------------
def _floor_scene(light):
    sb = SoupBuilder()
    sb.quad((-4.0, 0.0, -4.0), (8.0, 0, 0), (0, 0, 8.0), (0, 1, 0), 1, 1, 0)
    verts, attrs = sb.finish()
    cam = Camera(eye=(0, 6.0, -0.001), target=(0, 0, 0), up=(0, 0, 1), vfov_deg=50.0)
    mat = pack_material(color=(0.8, 0.8, 0.8, 1), roughness=1.0)
    return Scene("floor", verts, attrs, np.stack([mat]), np.stack([light]), NO_TEX, cam, environment_mode=0, environment_color=(0.0, 0.0, 0.0, 1.0))


def _floor_points(s, w, h):
    eye, d = _pixel_directions(s.camera, w, h)
    return eye + d * (-eye[1] / d[..., 1])[..., None]


def test_spot_cone_factor(oracle):
    pos, fwd = np.array([0.3, 2.0, -0.2]), np.array([0.2, -1.0, 0.1])
    outer, inner = 70.0, 40.0
    w, h = 96, 96
    kw = dict(spp=32, seed=4, max_bounces=1, russian_roulette=False)
    spot = _render(oracle, _floor_scene(scenes.pack_spot_light(pos, fwd, outer, inner, (4.0, 4.0, 4.0), rng=50.0)), w, h, **kw)[..., 0]
    point = _render(oracle, _floor_scene(scenes.pack_point_light(pos, (4.0, 4.0, 4.0), rng=50.0)), w, h, **kw)[..., 0]
    P = _floor_points(_floor_scene(scenes.pack_point_light(pos, (1, 1, 1))), w, h)
    to = P - pos
    cos_t = (to / np.linalg.norm(to, axis=-1, keepdims=True)) @ (fwd / np.linalg.norm(fwd))
    c_out, c_in = np.cos(np.radians(outer) / 2), np.cos(np.radians(inner) / 2)
    factor = np.clip((cos_t - c_out) / (c_in - c_out), 0.0, 1.0)          # 0 outside the cone, ramp in cos(theta), 1 inside the inner cone
    assert point.min() > 0
    ratio = spot / point
    safe = (np.abs(cos_t - c_out) > 0.01) & (np.abs(cos_t - c_in) > 0.01)  # the pixel filter rounds the two kinks of the ramp
    assert np.abs(ratio - factor)[safe].max() < 0.04, np.abs(ratio - factor)[safe].max()
    assert (spot[cos_t < c_out - 0.04] == 0).all() and np.allclose(ratio[cos_t > c_in + 0.04], 1.0, atol=1e-5)      # a few pixels clear of the edges (Gaussian pixel filter)


def test_range_falloff_factor(oracle):
    pos = np.array([0.0, 1.5, 0.0])
    w, h = 64, 64
    kw = dict(spp=32, seed=6, max_bounces=1, russian_roulette=False)
    far = _render(oracle, _floor_scene(scenes.pack_point_light(pos, (4.0, 4.0, 4.0), rng=1e6)), w, h, **kw)[..., 0]
    near = _render(oracle, _floor_scene(scenes.pack_point_light(pos, (4.0, 4.0, 4.0), rng=3.0)), w, h, **kw)[..., 0]
    d = np.linalg.norm(_floor_points(_floor_scene(scenes.pack_point_light(pos, (1, 1, 1))), w, h) - pos, axis=-1)
    r = d / 3.0
    atten = np.clip(1.0 / (1.0 + 25.0 * r * r) * np.clip((1.0 - r) * 5.0, 0.0, 1.0), 0.0, 1.0)     # util/light.hlsl:66-72
    atten[d > 3.0] = 0.0
    assert far.min() > 0
    ratio = near / far                                                      # at range 1e6 the factor is 1 - O(1e-11)
    inside = d < 2.9
    assert np.abs(ratio - atten)[inside].max() < 0.01 * atten[inside].max() + 3e-3
    assert (near[d > 3.3] == 0).all() and (d > 3.3).sum() > 200


# ---------------------------------------------------------------------------------------
# R1: the running mean of PathTracer.compute:89-98, on a scene whose every sample is the same colour
# ---------------------------------------------------------------------------------------
def test_progressive_mean_closed_form(oracle):
    sb = SoupBuilder()
    _hidden_geometry(sb)
    verts, attrs = sb.finish()
    sky = (0.25, 0.5, 0.75)
    s = Scene("const", verts, attrs, np.stack([pack_material()]), NO_LIGHTS, NO_TEX, Camera(eye=(0, 0, 0), target=(0, 0, 1)),
              environment_mode=0, environment_color=sky + (1.0,), environment_intensity=7.0)     # intensity applies from depth 1 on only
    nodes, tris = plugin.build_cwbvh(s.vertices)
    b = oracle.SceneBuffers(s, nodes, tris)
    w, h, n, k = 16, 8, 4, 10
    first, _ = oracle.render(b, scenes.frame_params(s, w, h, spp=n, current_sample=0))
    assert np.allclose(first[..., :3], sky, atol=1e-6) and (first[..., 3] == 1).all()
    acc = np.zeros((h, w, 4), F32)
    acc[..., :3] = (2.0, 1.0, 0.5)
    out, _ = oracle.render(b, scenes.frame_params(s, w, h, spp=n, current_sample=k), accumulated=acc)
    want = (n * np.array(sky) + k * np.array([2.0, 1.0, 0.5])) / (k + n)
    assert np.allclose(out[..., :3], want, rtol=1e-6) and (out[..., 3] == 1).all()


# ---------------------------------------------------------------------------------------
# PathTracer.compute:76-82: the firefly filter scales a sample down to MaxFireflyLuminance, keeping its colour
# ---------------------------------------------------------------------------------------
def test_firefly_filter_clamps_luminance(oracle):
    cam = Camera(eye=(0, 0, -4.0), target=(0, 0, 0), vfov_deg=20.0)
    s = _light_scene((0, 0, 0), (1, 0, 0), (0, -1, 0), (6.0, 6.0), cam, emission=(40.0, 20.0, 10.0))       # fills the frame
    img = _render(oracle, s, 8, 8, spp=4, seed=1, max_bounces=1, firefly=True, max_firefly_luminance=10.0)
    lum = 0.299 * 40.0 + 0.587 * 20.0 + 0.114 * 10.0
    assert np.allclose(img[..., :3], np.array([40.0, 20.0, 10.0]) * 10.0 / lum, rtol=1e-5)
    plain = _render(oracle, s, 8, 8, spp=4, seed=1, max_bounces=1)
    assert np.allclose(plain[..., :3], [40.0, 20.0, 10.0])


# ---------------------------------------------------------------------------------------
# util/pathtrace.hlsl:120-128: Russian roulette is unbiased -- same mean with and without it
# ---------------------------------------------------------------------------------------
def test_russian_roulette_is_unbiased(oracle):
    s = scenes.cornell_box()
    nodes, tris = plugin.build_cwbvh(s.vertices)
    b = oracle.SceneBuffers(s, nodes, tris)
    means = []
    for rr in (False, True):
        img, _ = oracle.render(b, scenes.frame_params(s, 40, 40, spp=192, seed=17, max_bounces=6, russian_roulette=rr))
        assert not np.isnan(img).any()
        means.append(img[..., :3].astype(np.float64).mean(axis=(0, 1)))
    assert np.allclose(means[0], means[1], rtol=0.03), means


# ---------------------------------------------------------------------------------------
# R8 + R12: EvalBRDF pointwise.  A SECOND transcription of util/material.hlsl:84-138 and util/brdf.hlsl:9-238 (+ util/sampling.hlsl
# GTR1 / GTR2Aniso / SmithG / SmithGAniso / SchlickWeight / DielectricFresnel, util/common.hlsl GetONB with ONB_METHOD 1), written
# in vectorised float64 numpy straight from the HLSL and sharing nothing with oracle/pt_oracle.cpp.  Two transcriptions that agree
# on value AND pdf for thousands of direction pairs and a dozen materials leave little room for a mis-copied term in either
# (the distribution tests of tests/test_oracle.py pin pdf against sampling, not the value f).
# ---------------------------------------------------------------------------------------
def _np_eval_brdf(md, V, N, L):
    """md: 32 floats of MaterialData (no textures); V, L: (n, 3) world directions; N: (3,).  Returns f (n, 3), pdf (n,) in float64."""
    md = md.astype(np.float64)
    base, opacity = md[0:3], md[3]
    metallic, rough = md[8], max(md[9], 0.001)
    ior = min(max(md[11], 1.001), 2.0)
    aniso = min(max(md[13], -0.9), 0.9)
    spec_tint, sheen, sheen_tint, subsurface, clearcoat = md[15], md[16], md[17], md[18], md[19]
    cc_rough = 0.1 + md[20] * (0.001 - 0.1)
    spec_trans = 1.0 - min(max(opacity, 0.0), 1.0)
    aspect = np.sqrt(1.0 - aniso * 0.9)
    ax, ay = max(0.001, rough / aspect), max(0.001, rough * aspect)
    N = N / np.linalg.norm(N)
    ray_dir = -V                                                   # MaterialFromData: the ray that arrives along -V
    eta = np.where((ray_dir @ N) < 0.0, 1.0 / ior, ior)           # (n,)
    # GetONB, ONB_METHOD 1
    k = 1.0 / max(1.0 + N[2], 0.00001)
    a_, b_ = N[1] * k, N[1] * N[1] * k
    c_ = -N[0] * a_
    X = np.array([N[2] + b_, c_, -N[0]]); X /= np.linalg.norm(X)
    Y = np.array([c_, 1.0 - b_, -N[1]]); Y /= np.linalg.norm(Y)
    Vl = np.stack([V @ X, V @ Y, V @ N], -1)
    Ll = np.stack([L @ X, L @ Y, L @ N], -1)
    H = np.where((Ll[:, 2] > 0.0)[:, None], Ll + Vl, Ll + Vl * eta[:, None])
    H /= np.linalg.norm(H, axis=-1, keepdims=True)
    H = np.where((H[:, 2] < 0.0)[:, None], -H, H)
    lum = lambda c: c[..., 0] * 0.299 + c[..., 1] * 0.587 + c[..., 2] * 0.114
    schlick = lambda u: np.clip(1.0 - u, 0.0, 1.0) ** 5
    lerp = lambda a, b, t: a + t * (b - a)

    def fresnel(cos_i, e):
        s2 = e * e * (1.0 - cos_i * cos_i)
        cos_t = np.sqrt(np.maximum(1.0 - s2, 0.0))
        rs = (e * cos_t - cos_i) / (e * cos_t + cos_i)
        rp = (e * cos_i - cos_t) / (e * cos_i + cos_t)
        return np.where(s2 > 1.0, 1.0, 0.5 * (rs * rs + rp * rp))

    def gtr2(h):
        c = (h[:, 0] / ax) ** 2 + (h[:, 1] / ay) ** 2 + h[:, 2] ** 2
        return 1.0 / (np.pi * ax * ay * c * c)

    def smith_aniso(w):
        nv = np.abs(w[:, 2])
        return 2.0 * nv / (nv + np.sqrt((w[:, 0] * ax) ** 2 + (w[:, 1] * ay) ** 2 + nv * nv))

    def smith(nv, alpha):
        a2, b2 = alpha * alpha, nv * nv
        return 2.0 * nv / (nv + np.sqrt(a2 + b2 - a2 * b2))

    lb = lum(base)
    ctint = base / lb if lb > 0.0 else np.ones(3)
    F0 = ((1.0 - eta) / (1.0 + eta)) ** 2                                         # (n,)
    Cspec0 = F0[:, None] * lerp(np.ones(3), ctint, spec_tint)
    Csheen = lerp(np.ones(3), ctint, sheen_tint)
    diel_wt, metal_wt, glass_wt = (1.0 - metallic) * (1.0 - spec_trans), metallic, (1.0 - metallic) * spec_trans
    sw = schlick(Vl[:, 2])
    diff_pr = diel_wt * lb * np.ones_like(sw)
    diel_pr = diel_wt * lum(lerp(Cspec0, 1.0, sw[:, None]))
    metal_pr = metal_wt * lum(lerp(base[None, :], 1.0, sw[:, None]))
    glass_pr = glass_wt * np.ones_like(sw)
    cc_pr = 0.25 * clearcoat * np.ones_like(sw)
    tot = diff_pr + diel_pr + metal_pr + glass_pr + cc_pr
    diff_pr, diel_pr, metal_pr, glass_pr, cc_pr = (p / tot for p in (diff_pr, diel_pr, metal_pr, glass_pr, cc_pr))
    refl = Ll[:, 2] * Vl[:, 2] > 0.0
    up = Ll[:, 2] > 0.0
    VdH = np.abs(np.einsum("ij,ij->i", Vl, H))
    f = np.zeros((V.shape[0], 3))
    pdf = np.zeros(V.shape[0])
    with np.errstate(all="ignore"):
        # diffuse
        LdH = np.einsum("ij,ij->i", Ll, H)
        Rr = 2.0 * rough * LdH * LdH
        FL, FV = schlick(Ll[:, 2]), schlick(Vl[:, 2])
        Fretro = Rr * (FL + FV + FL * FV * (Rr - 1.0))
        Fd = (1.0 - 0.5 * FL) * (1.0 - 0.5 * FV)
        Fss90 = 0.5 * Rr
        Fss = lerp(1.0, Fss90, FL) * lerp(1.0, Fss90, FV)
        ss = 1.25 * (Fss * (1.0 / (Ll[:, 2] + Vl[:, 2]) - 0.5) + 0.5)
        Fsheen = (schlick(LdH) * sheen)[:, None] * Csheen
        fd = base[None, :] / np.pi * lerp(Fd + Fretro, ss, subsurface)[:, None] + Fsheen
        m = (diff_pr > 0.0) & refl & up
        f += np.where(m[:, None], fd * diel_wt, 0.0)
        pdf += np.where(m, Ll[:, 2] / np.pi * diff_pr, 0.0)
        # microfacet reflection shared terms
        D, G1 = gtr2(H), smith_aniso(Vl)
        G2 = G1 * smith_aniso(Ll)
        mr_pdf = G1 * D / (4.0 * Vl[:, 2])
        mr = (D * G2 / (4.0 * Ll[:, 2] * Vl[:, 2]))[:, None]
        # dielectric reflection
        Fdi = np.where((F0 != 1.0) & (ior != 0.0), (fresnel(VdH, 1.0 / ior) - F0) / (1.0 - F0), 0.0)
        m = (diel_pr > 0.0) & refl & up
        f += np.where(m[:, None], lerp(Cspec0, 1.0, Fdi[:, None]) * mr * diel_wt, 0.0)
        pdf += np.where(m, mr_pdf * diel_pr, 0.0)
        # metal
        m = (metal_pr > 0.0) & refl & up
        f += np.where(m[:, None], lerp(base[None, :], 1.0, schlick(VdH)[:, None]) * mr * metal_wt, 0.0)
        pdf += np.where(m, mr_pdf * metal_pr, 0.0)
        # glass
        Fg = fresnel(VdH, eta)
        m = (glass_pr > 0.0) & refl & up
        f += np.where(m[:, None], Fg[:, None] * mr * glass_wt, 0.0)
        pdf += np.where(m, mr_pdf * glass_pr * Fg, 0.0)
        VH = np.einsum("ij,ij->i", Vl, H)
        denom = (LdH + VH * eta) ** 2
        jac = np.abs(LdH) / denom
        t_pdf = G1 * np.maximum(0.0, VH) * D * jac / Vl[:, 2]
        t_f = np.sqrt(base)[None, :] * ((1.0 - Fg) * D * G2 * np.abs(VH) * jac * eta * eta / np.abs(Ll[:, 2] * Vl[:, 2]))[:, None]
        m = (glass_pr > 0.0) & ~refl & (Ll[:, 2] < 0.0)
        f += np.where(m[:, None], t_f * glass_wt, 0.0)
        pdf += np.where(m, t_pdf * glass_pr * (1.0 - Fg), 0.0)
        # clearcoat
        a2 = cc_rough * cc_rough
        Dcc = np.full_like(VH, 1.0 / np.pi) if cc_rough >= 1.0 else (a2 - 1.0) / (np.pi * np.log(a2) * (1.0 + (a2 - 1.0) * H[:, 2] ** 2))
        Fcc = lerp(0.04, 1.0, schlick(VH))
        Gcc = smith(Ll[:, 2], 0.25) * smith(Vl[:, 2], 0.25)
        m = (cc_pr > 0.0) & refl & up
        f += np.where(m[:, None], (Fcc * Dcc * Gcc)[:, None] * 0.25 * clearcoat, 0.0)
        pdf += np.where(m, Dcc * H[:, 2] / (4.0 * VH) * cc_pr, 0.0)
    return f * np.abs(Ll[:, 2])[:, None], pdf


_POINTWISE_MATERIALS = [
    dict(color=(0.8, 0.6, 0.4, 1), roughness=0.7),
    dict(color=(0.8, 0.6, 0.4, 1), roughness=0.3, ior=1.6, specular_tint=0.7),
    dict(color=(0.9, 0.8, 0.7, 1), roughness=0.4, metallic=1.0),
    dict(color=(0.9, 0.8, 0.7, 1), roughness=0.5, metallic=1.0, anisotropic=0.7),
    dict(color=(0.5, 0.7, 0.9, 1), roughness=0.45, metallic=0.5, anisotropic=-0.5),
    dict(color=(0.7, 0.2, 0.2, 1), roughness=0.6, clearcoat=1.0, clearcoat_gloss=0.6),
    dict(color=(0.8, 0.5, 0.4, 1), roughness=0.8, sheen=0.9, sheen_tint=0.4, subsurface=0.6),
    dict(color=(0.9, 0.95, 1.0, 0.2), roughness=0.3, ior=1.5),                       # mostly glass
    dict(color=(0.9, 0.95, 1.0, 0.0), roughness=0.15, ior=1.33),                     # all glass
    dict(color=(0.6, 0.6, 0.6, 0.5), roughness=0.5, metallic=0.3, clearcoat=0.5, clearcoat_gloss=0.2, ior=1.8, sheen=0.3),
]


@pytest.mark.parametrize("mi", range(len(_POINTWISE_MATERIALS)))
def test_eval_brdf_pointwise_against_second_transcription(oracle, mi):
    import ctypes as C
    lib = oracle.load_oracle()
    md = np.ascontiguousarray(pack_material(**_POINTWISE_MATERIALS[mi]), dtype=F32)
    rng = np.random.RandomState(100 + mi)
    n = 4000
    L = rng.normal(size=(n, 3)); L /= np.linalg.norm(L, axis=-1, keepdims=True)
    L = np.ascontiguousarray(L.astype(F32))
    worst = 0.0
    for N, V in (((0.0, 0.0, 1.0), (0.3, 0.2, 0.9)), ((0.3, -0.5, 0.8), (0.7, 0.1, 0.4)), ((-0.2, 0.9, -0.4), (0.1, 0.8, 0.1))):
        N = np.ascontiguousarray((np.array(N, np.float64) / np.linalg.norm(N)).astype(F32))
        V = np.ascontiguousarray((np.array(V) / np.linalg.norm(V)).astype(F32))
        out = np.zeros((n, 4), F32)
        lib.oracle_eval_brdf_batch(md.ctypes.data, V.ctypes.data, N.ctypes.data, C.c_float(0.0), L.ctypes.data, C.c_uint64(n), out.ctypes.data)
        f64, p64 = _np_eval_brdf(md, np.broadcast_to(V.astype(np.float64), (n, 3)), N.astype(np.float64), L.astype(np.float64))
        got = out.astype(np.float64)
        # grazing configurations amplify fp32 rounding (1 / (4 Lz Vz), the refraction jacobian): compare where the value is well conditioned
        Lz = L.astype(np.float64) @ (N.astype(np.float64) / np.linalg.norm(N))
        ok = (np.abs(Lz) > 0.05) & np.isfinite(f64).all(axis=-1) & np.isfinite(p64)
        scale = np.maximum(np.abs(f64).max(axis=-1), 1e-3)
        err_f = np.abs(got[:, :3] - f64).max(axis=-1) / scale
        err_p = np.abs(got[:, 3] - p64) / np.maximum(np.abs(p64), 1e-3)
        assert ok.sum() > 0.9 * n
        worst = max(worst, err_f[ok].max(), err_p[ok].max())
        assert err_f[ok].max() < 2e-3 and err_p[ok].max() < 2e-3, (mi, err_f[ok].max(), err_p[ok].max(), int(np.argmax(np.where(ok, err_f, 0))))
        # and the typical agreement is at the level of fp32 rounding
        assert np.median(err_f[ok]) < 2e-6 and np.median(err_p[ok]) < 2e-6


# ---------------------------------------------------------------------------------------
# R3 + R13: SampleBRDF pointwise.  The PCG recurrence of util/random.hlsl:5-16 in uint32 numpy gives the three draws of every call
# (r1, r2 first, r3 after the lobe probabilities: util/brdf.hlsl:244-245,287); the lobe pick and the sampled direction follow in float64
# from util/brdf.hlsl:240-335 and util/sampling.hlsl:20-33,56-74,137-146 -- a second transcription again.  Compared with the direction the
# oracle draws for the same RNG chain.
# ---------------------------------------------------------------------------------------
def _pcg_chain(state, count):
    out = np.zeros(count, np.float64)
    s = np.uint32(state)
    with np.errstate(over="ignore"):
        for i in range(count):
            old = np.uint32(s + np.uint32(747796405) + np.uint32(2891336453))
            word = np.uint32((np.uint32(old >> np.uint32((old >> np.uint32(28)) + np.uint32(4))) ^ old) * np.uint32(277803737))
            s = np.uint32((word >> np.uint32(22)) ^ word)
            out[i] = float(np.float32(s)) / 4294967296.0          # (float)state / (float)0xffffffffu, the divisor rounds to 2^32
    return out


def _np_sample_dirs(md, V, N, r):
    """r: (n, 3) draws.  Returns world directions (n, 3), lobe ids (n,) and the distance of r3 to the nearest lobe boundary."""
    md = md.astype(np.float64)
    base, opacity = md[0:3], md[3]
    metallic, rough = md[8], max(md[9], 0.001)
    ior = min(max(md[11], 1.001), 2.0)
    aniso = min(max(md[13], -0.9), 0.9)
    spec_tint, clearcoat = md[15], md[19]
    cc_rough = 0.1 + md[20] * (0.001 - 0.1)
    spec_trans = 1.0 - min(max(opacity, 0.0), 1.0)
    aspect = np.sqrt(1.0 - aniso * 0.9)
    ax, ay = max(0.001, rough / aspect), max(0.001, rough * aspect)
    eta = 1.0 / ior if (-V @ N) < 0.0 else ior
    k = 1.0 / max(1.0 + N[2], 0.00001)
    a_, b_ = N[1] * k, N[1] * N[1] * k
    c_ = -N[0] * a_
    X = np.array([N[2] + b_, c_, -N[0]]); X /= np.linalg.norm(X)
    Y = np.array([c_, 1.0 - b_, -N[1]]); Y /= np.linalg.norm(Y)
    Vl = np.array([V @ X, V @ Y, V @ N])
    lum = lambda c: c[0] * 0.299 + c[1] * 0.587 + c[2] * 0.114
    lb = lum(base)
    ctint = base / lb if lb > 0 else np.ones(3)
    F0 = ((1.0 - eta) / (1.0 + eta)) ** 2
    Cspec0 = F0 * (1.0 + spec_tint * (ctint - 1.0))
    sw = min(max(1.0 - Vl[2], 0.0), 1.0) ** 5
    diel_wt, metal_wt, glass_wt = (1.0 - metallic) * (1.0 - spec_trans), metallic, (1.0 - metallic) * spec_trans
    pr = np.array([diel_wt * lb, diel_wt * lum(Cspec0 + sw * (1.0 - Cspec0)), metal_wt * lum(base + sw * (1.0 - base)), glass_wt, 0.25 * clearcoat])
    pr /= pr.sum()
    cdf = np.cumsum(pr)
    r1, r2, r3 = r[:, 0], r[:, 1], r[:, 2]
    lobe = np.where(r3 < cdf[0], 0, np.where(r3 < cdf[2], 1, np.where(r3 < cdf[3], 2, 3)))
    margin = np.abs(r3[:, None] - np.array([cdf[0], cdf[2], cdf[3]])[None, :]).min(axis=1)
    n = r.shape[0]
    with np.errstate(all="ignore"):
        # diffuse
        rr, ph = np.sqrt(r1), 2 * np.pi * r2
        dx, dy = rr * np.cos(ph), rr * np.sin(ph)
        Ld = np.stack([dx, dy, np.sqrt(np.maximum(0.0, 1.0 - dx * dx - dy * dy))], -1)
        # GGX VNDF
        Vh = np.array([ax * Vl[0], ay * Vl[1], Vl[2]]); Vh /= np.linalg.norm(Vh)
        lensq = Vh[0] ** 2 + Vh[1] ** 2
        T1 = np.array([-Vh[1], Vh[0], 0.0]) / np.sqrt(lensq) if lensq > 0 else np.array([1.0, 0.0, 0.0])
        T2 = np.cross(Vh, T1)
        t1, t2 = rr * np.cos(ph), rr * np.sin(ph)
        s = 0.5 * (1.0 + Vh[2])
        t2 = (1.0 - s) * np.sqrt(1.0 - t1 * t1) + s * t2
        Nh = t1[:, None] * T1 + t2[:, None] * T2 + np.sqrt(np.maximum(0.0, 1.0 - t1 * t1 - t2 * t2))[:, None] * Vh
        Hg = np.stack([ax * Nh[:, 0], ay * Nh[:, 1], np.maximum(0.0, Nh[:, 2])], -1)
        Hg /= np.linalg.norm(Hg, axis=-1, keepdims=True)
        VdH = Hg @ Vl
        refl = lambda H: 2.0 * (H @ Vl)[:, None] * H - Vl                           # reflect(-V, H)
        Lr = refl(Hg); Lr /= np.linalg.norm(Lr, axis=-1, keepdims=True)
        # glass: Fresnel picks reflection or refraction with the rescaled r3
        s2 = eta * eta * (1.0 - VdH * VdH)
        ct = np.sqrt(np.maximum(1.0 - s2, 0.0))
        ci = np.abs(VdH)
        Fg = np.where(s2 > 1.0, 1.0, 0.5 * (((eta * ct - ci) / (eta * ct + ci)) ** 2 + ((eta * ci - ct) / (eta * ci + ct)) ** 2))
        r3g = (r3 - cdf[2]) / (cdf[3] - cdf[2])
        Hs = np.where((Hg[:, 2] < 0.0)[:, None], -Hg, Hg)
        ndi = -(Hs @ Vl)                                                              # dot(n, i) with i = -V
        kk = 1.0 - eta * eta * (1.0 - ndi * ndi)
        Lt = np.where((kk < 0.0)[:, None], 0.0, eta * (-Vl)[None, :] - (eta * ndi + np.sqrt(np.maximum(kk, 0.0)))[:, None] * Hs)
        Lt /= np.linalg.norm(Lt, axis=-1, keepdims=True)
        Lgr = refl(Hs); Lgr /= np.linalg.norm(Lgr, axis=-1, keepdims=True)
        Lg = np.where((r3g < Fg)[:, None], Lgr, Lt)
        gmargin = np.abs(r3g - Fg)
        # clearcoat GTR1
        a = max(0.001, cc_rough); a2 = a * a
        cth = np.sqrt((1.0 - a2 ** (1.0 - r2)) / (1.0 - a2))
        sth = np.clip(np.sqrt(1.0 - cth * cth), 0.0, 1.0)
        phc = r1 * 2 * np.pi
        Hc = np.stack([sth * np.cos(phc), sth * np.sin(phc), cth], -1)
        Lc = refl(Hc); Lc /= np.linalg.norm(Lc, axis=-1, keepdims=True)
    Ll = np.where((lobe == 0)[:, None], Ld, np.where((lobe == 1)[:, None], Lr, np.where((lobe == 2)[:, None], Lg, Lc)))
    margin = np.where(lobe == 2, np.minimum(margin, gmargin), margin)
    return Ll[:, 0:1] * X + Ll[:, 1:2] * Y + Ll[:, 2:3] * N, lobe, margin


@pytest.mark.parametrize("mi", range(len(_POINTWISE_MATERIALS)))
def test_sample_brdf_pointwise_against_second_transcription(oracle, mi):
    import ctypes as C
    lib = oracle.load_oracle()
    md = np.ascontiguousarray(pack_material(**_POINTWISE_MATERIALS[mi]), dtype=F32)
    n, seed = 1500, 0xBEEF00 + mi
    draws = _pcg_chain(seed, 3 * n).reshape(n, 3)
    for Nv, Vv in (((0.0, 0.0, 1.0), (0.3, 0.2, 0.9)), ((0.3, -0.5, 0.8), (0.7, 0.1, 0.4))):
        N = np.ascontiguousarray((np.array(Nv, np.float64) / np.linalg.norm(Nv)).astype(F32))
        V = np.ascontiguousarray((np.array(Vv) / np.linalg.norm(Vv)).astype(F32))
        st = C.c_uint32(seed)
        Lo = np.zeros((n, 3), F32)
        fp = np.zeros((n, 4), F32)
        lib.oracle_sample_brdf_batch(md.ctypes.data, V.ctypes.data, N.ctypes.data, C.c_float(0.0), C.byref(st), C.c_uint64(n), Lo.ctypes.data, fp.ctypes.data)
        want, lobe, margin = _np_sample_dirs(md, V.astype(np.float64), N.astype(np.float64), draws)
        ok = (margin > 1e-4) & np.isfinite(want).all(axis=-1) & np.isfinite(Lo).all(axis=-1)      # a draw on a lobe boundary may fall either way in fp32
        assert ok.sum() > 0.97 * np.isfinite(Lo).all(axis=-1).sum()
        err = np.abs(Lo.astype(np.float64) - want).max(axis=-1)
        assert err[ok].max() < 5e-4 and np.median(err[ok]) < 1e-6, (mi, err[ok].max(), np.median(err[ok]), np.bincount(lobe, minlength=4))


# ---------------------------------------------------------------------------------------
# R7 + R10 + R11 + R12 end to end: a floor under a rectangle light, one bounce.  The reference adds the light twice without MIS --
# once through next-event estimation (uniform point on the rectangle, pdf = d^2 / (area |cos_l|), util/light.hlsl:7-23,96-104) and once
# when the BSDF-sampled ray hits it (util/pathtrace.hlsl:42-49) -- so the pixel converges to 2 x integral over the rectangle of
# Le f(V, w) cos_l / d^2 dA, with f from the float64 transcription above (it contains the cosine at the surface).
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("mat", [dict(color=(0.8, 0.7, 0.6, 1), roughness=0.8), dict(color=(0.7, 0.7, 0.8, 1), roughness=0.45, metallic=0.6)])
def test_rectangle_light_direct_lighting_closed_form(oracle, mat):
    h, size, Le = 1.5, (1.2, 0.8), np.array([6.0, 5.0, 4.0])
    light = scenes.pack_rect_light((0.2, h, 0.1), (1, 0, 0), (0, 0, 1), size, Le, rng=1e6)          # normal = cross(u, v) = (0, -1, 0): faces the floor
    s = _floor_scene(light)
    s.materials = np.stack([pack_material(**mat)])
    w = hh = 32
    img = _render(oracle, s, w, hh, spp=768, seed=31, max_bounces=1, russian_roulette=False)
    eye, d = _pixel_directions(s.camera, w, hh)
    P = eye + d * (-eye[1] / d[..., 1])[..., None]
    md = s.materials[0]
    N = np.array([0.0, 1.0, 0.0])
    g = 48
    us, vs = (np.arange(g) + 0.5) / g, (np.arange(g) + 0.5) / g
    U, Vg = np.meshgrid(us, vs)
    corner = np.array([0.2, h, 0.1]) - np.array([size[0], 0, 0]) * 0.5 - np.array([0, 0, size[1]]) * 0.5
    Q = corner + U[..., None] * np.array([size[0], 0, 0]) + Vg[..., None] * np.array([0, 0, size[1]])       # (g, g, 3) points on the light
    dA = size[0] * size[1] / (g * g)
    ys, xs = np.mgrid[12:20, 12:20]                                        # an 8 x 8 window under the light
    got, want = [], []
    for y, x in zip(ys.ravel(), xs.ravel()):
        to = Q.reshape(-1, 3) - P[y, x]
        dist = np.linalg.norm(to, axis=-1)
        wdir = to / dist[:, None]
        cos_l = wdir[:, 1]                                                 # dot(-w, light normal (0, -1, 0))
        f, _ = _np_eval_brdf(md, np.broadcast_to(-d[y, x], wdir.shape), N, wdir)
        want.append(2.0 * (Le[None, :] * f * (np.maximum(cos_l, 0.0) / dist ** 2)[:, None]).sum(axis=0) * dA)
        got.append(img[y, x, :3])
    got, want = np.array(got), np.array(want)
    ratio = got.mean(axis=0) / want.mean(axis=0)
    assert np.abs(ratio - 1.0).max() < 0.03, ratio
    assert np.abs(got / want - 1.0).mean() < 0.08                           # pixel by pixel: Monte-Carlo noise of 768 samples only


# ---------------------------------------------------------------------------------------
# R10 (environment branch) + R14 + R15: a floor under a uniform sky, one bounce.  Next-event estimation draws its direction with
# RandomCosineHemisphere (util/random.hlsl:34-41), whose last line reads sin(theta) * (cos(phi) x + sin(phi) y + cos(theta) z): the
# sine multiplies the z term too, and DirectLight normalises the result (util/light.hlsl:145) -- so the direction is
# normalize(cos(phi) x + sin(phi) y + sqrt(r) z), NOT cosine-distributed: its cosine to the normal is z = c / sqrt(1 + c^2) <= 1 / sqrt(2),
# c^2 uniform, density q(w) = z / (pi (1 - z^2)^2) on z <= 1 / sqrt(2) (no direction within 45 degrees of the normal is ever drawn).  It is
# weighted and divided with lightPdf = 1 / (4 pi) (util/light.hlsl:143-157); the BSDF-sampled ray that leaves the scene collects the sky
# with the power heuristic of (scatterPdf, 1 / (4 pi)) (util/pathtrace.hlsl:27-40, util/sky.hlsl:90-129).  Both expectations are
# hemisphere integrals of the float64 transcription's f and pdf:
#   E = integral[ q * w(1/4pi, p) * Li * f * 4 pi ] + integral[ w(p, 1/4pi) * Li * f ],   w(a, b) = a^2 / (a^2 + b^2).
# (A first version of this test assumed the cosine distribution the function's name promises and read 4 % low: the closed form caught
# the author's misreading, the restatement had the shader's parenthesis all along.)
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("mat", [dict(color=(0.8, 0.7, 0.6, 1), roughness=0.8), dict(color=(0.7, 0.7, 0.8, 1), roughness=0.5, metallic=0.5)])
def test_uniform_sky_one_bounce_closed_form(oracle, mat):
    sb = SoupBuilder()
    sb.quad((-40.0, 0.0, -40.0), (80.0, 0, 0), (0, 0, 80.0), (0, 1, 0), 1, 1, 0)
    verts, attrs = sb.finish()
    sky = np.array([0.9, 1.1, 1.4])
    cam = Camera(eye=(0, 3.0, -4.0), target=(0, 0, 0), vfov_deg=30.0)
    s = Scene("sky_floor", verts, attrs, np.stack([pack_material(**mat)]), NO_LIGHTS, NO_TEX, cam,
              environment_mode=0, environment_color=tuple(sky) + (1.0,), environment_intensity=1.0)
    w = h = 24
    img = _render(oracle, s, w, h, spp=1024, seed=77, max_bounces=1, russian_roulette=False)
    _, d = _pixel_directions(cam, w, h)
    N = np.array([0.0, 1.0, 0.0])
    nc, nphi = 400, 320                                                                 # fine in cos(theta): q ends in a step at 1 / sqrt(2)
    c = (np.arange(nc) + 0.5) / nc
    ph = (np.arange(nphi) + 0.5) / nphi * 2 * np.pi
    C, PH = np.meshgrid(c, ph, indexing="ij")
    sn = np.sqrt(1 - C * C)
    W = np.stack([sn * np.cos(PH), C, sn * np.sin(PH)], -1).reshape(-1, 3)          # upper hemisphere about +y
    dw = 2 * np.pi / (nc * nphi)
    lp = 1.0 / (4 * np.pi)
    got, want = [], []
    for y, x in ((8, 8), (8, 15), (12, 12), (15, 9), (16, 16)):
        assert d[y, x, 1] < 0                                                           # the pixel sees the floor
        f, p = _np_eval_brdf(s.materials[0], np.broadcast_to(-d[y, x], W.shape), N, W)
        w_nee = np.where(p > 0, lp * lp / (lp * lp + p * p), 0.0)
        w_bsdf = np.where(p > 0, p * p / (p * p + lp * lp), 0.0)
        z = W[:, 1]
        q = np.where(z <= np.sqrt(0.5), z / (np.pi * (1.0 - z * z) ** 2), 0.0)
        e_nee = (q * w_nee * 4 * np.pi)[:, None] * f
        e_bsdf = w_bsdf[:, None] * f
        want.append(sky * ((e_nee + e_bsdf).sum(axis=0) * dw))
        got.append(img[y - 1:y + 2, x - 1:x + 2, :3].mean(axis=(0, 1)))                # 3 x 3 pixels: the integrand varies slowly across them
    got, want = np.array(got), np.array(want)
    assert np.abs(got / want - 1.0).max() < 0.025, got / want


# ---------------------------------------------------------------------------------------
# R10 + R11 + R12: a floor under a point / spot light, one bounce.  No area to integrate: SampleOneLight reports pdf 0, EvalLight divides
# by 1 (util/light.hlsl:25-45,96-104), so a pixel is emission x range falloff x cone factor x f(V, direction) -- no inverse square
# law -- with f from the float64 transcription; the bounce ray leaves the scene into a black sky.
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind", ["point", "spot"])
def test_point_and_spot_light_closed_form(oracle, kind):
    pos, fwd, Le, rng_ = np.array([0.4, 1.8, -0.3]), np.array([0.1, -1.0, 0.2]), np.array([3.0, 2.5, 2.0]), 6.0
    outer, inner = 80.0, 50.0
    light = scenes.pack_point_light(pos, Le, rng=rng_) if kind == "point" else scenes.pack_spot_light(pos, fwd, outer, inner, Le, rng=rng_)
    s = _floor_scene(light)
    s.materials = np.stack([pack_material(color=(0.7, 0.6, 0.8, 1), roughness=0.55, metallic=0.3)])
    w = h = 96
    img = _render(oracle, s, w, h, spp=48, seed=8, max_bounces=1, russian_roulette=False)
    eye, d = _pixel_directions(s.camera, w, h)
    P = (eye + d * (-eye[1] / d[..., 1])[..., None]).reshape(-1, 3)
    Vv = (-d).reshape(-1, 3)
    to = pos - P
    dist = np.linalg.norm(to, axis=-1)
    wdir = to / dist[:, None]
    f, _ = _np_eval_brdf(s.materials[0], Vv, np.array([0.0, 1.0, 0.0]), wdir)
    r = dist / rng_
    atten = np.where(dist > rng_, 0.0, np.clip(1.0 / (1.0 + 25.0 * r * r) * np.clip((1.0 - r) * 5.0, 0.0, 1.0), 0.0, 1.0))
    cone = np.ones_like(dist)
    if kind == "spot":
        cos_t = (-wdir) @ (fwd / np.linalg.norm(fwd))
        c_out, c_in = np.cos(np.radians(outer) / 2), np.cos(np.radians(inner) / 2)
        cone = np.clip((cos_t - c_out) / (c_in - c_out), 0.0, 1.0)
        smooth = (np.abs(cos_t - c_out) > 0.03) & (np.abs(cos_t - c_in) > 0.03)             # away from the two kinks of the ramp
    else:
        smooth = np.ones_like(dist, bool)
    want = Le[None, :] * (atten * cone)[:, None] * f
    got = img[..., :3].reshape(-1, 3)
    lit = smooth & (want.max(axis=-1) > 1e-3)
    assert lit.sum() > 150
    rel = np.abs(got[lit] - want[lit]).max(axis=-1) / want[lit].max(axis=-1)
    assert np.median(rel) < 6e-3 and rel.max() < 0.10, (np.median(rel), rel.max())          # what is left is the pixel filter over a sloped integrand (the cone's ramp)
