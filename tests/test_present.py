"""The presentation blit (Assets/Resources/Presentation.shader:36-73 + util/tonemap.hlsl): the oracle's restatement against an
independent float64 numpy formulation of the same operators, known answers, and (gpu) the HIP kernel against the oracle."""
import numpy as np
import pytest

from unity_webgpu_pathtracer_amd import abi, scenes


def _params(w, h, mode, srgb=False, exposure=1.0, brightness=1.0, contrast=1.0, saturation=1.0, vignette=0.0):
    q = abi.PTPresentParams()
    q.OutputWidth, q.OutputHeight, q.Mode, q.sRGB = w, h, mode, 1 if srgb else 0
    q.Exposure, q.Brightness, q.Contrast, q.Saturation, q.Vignette = exposure, brightness, contrast, saturation, vignette
    return q


def _numpy_present(q, frame):
    c = frame[..., :3].astype(np.float64) * q.Exposure
    if q.Mode == 1:
        mi = np.array([[0.59719, 0.35458, 0.04823], [0.07600, 0.90834, 0.01566], [0.02840, 0.13383, 0.83777]])
        mo = np.array([[1.60475, -0.53108, -0.07367], [-0.10208, 1.10813, -0.00605], [-0.00327, -0.07276, 1.07602]])
        c = c @ mi.T
        c = (c * (c + 0.0245786) - 0.000090537) / (c * (0.983729 * c + 0.4329510) + 0.238081)
        c = c @ mo.T
    elif q.Mode == 2:
        X = np.maximum(0.0, c - 0.004)
        c = ((X * (6.2 * X + 0.5)) / (X * (6.2 * X + 1.7) + 0.06)) ** 2.2
    elif q.Mode == 3:
        c = c / (1.0 + c)
    elif q.Mode == 4:
        a, d, hdr, mid_in, mid_out = 1.6, 0.977, 8.0, 0.18, 0.267
        b = (-mid_in ** a + hdr ** a * mid_out) / ((hdr ** (a * d) - mid_in ** (a * d)) * mid_out)
        cc = (hdr ** (a * d) * mid_in ** a - hdr ** a * mid_in ** (a * d) * mid_out) / ((hdr ** (a * d) - mid_in ** (a * d)) * mid_out)
        c = c ** a / (c ** (a * d) * b + cc)
    if q.sRGB:
        with np.errstate(invalid="ignore"):
            high = np.power(c, 1 / 2.4) * 1.055 - 0.055          # NaN for negative input, as HLSL pow
        c = np.where(c > 0.0031308, high, c * 12.92)
    c = np.clip(0.5 + q.Contrast * (c - 0.5), 0.0, 1.0)
    c = c ** (1.0 / q.Brightness)
    lum = (c * np.array([0.299, 0.587, 0.114])).sum(-1, keepdims=True)
    c = lum + q.Saturation * (c - lum)
    h, w = frame.shape[:2]
    u = ((np.arange(w) + 0.5) / w - 0.5) * 2.0
    v = ((np.arange(h) + 0.5) / h - 0.5) * 2.0
    c = c * (1.0 - (u[None, :] ** 2 + v[:, None] ** 2) * q.Vignette)[..., None]
    out = np.ones(frame.shape, dtype=np.float64)
    out[..., :3] = c
    return out


def _hdr_frame(w, h, seed=11):
    rng = np.random.RandomState(seed)
    f = np.ones((h, w, 4), dtype=np.float32)
    f[..., :3] = (rng.gamma(0.6, 1.5, (h, w, 3))).astype(np.float32)
    f[0, 0, :3] = 0.0                                               # black pixel: pow(0, .) paths
    f[0, 1, :3] = (40.0, 0.0, 0.001)                                # saturated primary: ACES goes negative in other channels
    return f


CASES = [(0, False, {}), (1, False, {}), (2, False, {}), (3, False, {}), (4, False, {}),
         (1, True, {}), (3, True, dict(exposure=1.7, brightness=1.3, contrast=1.2, saturation=0.6, vignette=0.35)),
         (4, True, dict(exposure=0.5, brightness=0.8, contrast=0.9, saturation=1.4, vignette=0.1))]


@pytest.mark.parametrize("mode,srgb,kw", CASES)
def test_oracle_present_matches_numpy(oracle, mode, srgb, kw):
    w, h = 48, 20
    f = _hdr_frame(w, h)
    q = _params(w, h, mode, srgb, **kw)
    got = oracle.present(q, f).astype(np.float64)
    ref = _numpy_present(q, f)
    both_nan = np.isnan(got) & np.isnan(ref)
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    err = np.abs(got - ref)[~both_nan]
    assert err.max() < 3e-5, err.max()
    assert (got[..., 3] == 1.0).all()


def test_golden_cases(oracle):
    """Regression pin of the restatement (tests/golden/make_golden.py)."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "present_cases.npz"))
    for i, (mode, srgb, kw) in enumerate(CASES):
        got = oracle.present(_params(24, 10, mode, srgb, **kw), g["frame"])
        assert np.array_equal(got.view(np.uint32), g[f"case{i}"].view(np.uint32)), i


def test_known_answers(oracle):
    f = np.ones((1, 4, 4), dtype=np.float32)
    f[0, :, :3] = np.array([[0.0] * 3, [0.25] * 3, [1.0] * 3, [3.0] * 3], dtype=np.float32)
    ident = oracle.present(_params(4, 1, 0), f)
    assert np.array_equal(ident[0, :3, :3], f[0, :3, :3]) and (ident[0, 3, :3] == 1.0).all()       # mode None: clamp only
    rein = oracle.present(_params(4, 1, 3), f)
    assert np.allclose(rein[0, :, 0], [0.0, 0.2, 0.5, 0.75], atol=1e-7)
    srgb = oracle.present(_params(4, 1, 0, srgb=True), f)
    assert abs(srgb[0, 1, 0] - 0.537099) < 1e-5 and srgb[0, 0, 0] == 0.0                            # LinearToSrgb(0.25)
    gray = oracle.present(_params(4, 1, 0, saturation=0.0), f * np.array([1, 0.5, 0.25, 1], np.float32))
    assert np.allclose(gray[0, 1, 0], gray[0, 1, 1]) and np.allclose(gray[0, 1, 1], gray[0, 1, 2])


@pytest.mark.gpu
@pytest.mark.parametrize("mode,srgb,kw", CASES)
def test_gpu_present_bit_exact(oracle, mode, srgb, kw):
    import torch
    from unity_webgpu_pathtracer_amd.pathtracer import PathTracer
    w, h = 200, 120
    f = _hdr_frame(w, h, seed=12)
    pt = PathTracer(scenes.cornell_box(), width=w, height=h)
    q = _params(w, h, mode, srgb, **kw)
    src = torch.from_numpy(f).cuda()
    dst = torch.zeros_like(src)
    torch.cuda.synchronize()
    pt.present_to(q, src.data_ptr(), dst.data_ptr())
    pt.synchronize()
    gpu = dst.cpu().numpy()
    ref = oracle.present(q, f)
    assert np.array_equal(gpu.view(np.uint32), ref.view(np.uint32))
    pt.close()


@pytest.mark.gpu
def test_gpu_present_of_rendered_frame(oracle):
    """PathTracer.cs:255-266: render, then blit the current Output frame through the tonemapper (inspector defaults: Lottes)."""
    from unity_webgpu_pathtracer_amd.pathtracer import PathTracer
    s = scenes.cornell_box()
    pt = PathTracer(s, width=96, height=64, samplesPerPass=2)
    p = pt.params(seed=77)
    pt.render_pass(p)
    linear = pt.readback()
    img = pt.present()
    assert np.array_equal(img.view(np.uint32), oracle.present(pt.present_params(), linear).view(np.uint32))
    assert img[..., :3].min() >= 0.0 and img[..., :3].max() <= 1.0
    pt.close()
