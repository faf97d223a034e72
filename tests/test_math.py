"""include/ptmi_math.h: the canonical elementary functions stay within a few ulp of libm, and the RNG
reproduces the known-answer vectors computed from util/random.hlsl:5-16."""
import ctypes as C

import numpy as np
import pytest


def _ulp_err(got, ref64):
    ref32 = ref64.astype(np.float32)
    ulp = np.spacing(np.abs(ref32)).astype(np.float64)
    return np.abs(got.astype(np.float64) - ref64) / np.maximum(ulp, 1e-300)


def _apply(lib, fn, x, y=None):
    y = np.zeros_like(x) if y is None else y
    return np.array([lib.oracle_math(fn, float(a), float(b)) for a, b in zip(x, y)], dtype=np.float32)


@pytest.fixture(scope="module")
def lib(oracle):
    return oracle.load_oracle()


def test_sin_cos(lib):
    rng = np.random.RandomState(1)
    x = rng.uniform(0.0, 6.3, 20000).astype(np.float32)          # the shader only passes angles in [0, 2pi]
    assert np.abs(_apply(lib, 0, x) - np.sin(x.astype(np.float64))).max() < 1.5e-7
    assert np.abs(_apply(lib, 1, x) - np.cos(x.astype(np.float64))).max() < 1.5e-7
    big = rng.uniform(-2000.0, 2000.0, 5000).astype(np.float32)
    assert np.abs(_apply(lib, 0, big) - np.sin(big.astype(np.float64))).max() < 2e-6


def test_log_exp_pow_acos(lib):
    rng = np.random.RandomState(2)
    u = np.exp(-rng.uniform(0, 80, 20000)).astype(np.float32)
    assert _ulp_err(_apply(lib, 2, u), np.log(u.astype(np.float64))).max() <= 2.0
    assert _ulp_err(_apply(lib, 3, u), np.log2(u.astype(np.float64))).max() <= 3.0
    e = rng.uniform(-125, 125, 20000).astype(np.float32)
    assert _ulp_err(_apply(lib, 4, e), np.exp2(e.astype(np.float64))).max() <= 3.0
    a = rng.uniform(-1, 1, 20000).astype(np.float32)
    assert _ulp_err(_apply(lib, 6, a), np.arccos(a.astype(np.float64))).max() <= 3.0
    b = rng.uniform(1e-6, 1.0, 5000).astype(np.float32)
    y = rng.uniform(0, 1, 5000).astype(np.float32)
    rel = np.abs(_apply(lib, 5, b, y).astype(np.float64) / np.power(b.astype(np.float64), y.astype(np.float64)) - 1)
    assert rel.max() < 5e-6                                        # pow = exp2(y*log2(x)), as HLSL defines it


def test_atan2_fmod(lib):
    """Environment-map lookups (util/sky.hlsl:46-53) need atan2 and fmod."""
    rng = np.random.RandomState(3)
    y = rng.normal(0, 1, 20000).astype(np.float32)
    x = rng.normal(0, 1, 20000).astype(np.float32)
    got = _apply(lib, 10, y, x)
    assert np.abs(got - np.arctan2(y.astype(np.float64), x.astype(np.float64))).max() < 4e-7
    f = lambda a, b: lib.oracle_math(10, a, b)
    assert f(0.0, 0.0) == 0.0 and f(1.0, 0.0) == np.float32(np.pi / 2) and f(-1.0, 0.0) == np.float32(-np.pi / 2)
    assert f(0.0, -1.0) == np.float32(np.pi) and f(0.0, 1.0) == 0.0 and np.isnan(f(float("nan"), 1.0))
    a = rng.uniform(-3, 3, 5000).astype(np.float32)
    m = _apply(lib, 11, a, np.ones_like(a))
    assert np.array_equal(m, np.fmod(a, np.float32(1.0)))           # HLSL fmod keeps the sign of x, like C fmod


def test_special_values(lib):
    f = lambda fn, x, y=0.0: lib.oracle_math(fn, x, y)
    assert f(2, 1.0) == 0.0 and f(4, 0.0) == 1.0 and f(6, 1.0) == 0.0
    assert f(2, 0.0) == -np.inf and np.isnan(f(2, -1.0)) and np.isnan(f(6, 1.5))
    assert f(5, 0.0, 0.5) == 0.0 and f(5, 0.25, 0.5) == 0.5 and f(5, 1.0, 2.2) == 1.0
    assert f(4, -130.0) > 0 and f(4, 200.0) == np.inf
    assert f(8, 4.0) == 2.0 and f(9, 4.0) == 0.25 and f(9, 0.0) == np.inf


KNOWN = {   # state after each call : float returned
    0x00000000: [(0xa8beea3c, 0.659163117), (0x0a2a1484, 0.039704591), (0x1e93be90, 0.119441897), (0x75134d09, 0.457325757)],
    0x00000001: [(0xb94dd992, 0.723844171), (0x7d3246cc, 0.489048421), (0xcb994a9c, 0.795307815), (0x4dd1f399, 0.303984851)],
    0x12345678: [(0x199714cb, 0.0999615639), (0xe91f039e, 0.910629511), (0x76c6835c, 0.463966578), (0xbc936622, 0.736624122)],
    0xffffffff: [(0x982ff7a5, 0.594481945), (0xbefdfcf3, 0.746063054), (0x37723005, 0.216586113), (0xd7a537c4, 0.842364788)],
}


def test_rng_known_answers(lib):
    for seed, seq in KNOWN.items():
        s = C.c_uint32(seed)
        for state, value in seq:
            f = lib.oracle_random_float(C.byref(s))
            assert s.value == state
            assert np.float32(f) == np.float32(value)


def test_rng_top_states_round_to_one(lib):
    """(float)0xFFFFFF80 / 2^32 == 1.0: RandomFloat is inclusive of 1 -- the reason the light
    pick is clamped and the reason NaN NEE directions exist (util/random.hlsl:34-41)."""
    assert np.float32(0xFFFFFF7F) / np.float32(4294967296.0) < 1.0
    assert np.float32(0xFFFFFF80) / np.float32(4294967296.0) == 1.0
    # per-pixel seed example of SURVEY.md Appendix D: W = 256, pixel (10, 3), CurrentSample 0
    assert (3 * 256 + 10) * (0 + 1) + 0x12345678 == 0x12345982


def test_texture_wrap_closed_form_equals_the_reference_loops(lib):
    """util/texture.hlsl:41-48 wraps with `while (u > 1) u -= 1; while (u < 0) u += 1`.  The kernels use the closed form
    pt_wrap01 (bounded cost, no hang for huge / infinite uv); it must give the literal loops' result bit for bit."""
    rng = np.random.RandomState(5)
    vals = np.concatenate([
        rng.uniform(-8, 8, 4000), rng.uniform(-3000, 3000, 3000), rng.normal(0, 1e-6, 500), rng.uniform(-2e5, 2e5, 300),
        np.array([0.0, -0.0, 1.0, -1.0, 2.0, -2.0, 1.0000001, -1.0000001, 0.99999994, -1e-10, 1e-10, 3.5, -2.5, 4096.0, -4096.0,
                  16777215.0, -16777215.0, 16777216.0, -16777216.0, 3e7, -3e7, 1e30, float("inf"), float("-inf")]),
        np.arange(-40, 40) + np.float32(2 ** -23), np.arange(-40, 40) - np.float32(2 ** -23),
    ]).astype(np.float32)
    closed = _apply(lib, 12, vals)
    loops = _apply(lib, 13, vals)
    assert np.array_equal(closed.view(np.uint32), loops.view(np.uint32))
    assert ((closed >= 0) & (closed <= 1)).all()
    assert np.isnan(lib.oracle_math(12, float("nan"), 0.0)) and np.isnan(lib.oracle_math(13, float("nan"), 0.0))


def test_unorm8_equals_division_by_255_for_every_byte(lib):
    """The kernels turn a texel channel into a float with pt_unorm8 (mul + 2 fma); the oracle divides by 255 as
    util/texture.hlsl does.  All 256 inputs, bit for bit."""
    b = np.arange(256, dtype=np.float32)
    fast = _apply(lib, 14, b)
    div = _apply(lib, 15, b)
    assert np.array_equal(fast.view(np.uint32), div.view(np.uint32))
    assert np.array_equal(div, (b / np.float32(255.0)).astype(np.float32))
