"""BuildBVH / BuildTLAS of the drop-in are byte-identical to the reference plugin's output:
against committed golden dumps (made by tests/golden/make_golden.py from the reference compiled in place)
and, when oracle/_ref is present, against the reference plugin itself on larger and degenerate inputs."""
import glob
import os

import numpy as np
import pytest

from unity_webgpu_pathtracer_amd import abi, plugin, scenes

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "cwbvh_*.npz"))), ids=os.path.basename)
def test_cwbvh_golden(path):
    g = np.load(path)
    nodes, tris = plugin.build_cwbvh(g["vertices"])
    assert nodes.nbytes == g["nodes"].nbytes and tris.nbytes == g["tris"].nbytes
    assert np.array_equal(nodes, g["nodes"]), "CWBVH node bytes differ from the reference plugin's"
    assert np.array_equal(tris, g["tris"]), "CWBVH triangle bytes differ from the reference plugin's"


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "tlas_*.npz"))), ids=os.path.basename)
def test_tlas_golden(path):
    g = np.load(path)
    inst = g["instances"].view(abi.BLAS_INSTANCE)
    nodes, idx = plugin.build_tlas(inst)
    assert np.array_equal(nodes, g["nodes"]) and np.array_equal(idx, g["indices"])


def test_cornell_is_one_node():
    nodes, tris = plugin.build_cwbvh(scenes.cornell_box().vertices)
    assert nodes.nbytes == 80 and tris.nbytes == 8 * 48          # 8-tri Cornell -> 1 node + 384 B
    n = nodes.view(abi.CWBVH_NODE)[0]
    assert n["imask"] == 0                                        # no interior children
    prim = tris.view(np.uint32).reshape(-1, 12)[:, 11]
    assert sorted(prim.tolist()) == list(range(8))                # every primitive index appears once


def _soup(rng, n, spread, size):
    c = rng.uniform(-spread, spread, (n, 1, 3))
    v = c + rng.normal(0, size, (n, 3, 3))
    out = np.zeros((n * 3, 4), np.float32)
    out[:, :3] = v.reshape(-1, 3)
    return out


def test_against_reference_plugin_live(oracle):
    if oracle.load_ref_plugin() is None:
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    rng = np.random.RandomState(5)
    cases = [scenes.material_zoo().vertices, scenes.sponza_atrium(tex_size=4, detail=0.2).vertices,
             scenes.bunny_box(frequency=12).vertices, _soup(rng, 5000, 10, 0.3), _soup(rng, 2, 1, 1), _soup(rng, 7, 1, 1)]
    w = _soup(rng, 50, 5, 0.5)
    w[:, 3] = rng.uniform(-1, 1, w.shape[0])                      # w is ignored for bounds but flows into e1.w/e2.w
    cases.append(w)
    line = np.zeros((30, 4), np.float32)
    line[:, 0] = np.repeat(np.arange(10), 3) + np.tile([0, 0.5, 0.25], 10)   # degenerate: zero extent on two axes
    cases.append(line)
    for v in cases:
        rn, rt = oracle.ref_build_cwbvh(v)
        mn, mt = plugin.build_cwbvh(v)
        assert np.array_equal(rn, mn) and np.array_equal(rt, mt)
    for n in (1, 3, 64, 257):
        inst = np.zeros(n, abi.BLAS_INSTANCE)
        c = rng.uniform(-20, 20, (n, 3))
        e = rng.uniform(0.0, 3, (n, 3))
        inst["aabbMin"], inst["aabbMax"] = c - e, c + e
        rn, ri = oracle.ref_build_tlas(inst)
        mn, mi = plugin.build_tlas(inst)
        assert np.array_equal(rn, mn) and np.array_equal(ri, mi)


def test_benchmark_scene_sizes():
    """C3 = 8 + 20*59^2 triangles; C4 = 250,000 +- 1 %."""
    assert scenes.bunny_box().tri_count == 8 + 69620
    s = scenes.sponza_atrium(tex_size=4)
    assert abs(s.tri_count - 250000) <= 2500
    assert len(s.materials) == 24 and len(s.lights) == 2
