"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle on identical scenes, params and seeds.

Tolerance: north_star asks for accumulation-buffer L-inf < 1e-4.  Because both sides evaluate the canonical
fp32 arithmetic of include/ptmi_math.h the expected difference is exactly 0; the tests assert the 1e-4 bound
and additionally require every pixel to be bit-identical at these sizes.
"""
import numpy as np
import pytest

from unity_webgpu_pathtracer_amd import scenes
from unity_webgpu_pathtracer_amd.pathtracer import PathTracer

pytestmark = pytest.mark.gpu

LINF_TOL = 1e-4


def _oracle_frame(oracle, pt, p, accumulated=None, **kw):
    b = oracle.buffers_from_bvhscene(pt._bvhScene)
    return oracle.render(b, p, accumulated=accumulated, **kw)


def _compare(gpu, ref, what):
    assert gpu.shape == ref.shape
    assert not np.isnan(gpu).any(), f"{what}: NaN in GPU frame"
    diff = np.abs(gpu.astype(np.float64) - ref.astype(np.float64))
    linf = float(diff.max())
    nbad = int((gpu.view(np.uint32) != ref.view(np.uint32)).any(axis=-1).sum())
    print(f"[parity] {what}: Linf = {linf:.3e}, pixels not bit-identical = {nbad} / {gpu.shape[0] * gpu.shape[1]}")
    assert linf < LINF_TOL, f"{what}: Linf {linf} >= {LINF_TOL}"
    return linf, nbad


def _stats_equal(gpu_stats, ref_stats, fields):
    g, r = gpu_stats.as_dict(), ref_stats.as_dict()
    bad = {k: (g[k], r[k]) for k in fields if g[k] != r[k]}
    assert not bad, f"counter mismatch (gpu, oracle): {bad}"


ALL_COUNTERS = ["paths", "closestHitRays", "shadowRays", "nodeVisits", "triTests", "attrFetches", "materialFetches",
                "lightFetches", "texelFetches", "texDescriptorFetches", "pixelsWritten", "pixelsRead", "maxStackDepth",
                "stackOverflows", "tlasNodeVisits", "instanceVisits"]


SCHEDULES = [0, 1, 2, 3, 4]    # 4 fused persistent wavefront; 0 megakernel, 1 wavefront + refill trace, 2 wavefront one ray per lane, 3 wavefront + persistent dynamic-chunk trace


@pytest.mark.parametrize("schedule", SCHEDULES)
def test_c1_cornell_256_1spp(oracle, schedule):
    """BASELINE.json configs[0]: Cornell box 256x256, 1 spp, depth 4."""
    s = scenes.cornell_box()
    pt = PathTracer(s, width=256, height=256, samplesPerPass=1, schedule=schedule)
    pt.set_stats_level(1)
    p = pt.params(seed=0x12345678)
    pt.render_pass(p)
    gpu = pt.readback()
    ref, st = _oracle_frame(oracle, pt, p, shadow_any_hit=True)
    _, nbad = _compare(gpu, ref, "C1 cornell 256x256 1spp")
    assert nbad == 0
    _stats_equal(pt.stats(), st, ALL_COUNTERS)
    pt.close()


@pytest.mark.parametrize("schedule", SCHEDULES)
@pytest.mark.parametrize("name,kw,w,h,spp", [
    ("zoo", {}, 160, 120, 4),
    ("zoo", {"basic_sky": True}, 96, 64, 2),
    ("sponza", {"tex_size": 32, "detail": 0.15}, 160, 96, 2),
    ("furnace", {}, 64, 64, 4),
    ("zoo_env", {"env_map": (64, 32)}, 128, 80, 3),          # HAS_ENVIRONMENT_TEXTURE: importance-sampled lat-long map
])
def test_small_scenes_bit_exact(oracle, name, kw, w, h, spp, schedule):
    s = scenes.make_scene(name, **kw)
    pt = PathTracer(s, width=w, height=h, samplesPerPass=spp, schedule=schedule)
    pt.set_stats_level(1)
    p = pt.params(seed=0xC0FFEE)
    pt.render_pass(p)
    gpu = pt.readback()
    ref, st = _oracle_frame(oracle, pt, p, shadow_any_hit=True)
    _, nbad = _compare(gpu, ref, f"{name} {w}x{h} {spp}spp")
    assert nbad == 0
    _stats_equal(pt.stats(), st, ALL_COUNTERS)
    pt.close()


@pytest.mark.parametrize("schedule", SCHEDULES)
def test_progressive_accumulation(oracle, schedule):
    """Two passes through the C# host contract (PathTracer.cs:268-272): pass 2 reads pass 1 as AccumulatedOutput."""
    s = scenes.cornell_box()
    pt = PathTracer(s, width=128, height=128, samplesPerPass=4, schedule=schedule)
    seeds = [0x12345678, (0x12345678 + 0x9E3779B9) & 0xFFFFFFFF]
    p0 = pt.params(seeds[0])
    pt.OnRenderImage(seeds[0])
    f0 = pt.readback()
    p1 = pt.params(seeds[1])
    assert p1.CurrentSample == 4
    pt.OnRenderImage(seeds[1])
    f1 = pt.readback()
    r0, _ = _oracle_frame(oracle, pt, p0)
    r1, _ = _oracle_frame(oracle, pt, p1, accumulated=r0)
    _compare(f0, r0, "progressive pass 0")
    _, nbad = _compare(f1, r1, "progressive pass 1")
    assert nbad == 0
    pt.close()


@pytest.mark.parametrize("schedule", SCHEDULES)
def test_firefly_no_rr_depth1(oracle, schedule):
    s = scenes.material_zoo()
    pt = PathTracer(s, width=96, height=64, samplesPerPass=2, maxRayBounces=1, useRussianRoulette=False,
                    fireflyFilter=True, maxFireflyLuminance=2.0, schedule=schedule)
    p = pt.params(seed=7)
    pt.render_pass(p)
    gpu = pt.readback()
    ref, _ = _oracle_frame(oracle, pt, p)
    _, nbad = _compare(gpu, ref, "firefly/noRR/depth1")
    assert nbad == 0
    pt.close()


@pytest.mark.parametrize("schedule", SCHEDULES)
def test_reference_dispatch_edge_pixels(oracle, schedule):
    """Sizes that are not multiples of 8: ceil coverage renders everything; the C# host's floor-division
    dispatch (PathTracer.cs:207-208) leaves the edge pixels untouched."""
    s = scenes.cornell_box()
    for ref_dispatch in (False, True):
        pt = PathTracer(s, width=100, height=70, samplesPerPass=1, reference_dispatch=ref_dispatch, schedule=schedule)
        p = pt.params(seed=99)
        pt.render_pass(p)
        gpu = pt.readback()
        ref, _ = _oracle_frame(oracle, pt, p)
        _, nbad = _compare(gpu, ref, f"100x70 reference_dispatch={ref_dispatch}")
        assert nbad == 0
        if ref_dispatch:
            assert (gpu[:, 96:] == 0).all() and (gpu[64:, :] == 0).all()
        pt.close()


@pytest.mark.parametrize("schedule", SCHEDULES)
def test_tile_ownership_sums_to_full_frame(oracle, schedule):
    """PTSetTileOwnership: the per-rank partial frames sum (exact zeros elsewhere) to the single-GPU frame."""
    s = scenes.material_zoo()
    w, h, spp = 112, 80, 2
    full = PathTracer(s, width=w, height=h, samplesPerPass=spp)
    p = full.params(seed=1234)
    full.render_pass(p)
    f = full.readback()
    full.close()
    world = 3
    acc = np.zeros_like(f)
    for rank in range(world):
        pt = PathTracer(s, width=w, height=h, samplesPerPass=spp, rank=rank, world_size=world, schedule=schedule)
        pt.render_pass(p)
        part = pt.readback()
        ref, _ = _oracle_frame(oracle, pt, p, tile_rank=rank, tile_world=world)
        assert np.array_equal(part.view(np.uint32), ref.view(np.uint32))
        acc += part
        pt.close()
    assert np.array_equal(acc.view(np.uint32), f.view(np.uint32))


@pytest.mark.parametrize("schedule", SCHEDULES)
def test_nan_nee_ray_is_a_cheap_miss(oracle, schedule):
    """RandomFloat == 1.0 at the environment-NEE draw -> NaN shadow ray (util/random.hlsl:34-41).  The reference walks
    the whole tree for it; the kernels return the certain miss at once.  Frame identical to the faithful oracle,
    counters identical to the oracle's counting variant."""
    s = scenes.sponza_atrium(tex_size=4, detail=0.15)
    pt = PathTracer(s, width=8, height=8, samplesPerPass=1, schedule=schedule)
    pt.set_stats_level(1)
    p = pt.params(seed=0xA79B11)        # see tests/test_oracle.py::test_nan_nee_ray_counting_variant
    pt.render_pass(p)
    gpu = pt.readback()
    faithful, sf = _oracle_frame(oracle, pt, p)
    fast, sq = _oracle_frame(oracle, pt, p, shadow_any_hit=True)
    assert np.array_equal(gpu.view(np.uint32), faithful.view(np.uint32))
    _stats_equal(pt.stats(), sq, ALL_COUNTERS)
    assert sf.nodeVisits > sq.nodeVisits + 1000
    pt.close()


@pytest.mark.parametrize("iterations", [1, 3, 7])
def test_wavefront_cleanup_kernel_finishes_any_path(oracle, iterations):
    """With too few wavefront iterations almost every pixel is finished by the cleanup kernel (the megakernel loop
    resumed from slot state).  Frame and counters must not change."""
    s = scenes.material_zoo()
    pt = PathTracer(s, width=96, height=72, samplesPerPass=3, schedule=1)
    pt.set_wavefront_iterations(iterations)
    pt.set_stats_level(1)
    p = pt.params(seed=0xBEEF)
    pt.render_pass(p)
    gpu = pt.readback()
    ref, st = _oracle_frame(oracle, pt, p, shadow_any_hit=True)
    _, nbad = _compare(gpu, ref, f"cleanup after {iterations} iterations")
    assert nbad == 0
    _stats_equal(pt.stats(), st, ALL_COUNTERS)
    pt.close()


@pytest.mark.parametrize("schedule", [0, 1, 4])
def test_pipelined_passes_without_host_sync(oracle, schedule):
    """Six progressive passes issued back to back (the wavefront schedule overlaps them on separate state sets and
    streams; only the resolves are ordered).  The final frame must equal six sequential oracle passes bit for bit."""
    s = scenes.cornell_box()
    pt = PathTracer(s, width=160, height=96, samplesPerPass=2, schedule=schedule)
    seeds = [(0x12345678 + k * 0x9E3779B9) & 0xFFFFFFFF for k in range(6)]
    params = []
    for k in range(6):
        params.append(pt.params(seeds[k]))
        pt.OnRenderImage(seeds[k])            # no readback / synchronize in between
    gpu = pt.readback()
    ref = None
    for p in params:
        ref, _ = _oracle_frame(oracle, pt, p, accumulated=ref)
    _, nbad = _compare(gpu, ref, f"6 pipelined passes, schedule {schedule}")
    assert nbad == 0
    pt.close()


@pytest.mark.parametrize("schedule", [0, 1, 4])
def test_pass_batch_equals_sequential_passes(schedule):
    """PTRenderPassBatchTo: k progressive passes as ONE launch sequence (slots = passes x pixels; the resolve kernel applies the
    running mean pass by pass) against the same passes issued one by one with the caller ping-ponging frames: identical bits,
    identical ray counters.  Also across a batch boundary (the second batch starts at CurrentSample > 0) and with tile ownership."""
    import torch
    s = scenes.material_zoo()
    W, H, SPP = 136, 88, 2                                        # not multiples of 16: partially covered edge blocks
    seeds = [(0x12345678 + k * 0x9E3779B9) & 0xFFFFFFFF for k in range(7)]
    for rank, world in ((0, 1), (1, 3)):
        pt = PathTracer(s, width=W, height=H, samplesPerPass=SPP, schedule=schedule, rank=rank, world_size=world)
        frames = [torch.zeros((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(2)]
        params = [scenes.frame_params(s, W, H, spp=SPP, current_sample=k * SPP, seed=seeds[k]) for k in range(7)]
        cur = 0
        for k in range(7):                                           # reference: one pass per call
            pt.render_pass_to(params[k], frames[cur].data_ptr(), frames[1 - cur].data_ptr() if k > 0 else 0)
            cur = 1 - cur
        pt.synchronize()
        ref = frames[1 - cur].cpu().numpy().copy()
        st_ref = pt.stats()
        pt.reset_stats()
        a = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
        b = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
        pt.render_batch_to(params[0:3], a.data_ptr(), 0)             # passes 0-2
        pt.render_batch_to(params[3:4], b.data_ptr(), a.data_ptr())  # a batch of one
        if schedule == 1:
            pt.set_sub_frames(2)                                      # batches and sub-frames together: 2 launch sequences x 3 passes
        pt.render_batch_to(params[4:7], a.data_ptr(), b.data_ptr())  # passes 4-6, accumulated frame given
        pt.synchronize()
        got = a.cpu().numpy()
        st = pt.stats()
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (schedule, rank, world)
        assert (st.rays, st.paths, st.closestHitRays, st.shadowRays) == (st_ref.rays, st_ref.paths, st_ref.closestHitRays, st_ref.shadowRays)
        assert st.pixelsWritten == st_ref.pixelsWritten and st.pixelsRead == st_ref.pixelsRead
        # passes that differ in anything but RngSeedRoot / CurrentSample are refused
        bad = [params[0], scenes.frame_params(s, W, H, spp=SPP, current_sample=SPP, seed=1, max_bounces=2)]
        with pytest.raises(Exception):
            pt.render_batch_to(bad, a.data_ptr(), 0)
        pt.close()


@pytest.mark.parametrize("schedule,world", [(1, 1), (4, 1), (1, 3)])
def test_sub_frames_do_not_change_a_bit(oracle, schedule, world):
    """PTSetSubFrames: every pass cut into three interleaved sub-frames with their own launch sequences, state sets and streams,
    all writing one output frame -- two progressive passes equal the oracle's bit for bit; with tile ownership the pixels of other
    ranks stay exact zeros."""
    s = scenes.material_zoo()
    W, H, SPP = 200, 120, 2
    rank = world - 1
    pt = PathTracer(s, width=W, height=H, samplesPerPass=SPP, schedule=schedule, rank=rank, world_size=world)
    pt.set_sub_frames(3)
    seeds = [0xABCDEF01, 0x10FEDCBA]
    params, ref = [], None
    for seed in seeds:
        params.append(pt.params(seed))
        pt.OnRenderImage(seed)
    gpu = pt.readback()
    for p in params:
        ref, st = _oracle_frame(oracle, pt, p, accumulated=ref, tile_rank=rank, tile_world=world)
    _, nbad = _compare(gpu, ref, f"sub-frames, schedule {schedule}, world {world}")
    assert nbad == 0
    pt.close()


def test_render_pass_to_caller_buffers(oracle):
    """PTRenderPassTo with caller-owned device frames (torch tensors), ping-ponged by the caller as bench.py does."""
    import torch
    s = scenes.cornell_box()
    w, h = 128, 80
    pt = PathTracer(s, width=w, height=h, samplesPerPass=2)
    frames = [torch.zeros((h, w, 4), dtype=torch.float32, device="cuda") for _ in range(2)]
    torch.cuda.synchronize()
    ref, cur = None, 0
    for k in range(4):
        p = scenes.frame_params(s, w, h, spp=2, current_sample=2 * k, seed=1000 + k)
        pt.render_pass_to(p, frames[cur].data_ptr(), frames[1 - cur].data_ptr() if k else 0)
        ref, _ = _oracle_frame(oracle, pt, p, accumulated=ref)
        cur = 1 - cur
    pt.synchronize()
    gpu = frames[1 - cur].cpu().numpy()
    _, nbad = _compare(gpu, ref, "PTRenderPassTo x4")
    assert nbad == 0
    pt.close()


def test_plain_c_host(tmp_path, oracle):
    """host/pt_host.c: the plain-C host renders the Cornell box through the C-ABI, two progressive passes.  It dumps every byte
    it handed across the boundary (geometry, attributes, materials, light, the CWBVH it got back, each pass's PTFrameParams);
    the oracle renders exactly those inputs and the frames must be bit-identical."""
    import ctypes as C
    import os
    import subprocess
    from unity_webgpu_pathtracer_amd import abi
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", os.path.join(root, "host")], stdout=subprocess.DEVNULL)
    out, dump = str(tmp_path / "frame.f32"), str(tmp_path / "abi_dump.bin")
    subprocess.check_call([os.path.join(root, "host", "pt_host"), "64", "48", "2", "2", out, dump], cwd=root, stdout=subprocess.DEVNULL)
    gpu = np.fromfile(out, dtype=np.float32).reshape(48, 64, 4)
    raw = open(dump, "rb").read()
    hdr = np.frombuffer(raw, np.uint32, 8)
    assert hdr[0] == 0x44494250 and hdr[7] == C.sizeof(abi.PTFrameParams)
    nodes_b, tris_b, ntri, nmat, nlight, passes = (int(x) for x in hdr[1:7])
    off = 32

    def take(nbytes, dtype):
        nonlocal off
        a = np.frombuffer(raw, dtype, nbytes // np.dtype(dtype).itemsize, off).copy()
        off += nbytes
        return a
    verts = take(ntri * 3 * 16, np.float32).reshape(-1, 4)
    attrs = take(ntri * 128, abi.TRI_ATTR)
    mats = take(nmat * 128, np.float32).reshape(nmat, 32)
    lights = take(nlight * 64, np.float32).reshape(nlight, 16)
    nodes, tris = take(nodes_b, np.uint8), take(tris_b, np.uint8)
    ref_scene = scenes.cornell_box()
    s = scenes.Scene("c_host_cornell", verts, attrs, mats, lights, np.zeros(0, np.uint32), ref_scene.camera)
    b = oracle.SceneBuffers(s, nodes, tris)
    ref = None
    for k in range(passes):
        p = abi.PTFrameParams.from_buffer_copy(raw, off)
        off += C.sizeof(abi.PTFrameParams)
        assert p.structSize == C.sizeof(abi.PTFrameParams) and p.CurrentSample == 2 * k
        ref, _ = oracle.render(b, p, accumulated=ref)
    assert off == len(raw)
    assert np.array_equal(gpu.view(np.uint32), ref.view(np.uint32))
    # the same C host, one process driving "two devices" (the same GPU twice) through PTCreateMulti / PTGroup*: same bits
    out2 = str(tmp_path / "frame_group.f32")
    subprocess.check_call([os.path.join(root, "host", "pt_host"), "64", "48", "2", "2", out2], cwd=root, stdout=subprocess.DEVNULL,
                          env=dict(os.environ, PT_HOST_DEVICES="0,0"))
    assert np.array_equal(np.fromfile(out2, dtype=np.float32).view(np.uint32), gpu.reshape(-1).view(np.uint32))
    # and the C host's scene is the Python generator's Cornell box up to fp32-vs-fp64 rounding of the host-side setup
    assert np.allclose(mats, ref_scene.materials, rtol=1e-5) and np.allclose(verts, ref_scene.vertices)


@pytest.mark.parametrize("schedule", SCHEDULES)
def test_tlas_instanced_scene(oracle, schedule):
    """HAS_TLAS (util/tlas.hlsl): 15 instances of 4 meshes, rotations and non-uniform scales, two-level traversal.
    Frame and every counter (TLAS nodes, instance visits included) equal the oracle's."""
    s = scenes.instanced_scene()
    pt = PathTracer(s, width=144, height=96, samplesPerPass=3, schedule=schedule)
    pt.set_stats_level(1)
    p = pt.params(seed=0x7EA5)
    pt.render_pass(p)
    gpu = pt.readback()
    ref, st = _oracle_frame(oracle, pt, p, shadow_any_hit=True)
    _, nbad = _compare(gpu, ref, f"instanced (HAS_TLAS), schedule {schedule}")
    assert nbad == 0
    _stats_equal(pt.stats(), st, ALL_COUNTERS)
    assert st.tlasNodeVisits > 0 and st.instanceVisits > 0
    pt.close()


def test_tlas_progressive_and_cleanup(oracle):
    s = scenes.instanced_scene(count=9, detail=8)
    pt = PathTracer(s, width=96, height=64, samplesPerPass=2, schedule=1)
    pt.set_wavefront_iterations(3)
    seeds = [11, 22, 33]
    params, ref = [], None
    for k in range(3):
        params.append(pt.params(seeds[k]))
        pt.OnRenderImage(seeds[k])
    gpu = pt.readback()
    for p in params:
        ref, _ = _oracle_frame(oracle, pt, p, accumulated=ref)
    _, nbad = _compare(gpu, ref, "instanced, 3 pipelined passes, cleanup kernel")
    assert nbad == 0
    pt.close()


@pytest.mark.parametrize("schedule", SCHEDULES)
@pytest.mark.parametrize("w,h,spp,bounces", [(1, 1, 3, 4), (7, 5, 2, 1), (17, 33, 1, 9), (257, 3, 2, 4)])
def test_ragged_frame_sizes_and_depths(oracle, w, h, spp, bounces, schedule):
    """Frames far from the 16x16 block / 8x8 tile grid, one-pixel frames, depth 1 and depth 9."""
    s = scenes.material_zoo()
    pt = PathTracer(s, width=w, height=h, samplesPerPass=spp, maxRayBounces=bounces, schedule=schedule)
    pt.set_stats_level(1)
    p = pt.params(seed=0x5EED + w)
    pt.render_pass(p)
    gpu = pt.readback()
    ref, st = _oracle_frame(oracle, pt, p, shadow_any_hit=True)
    _, nbad = _compare(gpu, ref, f"{w}x{h} spp {spp} depth {bounces}")
    assert nbad == 0
    _stats_equal(pt.stats(), st, ALL_COUNTERS)
    pt.close()


@pytest.mark.parametrize("world", [2, 5, 8])
def test_tile_ownership_other_world_sizes(oracle, world):
    """Every pixel is rendered by exactly one rank for world sizes that do and do not divide the block grid."""
    s = scenes.cornell_box()
    w, h, spp = 200, 120, 1
    full = PathTracer(s, width=w, height=h, samplesPerPass=spp)
    p = full.params(seed=4321)
    full.render_pass(p)
    f = full.readback()
    full.close()
    acc = np.zeros_like(f)
    covered = np.zeros((h, w), dtype=np.int32)
    for rank in range(world):
        pt = PathTracer(s, width=w, height=h, samplesPerPass=spp, rank=rank, world_size=world)
        pt.render_pass(p)
        part = pt.readback()
        covered += (part[..., 3] != 0).astype(np.int32)          # alpha is 1 on rendered pixels, 0 elsewhere
        acc += part
        pt.close()
    assert (covered == 1).all()
    assert np.array_equal(acc.view(np.uint32), f.view(np.uint32))


def test_full_size_frame_windows_and_properties(oracle):
    """BASELINE.json's headline configuration at FULL size (Sponza-class, 1920x1080, 8 spp, depth 4, two progressive passes):
    windows of the frame against the oracle bit for bit, plus size-independent properties of the whole frame (determinism,
    alpha, finiteness, per-pass ray budget)."""
    s = scenes.sponza_atrium(tex_size=64)                          # full geometry (250,752 triangles); small textures keep the test light
    W, H, SPP = 1920, 1080, 8
    pt = PathTracer(s, width=W, height=H, samplesPerPass=SPP)
    seeds = [0x12345678, (0x12345678 + 0x9E3779B9) & 0xFFFFFFFF]
    p0 = pt.params(seeds[0])
    pt.OnRenderImage(seeds[0])
    f0 = pt.readback()
    p1 = pt.params(seeds[1])
    pt.OnRenderImage(seeds[1])
    f1 = pt.readback()
    st = pt.stats()
    assert np.isfinite(f1).all() and (f1[..., 3] == 1.0).all() and (f1[..., :3] >= 0).all()
    assert st.paths == 2 * W * H * SPP and st.pixelsWritten == 2 * W * H and st.pixelsRead == W * H
    assert st.closestHitRays <= st.paths * 5 and st.shadowRays <= st.closestHitRays * 2        # <= 5 bounce rays, <= 2 NEE rays per hit
    b = oracle.buffers_from_bvhscene(pt._bvhScene)
    windows = [(0, 0, 24, 12), (948, 534, 972, 546), (1896, 1068, 1920, 1080), (300, 900, 324, 912), (1500, 200, 1524, 212), (700, 40, 724, 52)]
    for (x0, y0, x1, y1) in windows:
        r0, _ = oracle.render(b, p0, window=(x0, y0, x1, y1))
        r1, _ = oracle.render(b, p1, accumulated=f0, window=(x0, y0, x1, y1))     # pass 2 reads pass 1 only at its own pixels
        assert np.array_equal(f0[y0:y1, x0:x1].view(np.uint32), r0[y0:y1, x0:x1].view(np.uint32)), (x0, y0)
        assert np.array_equal(f1[y0:y1, x0:x1].view(np.uint32), r1[y0:y1, x0:x1].view(np.uint32)), (x0, y0)
    # determinism: the same two passes again give the same bits, whatever the scheduling of waves and passes was
    pt.Reset()
    pt.OnRenderImage(seeds[0])
    pt.OnRenderImage(seeds[1])
    assert np.array_equal(pt.readback().view(np.uint32), f1.view(np.uint32))
    pt.close()


def _windows_vs_oracle(oracle, scene, W, H, SPP, windows, seeds, bounces=4):
    """Renders len(seeds) progressive passes of `scene` at W x H x SPP through the C-ABI and compares `windows` of every pass
    with the oracle bit for bit (each pass reading the GPU's previous frame as AccumulatedOutput at its own pixels).  Returns
    the frames and the stats."""
    pt = PathTracer(scene, width=W, height=H, samplesPerPass=SPP, maxRayBounces=bounces)
    b = oracle.buffers_from_bvhscene(pt._bvhScene)
    frames, params = [], []
    for seed in seeds:
        params.append(pt.params(seed))
        pt.OnRenderImage(seed)
        frames.append(pt.readback())
    st = pt.stats()
    for k, (p, f) in enumerate(zip(params, frames)):
        assert np.isfinite(f).all() and (f[..., 3] == 1.0).all() and (f[..., :3] >= 0).all()
        for (x0, y0, x1, y1) in windows:
            r, _ = oracle.render(b, p, accumulated=frames[k - 1] if k else None, window=(x0, y0, x1, y1))
            assert np.array_equal(f[y0:y1, x0:x1].view(np.uint32), r[y0:y1, x0:x1].view(np.uint32)), (scene.name, k, x0, y0)
    assert st.paths == len(seeds) * W * H * SPP and st.pixelsWritten == len(seeds) * W * H
    pt.close()
    return frames, st


def test_c2_cornell_1080p_8spp_windows(oracle):
    """BASELINE.json configs[1]: Cornell box 1920x1080, 8 spp, depth 4 -- FULL size (auto schedule = megakernel for this scene)."""
    s = scenes.cornell_box()
    wins = [(0, 0, 32, 16), (944, 532, 976, 548), (1888, 1064, 1920, 1080), (400, 100, 432, 116), (1300, 900, 1332, 916)]
    _windows_vs_oracle(oracle, s, 1920, 1080, 8, wins, [0x12345678, (0x12345678 + 0x9E3779B9) & 0xFFFFFFFF])


def test_c3_bunny_1080p_8spp_windows(oracle):
    """BASELINE.json configs[2]: bunny-class (69,620-triangle displaced icosphere in the Cornell box) 1920x1080, 8 spp -- FULL
    size, the scene bench.py times under other_workloads."""
    s = scenes.make_scene("bunny")
    assert 69_000 < s.tri_count < 70_500
    wins = [(0, 0, 24, 12), (948, 534, 972, 546), (1896, 1068, 1920, 1080), (800, 400, 824, 412), (1100, 640, 1124, 652), (960, 300, 984, 312)]
    _windows_vs_oracle(oracle, s, 1920, 1080, 8, wins, [0x12345678, (0x12345678 + 0x9E3779B9) & 0xFFFFFFFF])


def test_c4_benched_sponza_tex1024_windows_and_bench_crc(oracle):
    """The scene bench.py BENCHES (BASELINE.json configs[3]): Sponza-class with its 24 x 1024^2 textures (the 100 MB atlas, its
    offsets, the (float)width - 1 arithmetic at 1024), 1920x1080, 8 spp, depth 4, bench.py's seeds and pass schedule.  Windows
    equal the oracle bit for bit, and the frame's CRC-32 is computed the way bench.py computes config.frame_crc32, from the
    same scene constructor and seed schedule, so the benched frame IS a checked frame."""
    import zlib
    import bench
    wl = bench.WORKLOADS["sponza"]
    assert wl["kw"]["tex_size"] == 1024
    s = scenes.make_scene(wl["scene"], **wl["kw"])
    assert s.texture_data.nbytes > 100e6
    seeds = [bench.seed_of(k) for k in range(3)]
    wins = [(0, 0, 24, 12), (948, 534, 972, 546), (1896, 1068, 1920, 1080), (300, 900, 324, 912), (1500, 200, 1524, 212), (700, 40, 724, 52)]
    frames, _ = _windows_vs_oracle(oracle, s, 1920, 1080, 8, wins, seeds)
    # bench.py --steps 2 --warmup 1 ends on pass index 2 and prints the CRC-32 of that frame
    crc = zlib.crc32(frames[2].tobytes())
    print(f"[parity] benched sponza frame after passes 0..2: crc32 = {crc}")
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--no-extra", "--no-cpu-baseline", "--no-alg-replay"],
                         capture_output=True, text=True, cwd=root, timeout=600)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert line, out.stderr[-2000:]
    assert json.loads(line[-1])["config"]["frame_crc32"] == crc


def test_api_error_behaviour():
    """Every PT* entry point returns 0 or a negative PT_ERR_* code with a message; nothing is rendered on bad input."""
    import ctypes as C
    from unity_webgpu_pathtracer_amd import abi, plugin
    lib = plugin.load_library()
    ctx = C.c_void_p()
    assert lib.PTCreate(99, C.byref(ctx)) in (abi.PT_ERR_NO_DEVICE, abi.PT_ERR_INVALID_ARG, abi.PT_ERR_HIP) and not ctx.value   # no such device
    assert lib.PTCreate(0, C.byref(ctx)) == abi.PT_OK
    s = scenes.cornell_box()
    p = scenes.frame_params(s, 64, 64)
    assert lib.PTRenderPass(ctx, C.byref(p)) == abi.PT_ERR_NO_SCENE and b"PTSetScene" in lib.PTGetLastError()
    assert lib.PTRenderPass(ctx, None) == abi.PT_ERR_INVALID_ARG
    assert lib.PTSetScene(ctx, None) == abi.PT_ERR_INVALID_ARG
    bad = abi.PTSceneDesc()                                        # all NULL
    assert lib.PTSetScene(ctx, C.byref(bad)) == abi.PT_ERR_INVALID_ARG
    lib.PTDestroy(ctx)

    pt = PathTracer(s, width=64, height=64)
    d = pt._bvhScene.desc()
    d.features |= abi.PT_FEATURE_HAS_TEXTURES                      # keyword on, no data
    assert lib.PTSetScene(pt.ctx, C.byref(d)) == abi.PT_ERR_INVALID_ARG
    d = pt._bvhScene.desc()
    d.bvhNodesBytes = 81                                           # not a multiple of the 80-byte node
    assert lib.PTSetScene(pt.ctx, C.byref(d)) == abi.PT_ERR_INVALID_ARG
    d = pt._bvhScene.desc()
    d.bvhNodesBytes = 80 * ((1 << 32) // 80 + 1)                   # 4 GiB of nodes and more: refused before a byte is read (32-bit node offsets)
    assert lib.PTSetScene(pt.ctx, C.byref(d)) == abi.PT_ERR_INVALID_ARG and b"4 GiB" in lib.PTGetLastError()
    d = pt._bvhScene.desc()
    d.features |= abi.PT_FEATURE_HAS_ENVIRONMENT_TEXTURE
    assert lib.PTSetScene(pt.ctx, C.byref(d)) == abi.PT_ERR_INVALID_ARG
    zero = scenes.frame_params(s, 64, 64)
    zero.OutputWidth = 0
    assert lib.PTRenderPass(pt.ctx, C.byref(zero)) == abi.PT_ERR_INVALID_ARG
    huge = scenes.frame_params(s, 64, 64)
    huge.OutputWidth, huge.OutputHeight = 65536, 65536
    assert lib.PTRenderPass(pt.ctx, C.byref(huge)) == abi.PT_ERR_INVALID_ARG
    later = scenes.frame_params(s, 64, 64, current_sample=8)
    assert lib.PTRenderPassTo(pt.ctx, C.byref(later), C.c_void_p(pt.lib.PTGetFramePointer(pt.ctx, 0)), None) == abi.PT_ERR_INVALID_ARG
    assert lib.PTRenderPassTo(pt.ctx, C.byref(p), None, None) == abi.PT_ERR_INVALID_ARG
    assert lib.PTSetSchedule(pt.ctx, 9) == abi.PT_ERR_UNSUPPORTED and lib.PTSetSchedule(pt.ctx, -1) == abi.PT_OK
    assert lib.PTSetTileOwnership(pt.ctx, 3, 2) == abi.PT_ERR_INVALID_ARG
    # round 3: scheduling knobs and pass batches
    assert lib.PTSetPassesInFlight(pt.ctx, 99) == abi.PT_ERR_INVALID_ARG and lib.PTSetPassesInFlight(pt.ctx, -1) == abi.PT_ERR_INVALID_ARG
    assert lib.PTSetPassesInFlight(pt.ctx, 2) == abi.PT_OK and lib.PTGetPassesInFlight(pt.ctx) == 2
    assert lib.PTSetPassesInFlight(pt.ctx, 0) == abi.PT_OK and lib.PTGetPassesInFlight(pt.ctx) in (3, 6, 12)     # sized to GPU_MAX_HW_QUEUES
    assert lib.PTSetSubFrames(pt.ctx, 0) == abi.PT_ERR_INVALID_ARG and lib.PTSetSubFrames(pt.ctx, 99) == abi.PT_ERR_INVALID_ARG
    assert lib.PTSetSubFrames(pt.ctx, 1) == abi.PT_OK
    assert lib.PTRenderPass(pt.ctx, C.byref(p)) == abi.PT_OK              # allocates the internal frame pair
    frame = C.c_void_p(pt.lib.PTGetFramePointer(pt.ctx, 0))
    two = (abi.PTFrameParams * 2)(p, scenes.frame_params(s, 64, 64, current_sample=1, seed=7))
    assert lib.PTRenderPassBatchTo(pt.ctx, two, 0, frame, None) == abi.PT_ERR_INVALID_ARG
    assert lib.PTRenderPassBatchTo(pt.ctx, two, 9, frame, None) == abi.PT_ERR_INVALID_ARG
    assert lib.PTRenderPassBatchTo(pt.ctx, None, 2, frame, None) == abi.PT_ERR_INVALID_ARG
    assert lib.PTRenderPassBatchTo(pt.ctx, two, 2, None, None) == abi.PT_ERR_INVALID_ARG
    differ = (abi.PTFrameParams * 2)(p, scenes.frame_params(s, 64, 64, current_sample=1, seed=7, max_bounces=2))
    assert lib.PTRenderPassBatchTo(pt.ctx, differ, 2, frame, None) == abi.PT_ERR_INVALID_ARG and b"RngSeedRoot" in lib.PTGetLastError()
    assert lib.PTRenderPassBatchTo(pt.ctx, two, 2, frame, None) == abi.PT_OK and lib.PTSynchronize(pt.ctx) == abi.PT_OK
    assert lib.PTRenderPassBatch(pt.ctx, two, 2) == abi.PT_OK and lib.PTSynchronize(pt.ctx) == abi.PT_OK
    # "Versioning": a host that does not set structSize is refused, one built against an older (shorter) header is served
    unsized = pt._bvhScene.desc()
    unsized.structSize = 0
    assert lib.PTSetScene(pt.ctx, C.byref(unsized)) == abi.PT_ERR_INVALID_ARG and b"structSize" in lib.PTGetLastError()
    older = pt._bvhScene.desc()
    older.structSize = abi.PTSceneDesc.envTexture.offset            # the layout before the environment-texture members
    older.envTexture, older.envWidth = 0xDEAD0000, 7                # must be ignored: beyond the host's structSize
    assert lib.PTSetScene(pt.ctx, C.byref(older)) == abi.PT_OK
    nop = scenes.frame_params(s, 64, 64)
    nop.structSize = 0
    assert lib.PTRenderPass(pt.ctx, C.byref(nop)) == abi.PT_ERR_INVALID_ARG and b"structSize" in lib.PTGetLastError()
    many = scenes.frame_params(s, 64, 64, spp=5000)
    assert lib.PTSetSchedule(pt.ctx, 1) == abi.PT_OK
    assert lib.PTRenderPass(pt.ctx, C.byref(many)) == abi.PT_ERR_UNSUPPORTED          # the wavefront packs spp into 12 bits
    buf = np.zeros(16, np.float32)
    assert lib.PTReadback(pt.ctx, buf.ctypes.data_as(C.c_void_p), buf.size) == abi.PT_ERR_INVALID_ARG   # no frame yet / too small
    # a good pass still works after all of that
    assert lib.PTSetSchedule(pt.ctx, -1) == abi.PT_OK
    pt.render_pass(pt.params(seed=5))
    assert np.isfinite(pt.readback()).all()
    pt.close()


def test_scene_with_bad_indices_is_refused():
    """PTSetScene follows every index the kernels will follow (D3D / WebGPU return zeros for an out-of-range read, a HIP kernel
    faults the GPU): a bad material index, texture descriptor, CWBVH child / triangle range, primitive index, TLAS index or
    instance offset is PT_ERR_INVALID_ARG, and the context keeps rendering the previous scene."""
    import ctypes as C
    from unity_webgpu_pathtracer_amd import abi, plugin
    lib = plugin.load_library()
    s = scenes.material_zoo()
    pt = PathTracer(s, width=48, height=32, samplesPerPass=1)
    pt.render_pass(pt.params(seed=1))
    good = pt.readback().copy()
    bs = pt._bvhScene

    def refused(desc, what):
        rc = lib.PTSetScene(pt.ctx, C.byref(desc))
        assert rc == abi.PT_ERR_INVALID_ARG, (what, rc)
        assert what.encode() in lib.PTGetLastError(), (what, lib.PTGetLastError())

    attrs = bs.scene.tri_attrs.copy()
    attrs["materialIndex"][5] = 10_000
    d = bs.desc(); d.triAttrs = attrs.ctypes.data
    refused(d, "materialIndex")
    mats = np.ascontiguousarray(bs.scene.materials, np.float32).copy()
    mats[0, 22] = 4000.0                                           # base-colour texture slot -> no such descriptor
    d = bs.desc(); d.materials = mats.ctypes.data
    refused(d, "texture index")
    tex = bs.scene.texture_data.copy()
    tex[2] = tex.size - 3                                          # descriptor 0: offset + w*h runs past the end
    d = bs.desc(); d.textureData = tex.ctypes.data
    refused(d, "descriptor")
    nodes = np.frombuffer(bs.bvh_nodes.tobytes(), abi.CWBVH_NODE).copy()
    k = int(np.nonzero(nodes["imask"])[0][0])
    nodes["childBaseIndex"][k] = len(nodes)                        # children past the node array
    d = bs.desc(); d.bvhNodes = nodes.ctypes.data
    refused(d, "child index")
    nodes = np.frombuffer(bs.bvh_nodes.tobytes(), abi.CWBVH_NODE).copy()
    leafy = int(np.nonzero((nodes["meta"] != 0).any(axis=1) & (nodes["imask"] == 0))[0][0])
    nodes["triBaseIndex"][leafy] = 0x7FFFFFF0
    d = bs.desc(); d.bvhNodes = nodes.ctypes.data
    refused(d, "triangle rows")
    tris = np.frombuffer(bs.bvh_tris.tobytes(), np.uint32).copy().reshape(-1, 4)
    tris[2, 3] = 0x00FFFFFF                                        # primitive index of triangle 0
    d = bs.desc(); d.bvhTris = tris.ctypes.data
    refused(d, "primitive index")
    # round 3 (ADVICE): a child the KERNEL treats as inner (meta bits 3 and 4 set) but whose upper bits are not 001, hidden behind
    # imask == 0: validation must decode the byte as cwbvh_node_hitmask does, not as the builder writes it
    nodes = np.frombuffer(bs.bvh_nodes.tobytes(), abi.CWBVH_NODE).copy()
    nodes["meta"][leafy][0] = 0x58
    d = bs.desc(); d.bvhNodes = nodes.ctypes.data
    refused(d, "inner child whose meta byte")
    nodes = np.frombuffer(bs.bvh_nodes.tobytes(), abi.CWBVH_NODE).copy()
    nodes["childBaseIndex"][k] = 0                                 # a child pointer that points back at the root: traversal would never end
    d = bs.desc(); d.bvhNodes = nodes.ctypes.data
    refused(d, "reachable twice")
    nodes = np.frombuffer(bs.bvh_nodes.tobytes(), abi.CWBVH_NODE).copy()
    nodes["meta"][leafy][:] = 0
    nodes["meta"][leafy][0] = (7 << 5) | 23                        # three triangle bits starting at bit 23: they leave the 24-bit mask
    d = bs.desc(); d.bvhNodes = nodes.ctypes.data
    refused(d, "leave the 24-bit mask")
    pt.render_pass(pt.params(seed=1))                              # the old scene is still there and still renders the same bits
    assert np.array_equal(pt.readback().view(np.uint32), good.view(np.uint32))
    pt.close()
    # HAS_TLAS: instance offsets and TLAS indices
    si = scenes.instanced_scene(count=5, detail=6)
    pt = PathTracer(si, width=32, height=32, samplesPerPass=1)
    bs = pt._bvhScene
    inst = bs.gpu_instances.copy()
    inst["bvhOffset"][1] = 1 << 28
    d = bs.desc(); d.gpuInstances = inst.ctypes.data
    refused(d, "instance 1")
    tl = bs.tlas_data.copy()
    tl.view(np.uint32)[bs.tlas_index_offset] = 99
    d = bs.desc(); d.tlasData = tl.ctypes.data
    refused(d, "TLAS index")
    tl = bs.tlas_data.copy()
    inner = [n for n in range(bs.tlas_index_offset // 16) if tl.view(np.uint32)[n * 16 + 11] == 0]
    tl.view(np.uint32)[inner[-1] * 16 + 3] = 0                     # left child of an inner TLAS node -> the root: a cycle
    d = bs.desc(); d.tlasData = tl.ctypes.data
    refused(d, "TLAS node 0 is reachable twice")
    d = bs.desc(); d.tlasIndexOffset = 0
    rc = lib.PTSetScene(pt.ctx, C.byref(d))
    assert rc == abi.PT_ERR_INVALID_ARG
    pt.close()


def test_c5_4k_64spp_one_rank_of_eight(oracle):
    """BASELINE.json configs[4] as ONE rank sees it: Sponza-class, 3840x2160, 64 spp per pass, rank 3 of 8.  Owned windows equal
    the oracle's bits, non-owned pixels are exact zeros, and the rank renders its 1/8 of the paths."""
    s = scenes.sponza_atrium(tex_size=64)
    W, H, SPP, rank, world = 3840, 2160, 64, 3, 8
    pt = PathTracer(s, width=W, height=H, samplesPerPass=SPP, rank=rank, world_size=world)
    p = pt.params(seed=0xC5C5C5)
    pt.render_pass(p)
    f = pt.readback()
    st = pt.stats()
    from unity_webgpu_pathtracer_amd import distributed
    mask = distributed.ownership_mask(W, H, rank, world)
    assert st.paths == int(mask.sum()) * SPP and abs(mask.mean() - 1.0 / world) < 0.01
    assert (f[~mask] == 0).all() and (f[mask][:, 3] == 1.0).all() and np.isfinite(f).all()
    b = oracle.buffers_from_bvhscene(pt._bvhScene)
    checked = 0
    for (bx, by) in ((0, 0), (100, 60), (239, 134), (17, 90)):                 # 16x16 blocks; pick the ones this rank owns nearby
        while (bx + by) % world != rank:
            bx += 1
        x0, y0 = bx * 16, by * 16
        x1, y1 = min(x0 + 16, W), min(y0 + 8, H)
        r, _ = oracle.render(b, p, window=(x0, y0, x1, y1), tile_rank=rank, tile_world=world)
        assert mask[y0:y1, x0:x1].all()
        assert np.array_equal(f[y0:y1, x0:x1].view(np.uint32), r[y0:y1, x0:x1].view(np.uint32)), (x0, y0)
        checked += 1
    assert checked == 4
    pt.close()


@pytest.mark.parametrize("schedule", [0, 1])
def test_all_keywords_together(oracle, schedule):
    """HAS_TLAS + HAS_ENVIRONMENT_TEXTURE + HAS_LIGHTS (+ thin lens, firefly filter, map rotation): every shader keyword of
    PathTracer.compute:6-9 on at once."""
    s = scenes.instanced_scene(count=9, detail=8)
    s.environment_texture = scenes.sky_environment(48, 24, sun_dir=(-0.3, 0.6, 0.4))
    s.environment_map_rotation = 0.35
    s.environment_intensity = 0.8
    s.camera.aperture, s.camera.focal_length = 0.05, 8.0
    assert s.features & 0xD == 0xD
    pt = PathTracer(s, width=120, height=72, samplesPerPass=3, fireflyFilter=True, maxFireflyLuminance=6.0, schedule=schedule)
    pt.set_stats_level(1)
    p = pt.params(seed=0xA11)
    pt.render_pass(p)
    gpu = pt.readback()
    ref, st = _oracle_frame(oracle, pt, p, shadow_any_hit=True)
    _, nbad = _compare(gpu, ref, f"all keywords, schedule {schedule}")
    assert nbad == 0
    _stats_equal(pt.stats(), st, ALL_COUNTERS)
    pt.close()


def _analytic_scenes():
    """The constructed scenes of tests/test_oracle_analytic.py (closed-form images on the oracle): thin lens, a rectangle light seen
    directly, spot and point lights with a short range, an emission texture -- through the kernels, bit for bit."""
    import test_oracle_analytic as A
    cam = scenes.Camera(eye=(0, 0, 0), target=(0, 0, 1), vfov_deg=30.0, aperture=0.3, focal_length=6.0)
    yield "thin lens", A._light_scene((0.0, 0.0, 4.0), (1, 0, 0), (0, -1, 0), (0.4, 0.4), cam, emission=(1.0, 1.0, 1.0)), 64, 64, 8
    yield "spot light", A._floor_scene(scenes.pack_spot_light((0.3, 2.0, -0.2), (0.2, -1.0, 0.1), 70.0, 40.0, (4.0, 4.0, 4.0), rng=50.0)), 64, 64, 4
    yield "point light, short range", A._floor_scene(scenes.pack_point_light((0.0, 1.5, 0.0), (4.0, 4.0, 4.0), rng=3.0)), 64, 64, 4


@pytest.mark.parametrize("schedule", [0, 1, 4])
def test_analytic_scenes_bit_exact(oracle, schedule):
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    for what, s, w, h, spp in _analytic_scenes():
        pt = PathTracer(s, width=w, height=h, samplesPerPass=spp, schedule=schedule, maxRayBounces=2)
        pt.set_stats_level(1)
        p = pt.params(seed=0xA11CE)
        pt.render_pass(p)
        gpu = pt.readback()
        ref, st = _oracle_frame(oracle, pt, p, shadow_any_hit=True)
        _, nbad = _compare(gpu, ref, what)
        assert nbad == 0
        _stats_equal(pt.stats(), st, ALL_COUNTERS)
        pt.close()


def _fuzz_scene(seed):
    """A random small scene: every material parameter drawn over its whole range (extremes included), random lights of all three
    types, random textures in random slots, random camera with or without a lens, one of the three sky settings."""
    rng = np.random.RandomState(seed)
    pick = lambda lo, hi: float(rng.choice([lo, hi, rng.uniform(lo, hi), rng.uniform(lo, hi)]))
    n_tex = int(rng.randint(0, 4))
    images = [rng.uniform(0, 1, size=(int(rng.choice([1, 2, 5, 16])), int(rng.choice([1, 3, 8, 16])), 4)).astype(np.float32) for _ in range(n_tex)]
    tex = scenes.pack_textures(images)
    tslot = lambda: int(rng.randint(0, n_tex)) if n_tex and rng.rand() < 0.5 else -1
    mats = []
    for _ in range(int(rng.randint(2, 7))):
        mats.append(scenes.pack_material(
            color=(pick(0, 1), pick(0, 1), pick(0, 1), pick(0, 1)), transmission=pick(0, 1) if rng.rand() < 0.3 else 0.0,
            emission=tuple(rng.uniform(0, 3, 3)) if rng.rand() < 0.2 else (0, 0, 0), metallic=pick(0, 1), roughness=pick(0, 1),
            ior=pick(0.5, 2.5), alpha_mode=int(rng.randint(0, 3)), alpha_cutoff=pick(0, 1), anisotropic=pick(-1, 1), specular_tint=pick(0, 1),
            sheen=pick(0, 1), sheen_tint=pick(0, 1), subsurface=pick(0, 1), clearcoat=pick(0, 1), clearcoat_gloss=pick(0, 1),
            tex_base=tslot(), tex_mr=tslot(), tex_emission=tslot(), tex_occlusion=tslot(),
            uv_scale=(pick(0.5, 4), pick(0.5, 4)), uv_offset=(pick(-2, 2), pick(-2, 2))))
    mats = np.stack(mats)
    sb = scenes.SoupBuilder()
    sb.quad((-4, 0, -4), (8, 0, 0), (0, 0, 8), (0, 1, 0), 3, 3, 0)
    sb.quad((-4, 0, 4), (8, 0, 0), (0, 5, 0), (0, 0, -1), 2, 2, int(rng.randint(0, len(mats))))
    for _ in range(int(rng.randint(2, 6))):
        c = (rng.uniform(-2.5, 2.5), rng.uniform(0.3, 1.5), rng.uniform(-1.5, 2.5))
        if rng.rand() < 0.6:
            sb.uv_sphere(c, rng.uniform(0.2, 0.7), 10, 6, int(rng.randint(0, len(mats))))
        else:
            sb.quad(c, (rng.uniform(0.5, 2), 0, rng.uniform(-0.5, 0.5)), (0, rng.uniform(0.5, 2), rng.uniform(-0.5, 0.5)), (0, 0.2, -1), 2, 2, int(rng.randint(0, len(mats))))
    verts, attrs = sb.finish()
    lights = []
    for _ in range(int(rng.randint(0, 4))):
        k = int(rng.randint(0, 3))
        p = (rng.uniform(-3, 3), rng.uniform(2, 4.5), rng.uniform(-3, 3))
        col = tuple(rng.uniform(1, 20, 3))
        if k == 0:
            lights.append(scenes.pack_rect_light(p, (1, 0, 0), (0, 0, 1), (rng.uniform(0.3, 2), rng.uniform(0.3, 2)), col, rng=pick(3, 50)))
        elif k == 1:
            lights.append(scenes.pack_point_light(p, col, rng=pick(3, 20)))
        else:
            lights.append(scenes.pack_spot_light(p, (rng.uniform(-1, 1), -1.0, rng.uniform(-1, 1)), rng.uniform(30, 120), rng.uniform(5, 30), col, rng=pick(3, 20)))
    lights = np.stack(lights) if lights else np.zeros((0, 16), np.float32)
    lens = rng.rand() < 0.4
    cam = scenes.Camera(eye=(rng.uniform(-1, 1), rng.uniform(1.5, 3), -5.5), target=(rng.uniform(-0.5, 0.5), 0.7, 0.5), vfov_deg=rng.uniform(25, 70),
                        aperture=rng.uniform(0.02, 0.3) if lens else 0.0, focal_length=rng.uniform(3, 8) if lens else 0.0)
    mode = int(rng.randint(0, 2))
    return scenes.Scene(f"fuzz{seed}", verts, attrs, mats, lights, tex, cam, environment_mode=mode,
                        environment_color=tuple(rng.uniform(0, 1.5, 3)) + (1.0,), environment_intensity=pick(0, 2))


@pytest.mark.parametrize("schedule", [0, 1, 4])
def test_fuzzed_scenes_bit_exact(oracle, schedule):
    """Thirty-two random scenes (materials over their whole parameter ranges, all light types, textures in every slot, lens on / off, both sky
    modes), 48 x 32 at 3 spp with Russian roulette and the firefly filter on every other one: frames and all counters equal the oracle's."""
    for seed in range(32):
        s = _fuzz_scene(1000 + seed)
        pt = PathTracer(s, width=48, height=32, samplesPerPass=3, schedule=schedule, maxRayBounces=1 + seed % 5)
        pt.set_stats_level(1)
        p = scenes.frame_params(s, 48, 32, spp=3, seed=0xF00D + seed, max_bounces=1 + seed % 5, russian_roulette=seed % 2 == 0,
                                firefly=seed % 2 == 1, max_firefly_luminance=4.0)
        pt.render_pass(p)
        gpu = pt.readback()
        ref, st = _oracle_frame(oracle, pt, p, shadow_any_hit=True)
        nan_g, nan_r = np.isnan(gpu), np.isnan(ref)
        assert np.array_equal(nan_g, nan_r), f"fuzz {seed}: NaN pattern differs"
        assert np.array_equal(gpu.view(np.uint32), ref.view(np.uint32)), f"fuzz {seed}: {int((gpu.view(np.uint32) != ref.view(np.uint32)).any(axis=-1).sum())} pixels differ"
        _stats_equal(pt.stats(), st, ALL_COUNTERS)
        pt.close()


@pytest.mark.parametrize("schedule", [0, 1])
def test_fuzzed_instanced_scenes_bit_exact(oracle, schedule):
    """HAS_TLAS: ten instanced scenes with random instance counts (1 ... 60), placements, rotations and non-uniform scales."""
    for k, (count, seed) in enumerate(((1, 3), (2, 4), (3, 5), (7, 6), (12, 7), (20, 8), (33, 9), (47, 10), (60, 11), (25, 12))):
        s = scenes.instanced_scene(count=count, detail=6 + k % 3 * 3, seed=seed)
        pt = PathTracer(s, width=64, height=40, samplesPerPass=2, schedule=schedule)
        pt.set_stats_level(1)
        p = pt.params(seed=0x7145 + k)
        pt.render_pass(p)
        gpu = pt.readback()
        ref, st = _oracle_frame(oracle, pt, p, shadow_any_hit=True)
        assert np.array_equal(gpu.view(np.uint32), ref.view(np.uint32)), f"instanced {count}/{seed}: {int((gpu.view(np.uint32) != ref.view(np.uint32)).any(axis=-1).sum())} pixels differ"
        _stats_equal(pt.stats(), st, ALL_COUNTERS)
        pt.close()
