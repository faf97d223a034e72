"""bench.py end to end on the GPU box: the JSON contract at N=1, and the N>1 code path rehearsed with two ranks that
share cuda:0 (gloo stands in for RCCL, which refuses two ranks on one device).  Tile sharding must not change a bit of
the assembled frame: frame_crc32 is the same for N=1 and N=2."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--workload", "cornell", "--width", "320", "--height", "200", "--steps", "3", "--warmup", "1", "--no-extra", "--no-cpu-baseline"]


def _last_json(out):
    lines = [l for l in out.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.decode()[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_contract_and_two_rank_rehearsal():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    one = _last_json(subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + SMALL,
                                             env=env, timeout=600, stderr=subprocess.STDOUT))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in one
    assert one["n_gpus"] == 1 and one["steps"] == 3 and one["value"] > 0 and one["vs_baseline"] is None
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(one["roofline"])
    assert one["roofline"]["achieved"] > 0

    two = _last_json(subprocess.check_output(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
         "--master-port", "29611", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--same-device"] + SMALL,
        env=env, timeout=900, stderr=subprocess.STDOUT))
    assert two["n_gpus"] == 2 and two["value"] > 0
    assert two["config"]["rays_per_step"] == one["config"]["rays_per_step"]          # strong scaling: same frame, same rays
    assert two["config"]["frame_crc32"] == one["config"]["frame_crc32"] != 0
    # `python bench.py --gpus 2` WITHOUT an outer launcher: bench.py starts its own ranks as a child process and relays the line
    env2 = {k: v for k, v in env.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    self_launched = _last_json(subprocess.check_output(
        [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--same-device", "--master-port", "29613"] + SMALL,
        env=env2, timeout=900, stderr=subprocess.STDOUT))
    assert self_launched["n_gpus"] == 2 and self_launched["config"]["frame_crc32"] == one["config"]["frame_crc32"]
    # N = 1 carries the latency leg (host sync after every pass) next to the pipelined throughput
    assert one["latency_mode"]["ms_per_pass"] > 0 and one["latency_mode"]["passes"] >= 1


@pytest.mark.gpu
def test_bench_rccl_code_path_with_one_rank():
    """The N>1 path with the real backend (nccl = RCCL): process group on the device, per-frame reduce on the context's
    stream, barriers.  One rank is all a one-GPU box allows RCCL; the frame must equal the plain N=1 frame."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    plain = _last_json(subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + SMALL,
                                               env=env, timeout=600, stderr=subprocess.STDOUT))
    forced = _last_json(subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist"] + SMALL,
                                                env=env, timeout=600, stderr=subprocess.STDOUT))
    assert forced["config"]["frame_crc32"] == plain["config"]["frame_crc32"] != 0
    assert "RCCL" in forced["config"]["sharding"] and forced["value"] > 0


@pytest.mark.gpu
def test_group_api_two_contexts_one_device_matches_single_gpu():
    """PTCreateMulti / PTGroup*: ONE process, N devices (here the same device twice: a one-GPU box).  Scene replicated, tiles
    of rank i on context i, packed tiles moved with hipMemcpyPeerAsync, unpacked on the root.  Two progressive passes of the
    zoo scene: the assembled frame's bits (and CRC-32) equal the single-context render, and the summed counters match."""
    import zlib
    import numpy as np
    sys.path.insert(0, ROOT)
    from unity_webgpu_pathtracer_amd import scenes
    from unity_webgpu_pathtracer_amd.distributed import PathTracerGroup
    from unity_webgpu_pathtracer_amd.pathtracer import PathTracer
    s = scenes.material_zoo()
    W, H, SPP = 200, 120, 3                                        # not multiples of 16: partially covered edge blocks
    seeds = [0x12345678, (0x12345678 + 0x9E3779B9) & 0xFFFFFFFF]
    one = PathTracer(s, width=W, height=H, samplesPerPass=SPP)
    for seed in seeds:
        one.OnRenderImage(seed)
    ref = one.readback()
    ref_stats = one.stats()
    one.close()
    for devices in ([0, 0], [0, 0, 0]):
        grp = PathTracerGroup(s, devices, W, H, samplesPerPass=SPP)
        for seed in seeds:
            grp.OnRenderImage(seed)
        got = grp.readback()
        st = grp.stats()
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), devices
        assert zlib.crc32(got.tobytes()) == zlib.crc32(ref.tobytes())
        assert st.rays == ref_stats.rays and st.paths == ref_stats.paths and st.pixelsWritten == ref_stats.pixelsWritten
        grp.Reset()
        grp.OnRenderBatch(seeds)                                   # the two passes as ONE launch sequence per device, one assembly
        assert np.array_equal(grp.readback().view(np.uint32), ref.view(np.uint32)), devices
        grp.Reset()
        grp.OnRenderImage(seeds[0])                                # Reset(): the first pass again gives the first frame again
        again = grp.readback()
        grp.close()
        solo = PathTracer(s, width=W, height=H, samplesPerPass=SPP)
        solo.OnRenderImage(seeds[0])
        assert np.array_equal(again.view(np.uint32), solo.readback().view(np.uint32))
        solo.close()


@pytest.mark.gpu
def test_pack_unpack_kernels_match_the_numpy_mirror():
    import ctypes as C
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    from unity_webgpu_pathtracer_amd import distributed, plugin, scenes
    from unity_webgpu_pathtracer_amd.pathtracer import PathTracer
    s = scenes.cornell_box()
    lib = plugin.load_library()
    for (W, H, world) in ((333, 77, 3), (1920, 1080, 8)):
        frame = torch.rand((H, W, 4), dtype=torch.float32, device="cuda")
        out = torch.zeros_like(frame)
        for rank in range(world):
            pt = PathTracer(s, width=W, height=H, rank=rank, world_size=world)
            p = pt.params(seed=1)
            n = C.c_uint64()
            plugin.check(lib.PTGetOwnedTileSlots(pt.ctx, C.byref(p), C.byref(n)))
            assert n.value == distributed.slot_pixels(W, H, rank, world)[0].size
            packed = torch.full((n.value, 4), -1.0, dtype=torch.float32, device="cuda")
            torch.cuda.synchronize()
            plugin.check(lib.PTPackOwnedTiles(pt.ctx, C.byref(p), C.c_void_p(frame.data_ptr()), C.c_void_p(packed.data_ptr())))
            plugin.check(lib.PTUnpackTiles(pt.ctx, C.byref(p), rank, world, C.c_void_p(packed.data_ptr()), C.c_void_p(out.data_ptr())))
            pt.synchronize()
            assert np.array_equal(packed.cpu().numpy(), distributed.pack_tiles_numpy(frame.cpu().numpy(), rank, world))
            pt.close()
        assert torch.equal(out, frame)
