"""The N > 1 path on CPU: two processes over gloo shard a frame as interleaved 16x16 blocks, each produces its partial
frame (here with the oracle standing in for the GPU kernels, which cannot run in this container), packs its owned tiles,
and ONE gather onto rank 0 + unpack reproduces the single-process frame bit for bit -- the same exchange bench.py issues
over RCCL (unity_webgpu_pathtracer_amd/distributed.py gather_frame; pack / unpack here through the numpy mirror of the kernels)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_path, w=80, h=48):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pyoracle
    from unity_webgpu_pathtracer_amd import distributed, plugin, scenes
    s = scenes.cornell_box()
    nodes, tris = plugin.build_cwbvh(s.vertices)
    b = pyoracle.SceneBuffers(s, nodes, tris)
    frames = [None, None]
    cur = 0
    for k in range(2):                                   # two progressive passes, each rank keeps only its own history
        p = scenes.frame_params(s, w, h, spp=2, current_sample=2 * k, seed=(0x12345678 + k * 0x9E3779B9) & 0xFFFFFFFF)
        part, _ = pyoracle.render(b, p, accumulated=frames[1 - cur], tile_rank=rank, tile_world=world, threads=1)
        frames[cur] = part
        t = distributed.gather_frame(torch.from_numpy(part.copy()), rank, world, dst=0)
        cur = 1 - cur
    if rank == 0:
        np.save(out_path, t.numpy())
    counts = torch.tensor(distributed.owned_pixel_counts(w, h, world))
    assert int(counts.sum()) == w * h
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_tile_sharding_is_bit_exact(tmp_path, oracle):
    world = 2
    out = str(tmp_path / "reduced.npy")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    reduced = np.load(out)
    from unity_webgpu_pathtracer_amd import plugin, scenes
    s = scenes.cornell_box()
    nodes, tris = plugin.build_cwbvh(s.vertices)
    b = oracle.SceneBuffers(s, nodes, tris)
    prev = None
    for k in range(2):
        p = scenes.frame_params(s, 80, 48, spp=2, current_sample=2 * k, seed=(0x12345678 + k * 0x9E3779B9) & 0xFFFFFFFF)
        prev, _ = oracle.render(b, p, accumulated=prev)
    assert np.array_equal(reduced.view(np.uint32), prev.view(np.uint32))


@pytest.mark.timeout(600)
def test_eight_rank_tile_sharding_is_bit_exact(tmp_path, oracle):
    """The world size of BASELINE.json configs[4]: eight processes over gloo, a frame whose width is not a multiple of
    8 x 16 pixels (ranks own different numbers of blocks per row, the last block column is partially covered), two progressive
    passes; the gathered frame equals the single-process one bit for bit."""
    world, w, h = 8, 200, 72
    out = str(tmp_path / "reduced8.npy")
    mp.spawn(_worker, args=(world, _free_port(), out, w, h), nprocs=world, join=True)
    reduced = np.load(out)
    from unity_webgpu_pathtracer_amd import plugin, scenes
    s = scenes.cornell_box()
    nodes, tris = plugin.build_cwbvh(s.vertices)
    b = oracle.SceneBuffers(s, nodes, tris)
    prev = None
    for k in range(2):
        p = scenes.frame_params(s, w, h, spp=2, current_sample=2 * k, seed=(0x12345678 + k * 0x9E3779B9) & 0xFFFFFFFF)
        prev, _ = oracle.render(b, p, accumulated=prev)
    assert np.array_equal(reduced.view(np.uint32), prev.view(np.uint32))


def test_ownership_is_a_partition():
    from unity_webgpu_pathtracer_amd import distributed
    for world in (1, 2, 3, 4, 8):
        total = sum(distributed.ownership_mask(1920, 1080, r, world).astype(np.int64) for r in range(world))
        assert (total == 1).all()
        counts = distributed.owned_pixel_counts(1920, 1080, world)
        assert max(counts) - min(counts) <= 0.02 * 1920 * 1080 / world     # interleaving balances within 2 %


def test_pack_unpack_mirror_roundtrip():
    """The numpy mirror of PTPackOwnedTiles / PTUnpackTiles: every pixel is in exactly one rank's packed buffer, packed sizes
    are 1/world of the frame (+ block padding), and unpacking all ranks reproduces the frame -- also for sizes that are not
    multiples of 16 and for more ranks than blocks in a row."""
    from unity_webgpu_pathtracer_amd import distributed
    rng = np.random.RandomState(1)
    for (w, h, world) in ((80, 48, 2), (1920, 1080, 8), (333, 77, 3), (40, 40, 8), (16, 16, 4)):
        frame = rng.rand(h, w, 4).astype(np.float32)
        out = np.zeros_like(frame)
        seen = np.zeros((h, w), np.int32)
        for r in range(world):
            px, py, valid = distributed.slot_pixels(w, h, r, world)
            assert px.size % 256 == 0 and px.size <= distributed.max_packed_slots(w, h, world)
            m = distributed.ownership_mask(w, h, r, world)
            assert int(valid.sum()) == int(m.sum()) and m[py[valid], px[valid]].all()
            np.add.at(seen, (py[valid], px[valid]), 1)
            distributed.unpack_tiles_numpy(distributed.pack_tiles_numpy(frame * m[..., None], r, world), out, r, world)
        assert (seen == 1).all() and np.array_equal(out, frame)
