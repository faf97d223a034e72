"""Screen-tile sharding of a frame over the GPUs of one node -- the host side.

No reference counterpart: the reference is single-GPU.  The path shards naturally because a pixel's value depends only on
(pixelIndex, CurrentSample, RngSeedRoot, scene, params) (PathTracer.compute:60): the frame is cut into 16x16-pixel blocks,
block (bx, by) belongs to rank (bx + by) % world_size (PTSetTileOwnership), every rank renders only its blocks and keeps
only its own tiles' history for progressive accumulation, and ONE exchange per frame assembles the image on rank 0.

The exchange is a GATHER OF OWNED TILES, not a reduce of whole frames: a rank packs its tiles densely (PTPackOwnedTiles:
16 bytes per owned pixel, in the order the kernels enumerate pixels -- `slot_pixels` below is the numpy mirror of that
order), the packed buffers are gathered on the root (1/world of a whole-frame reduce: 4.1 MB instead of 33.2 MB per rank at
1080p / 8 ranks), and the root scatters every rank's buffer into the frame (PTUnpackTiles).  No float is ever added, so the
assembled frame is bit-identical to a single-GPU render by construction.

Two ways to run it:
  * one process per GPU (bench.py, torchrun): `gather_frame` = pack -> torch.distributed gather (backend "nccl" = RCCL over
    xGMI; "gloo" in the CPU tests) -> unpack;
  * one process, N devices (a C or C# host): `PathTracerGroup` wraps the library's own PTCreateMulti / PTGroup* entry
    points, which move the packed tiles with hipMemcpyPeerAsync.
"""
import ctypes as C

import numpy as np

from . import abi, plugin

BLOCK = 16


def owner_of_block(bx, by, world_size):
    return (bx + by) % world_size


def ownership_mask(width, height, rank, world_size):
    """(H, W) bool: pixels rendered by `rank` -- the same rule the kernels use (csrc/pt_launch.h pt_slot_to_pixel)."""
    ys, xs = np.mgrid[0:height, 0:width]
    return ((xs // BLOCK + ys // BLOCK) % world_size) == rank


def owned_pixel_counts(width, height, world_size):
    return [int(ownership_mask(width, height, r, world_size).sum()) for r in range(world_size)]


def slot_pixels(width, height, rank, world_size):
    """numpy mirror of csrc/pt_launch.h pt_num_slots / pt_slot_to_pixel for a dispatch that covers the whole frame:
    returns (px, py, valid), each of length numSlots.  Slot = (owned block in row-major (block row, k-th owned block of the
    row)) * 256 + wave * 64 + lane, one wave per 8x8 tile of the 16x16 block."""
    bx_n, by_n = (width + 15) // 16, (height + 15) // 16
    bpr = (bx_n + world_size - 1) // world_size
    slot = np.arange(bpr * by_n * 256, dtype=np.int64)
    block, tid = slot >> 8, slot & 255
    by, k = block // bpr, block % bpr
    first = (rank - by % world_size + world_size) % world_size
    bx = first + k * world_size
    wave, lane = tid >> 6, tid & 63
    px = bx * 16 + (wave & 1) * 8 + (lane & 7)
    py = by * 16 + (wave >> 1) * 8 + (lane >> 3)
    valid = (bx < bx_n) & (by < by_n) & (px < width) & (py < height)
    return px, py, valid


def pack_tiles_numpy(frame, rank, world_size):
    """What PTPackOwnedTiles computes, on a (H, W, 4) numpy frame: (numSlots, 4) with zeros in padding slots."""
    h, w = frame.shape[:2]
    px, py, valid = slot_pixels(w, h, rank, world_size)
    out = np.zeros((px.size, 4), dtype=frame.dtype)
    out[valid] = frame[py[valid], px[valid]]
    return out


def unpack_tiles_numpy(packed, frame, rank, world_size):
    """What PTUnpackTiles computes: scatter rank's packed tiles into `frame` (in place)."""
    h, w = frame.shape[:2]
    px, py, valid = slot_pixels(w, h, rank, world_size)
    frame[py[valid], px[valid]] = packed[:px.size][valid]
    return frame


def max_packed_slots(width, height, world_size):
    return max(slot_pixels(width, height, r, world_size)[0].size for r in range(world_size))


def gather_frame(frame, rank, world_size, dst=0, group=None, assembled=None, pt=None, params=None, scratch=None):
    """Assemble the frame on `dst` from every rank's owned tiles: pack -> gather -> unpack, once per frame.

    `frame`: this rank's (H, W, 4) float32 torch tensor (its tiles; anything elsewhere).  With `pt` (a PathTracer) and
    `params` the pack / unpack run on the GPU through the C-ABI, ordered on the context's stream, and the tensors must live on
    that device; without them (CPU tests, gloo) the numpy mirror does the same thing.  Returns `assembled` on dst.
    Message size: numSlots * 16 bytes per rank = 1/world of the frame (1080p, 8 ranks: 4.1 MB; 4K: 16.6 MB)."""
    import torch
    import torch.distributed as dist
    h, w = frame.shape[:2]
    n = max_packed_slots(w, h, world_size)
    if scratch is None:
        scratch = {}
    if "packed" not in scratch:
        scratch["packed"] = torch.zeros((n, 4), dtype=torch.float32, device=frame.device)
        scratch["parts"] = [torch.zeros((n, 4), dtype=torch.float32, device=frame.device) for _ in range(world_size)] if rank == dst else None
    packed, parts = scratch["packed"], scratch["parts"]
    lib = plugin.load_library()
    if pt is not None:
        plugin.check(lib.PTPackOwnedTiles(pt.ctx, C.byref(params), C.c_void_p(frame.data_ptr()), C.c_void_p(packed.data_ptr())))
    else:
        mine = pack_tiles_numpy(frame.numpy(), rank, world_size)
        packed[:mine.shape[0]] = torch.from_numpy(mine)
    if packed.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal on a one-GPU box: gloo gathers host tensors only (the copies are ordered on the current stream)
        host_parts = [torch.zeros((n, 4), dtype=torch.float32) for _ in range(world_size)] if rank == dst else None
        dist.gather(packed.cpu(), gather_list=host_parts, dst=dst, group=group)
        if rank == dst:
            for r in range(world_size):
                parts[r].copy_(host_parts[r])
    else:
        dist.gather(packed, gather_list=parts, dst=dst, group=group)
    if rank != dst:
        return None
    if assembled is None:
        assembled = torch.zeros_like(frame)
    for r in range(world_size):
        if pt is not None:
            plugin.check(lib.PTUnpackTiles(pt.ctx, C.byref(params), r, world_size, C.c_void_p(parts[r].data_ptr()), C.c_void_p(assembled.data_ptr())))
        else:
            unpack_tiles_numpy(parts[r].numpy(), assembled.numpy(), r, world_size)
    return assembled


class PathTracerGroup:
    """One process driving N devices through the library's PTCreateMulti / PTGroup* entry points (include/ptmi_plugin.h):
    the multi-GPU path a C or C# host uses.  Mirrors PathTracer's OnRenderImage bookkeeping (PathTracer.cs:224-272)."""

    def __init__(self, scene, devices, width, height, samplesPerPass=1, maxRayBounces=4, maxSamples=100000):
        from .pathtracer import BVHScene
        from . import scenes as _scenes
        self.lib = plugin.load_library()
        self.scene, self.width, self.height = scene, width, height
        self.samplesPerPass, self.maxRayBounces, self.maxSamples = samplesPerPass, maxRayBounces, maxSamples
        self._scenes = _scenes
        self._currentSample = 0
        arr = (C.c_int * len(devices))(*devices)
        grp = C.c_void_p()
        plugin.check(self.lib.PTCreateMulti(arr, len(devices), C.byref(grp)))
        self.group = grp
        self._bvhScene = BVHScene(scene)
        plugin.check(self.lib.PTGroupSetScene(self.group, C.byref(self._bvhScene.desc())))

    def params(self, seed):
        return self._scenes.frame_params(self.scene, self.width, self.height, spp=self.samplesPerPass,
                                         current_sample=self._currentSample, seed=seed, max_bounces=self.maxRayBounces)

    def OnRenderImage(self, seed):
        p = self.params(seed)
        plugin.check(self.lib.PTGroupRenderPass(self.group, C.byref(p)))
        self._currentSample += max(1, self.samplesPerPass)
        if self._currentSample < self.maxSamples:
            plugin.check(self.lib.PTGroupFlipFrames(self.group))
        return p

    def OnRenderBatch(self, seeds):
        """len(seeds) (1..8) consecutive passes as ONE launch sequence per device and one assembly (PTGroupRenderPassBatch)."""
        ps = []
        for k, seed in enumerate(seeds):
            ps.append(self._scenes.frame_params(self.scene, self.width, self.height, spp=self.samplesPerPass,
                                                current_sample=self._currentSample + k * max(1, self.samplesPerPass), seed=seed,
                                                max_bounces=self.maxRayBounces))
        arr = (abi.PTFrameParams * len(ps))(*ps)
        plugin.check(self.lib.PTGroupRenderPassBatch(self.group, arr, len(ps)))
        self._currentSample += len(ps) * max(1, self.samplesPerPass)
        if self._currentSample < self.maxSamples:
            plugin.check(self.lib.PTGroupFlipFrames(self.group))
        return ps

    def Reset(self):
        self._currentSample = 0
        plugin.check(self.lib.PTGroupResetFrames(self.group))

    def context(self, i):
        return C.c_void_p(self.lib.PTGroupGetContext(self.group, i))

    def readback(self):
        out = np.zeros((self.height, self.width, 4), dtype=np.float32)
        plugin.check(self.lib.PTGroupReadback(self.group, out.ctypes.data_as(C.c_void_p), out.size))
        return out

    def stats(self):
        st = abi.PTStats()
        plugin.check(self.lib.PTGroupGetStats(self.group, C.byref(st)))
        return st

    def close(self):
        if self.group:
            self.lib.PTGroupDestroy(self.group)
            self.group = None
