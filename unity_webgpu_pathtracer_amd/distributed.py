"""Screen-tile sharding of a frame over the GPUs of one node -- host-side helpers.

No reference counterpart: the reference is single-GPU.  The path shards naturally because a pixel's value
depends only on (pixelIndex, CurrentSample, RngSeedRoot, scene, params) (PathTracer.compute:60): the frame is cut
into 16x16-pixel blocks, block (bx, by) belongs to rank (bx + by) % world_size (PTSetTileOwnership), every rank
renders only its blocks into a zero-initialised full frame, and ONE sum-reduce per frame onto rank 0 assembles
the image.  Adding exact zeros is exact in fp32, so the assembled frame is bit-identical to a single-GPU render.
One process per GPU; the collective is torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo"
in the CPU tests).
"""
import numpy as np

BLOCK = 16


def owner_of_block(bx, by, world_size):
    return (bx + by) % world_size


def ownership_mask(width, height, rank, world_size):
    """(H, W) bool: pixels rendered by `rank` -- the same rule the kernels use (csrc/pt_kernels.hip)."""
    ys, xs = np.mgrid[0:height, 0:width]
    return ((xs // BLOCK + ys // BLOCK) % world_size) == rank


def owned_pixel_counts(width, height, world_size):
    return [int(ownership_mask(width, height, r, world_size).sum()) for r in range(world_size)]


def reduce_frame(frame, dst=0, group=None):
    """Sum the per-rank partial frames onto `dst` (in place).  `frame` is a torch tensor (H, W, 4) float32 that holds
    this rank's blocks and exact zeros elsewhere.  Message size: W*H*16 bytes per rank (1080p: 33.2 MB; 4K: 132.7 MB);
    over xGMI a ring reduce of that is ~0.4 ms / ~1.5 ms, so it is issued once per frame, never per pass."""
    import torch.distributed as dist
    dist.reduce(frame, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return frame
