// pt_tiles.hip -- frame assembly for screen-tile sharding (no reference counterpart: the reference renders on one GPU).
//
// A rank owns the 16x16-pixel blocks (bx + by) % world == rank (PTSetTileOwnership).  Its frame holds its own tiles and exact
// zeros elsewhere, so the frame CAN be assembled with a sum-reduce of whole frames; moving only what a rank owns is 1/world of
// those bytes: pack the owned tiles into a dense buffer in slot order (pt_launch.h: pt_slot_to_pixel, the order the kernels
// enumerate pixels in), move that buffer (hipMemcpyPeerAsync inside one process, an RCCL gather between processes), and
// scatter it into the assembled frame on the root.  Both kernels are pure 16-byte copies: one read + one write per owned
// pixel, coalesced in 8-pixel (128-byte) runs on the frame side and fully on the packed side.
#include "pt_launch.h"

namespace {

__global__ __launch_bounds__(256) void pt_pack_tiles(PTTileMap tm, uint32_t frameWidth, uint32_t numSlots, const float4* __restrict__ frame,
                                                     float4* __restrict__ packed)
{
    const uint32_t slot = blockIdx.x * 256u + threadIdx.x;
    if (slot >= numSlots) return;
    uint32_t px, py;
    packed[slot] = pt_slot_to_pixel(tm, slot, px, py) ? frame[(size_t)py * frameWidth + px] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

__global__ __launch_bounds__(256) void pt_unpack_tiles(PTTileMap tm, uint32_t frameWidth, uint32_t numSlots, const float4* __restrict__ packed,
                                                       float4* __restrict__ frame)
{
    const uint32_t slot = blockIdx.x * 256u + threadIdx.x;
    if (slot >= numSlots) return;
    uint32_t px, py;
    if (pt_slot_to_pixel(tm, slot, px, py)) frame[(size_t)py * frameWidth + px] = packed[slot];
}

} // namespace

hipError_t pt_launch_pack_tiles(const PTTileMap& tm, uint32_t frameWidth, const float4* frame, float4* packed, hipStream_t stream)
{
    const uint32_t n = pt_num_slots(tm);
    hipLaunchKernelGGL(pt_pack_tiles, dim3(n / 256u), dim3(256), 0, stream, tm, frameWidth, n, frame, packed);
    return hipGetLastError();
}

hipError_t pt_launch_unpack_tiles(const PTTileMap& tm, uint32_t frameWidth, const float4* packed, float4* frame, hipStream_t stream)
{
    const uint32_t n = pt_num_slots(tm);
    hipLaunchKernelGGL(pt_unpack_tiles, dim3(n / 256u), dim3(256), 0, stream, tm, frameWidth, n, packed, frame);
    return hipGetLastError();
}
