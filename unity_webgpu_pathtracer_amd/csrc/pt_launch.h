// pt_launch.h — host-visible launcher declarations shared by pt_kernels.hip and pt_api.hip.
#pragma once
#include <hip/hip_runtime.h>
#include "pt_device.h"

// Which 16x16-pixel blocks this context renders (PTSetTileOwnership) and how much of the frame the
// dispatch covers (PTFrameParams.DispatchGroupsX/Y, PathTracer.cs:203-208).
struct PTTileMap {
    uint32_t rank, world;
    uint32_t blocksX, blocksY;     // ceil(coverW/16), ceil(coverH/16)
    uint32_t coverW, coverH;       // pixels with x < coverW && y < coverH are rendered
};

// slot <-> pixel: owned 16x16 blocks in row-major order of (block row, k-th owned block of that row), 256 slots per block,
// one wave per 8x8 tile.  Shared by the megakernel, the wavefront schedules and the tile pack / unpack of the multi-GPU gather.
__host__ __device__ inline uint32_t pt_num_slots(const PTTileMap& tm)
{
    const uint32_t bpr = (tm.blocksX + tm.world - 1u) / tm.world;       // owned blocks per block-row (upper bound)
    return bpr * tm.blocksY * 256u;
}
__host__ __device__ inline bool pt_slot_to_pixel(const PTTileMap& tm, uint32_t slot, uint32_t& px, uint32_t& py)
{
    const uint32_t bpr = (tm.blocksX + tm.world - 1u) / tm.world;
    const uint32_t block = slot >> 8, tid = slot & 255u;
    const uint32_t by = block / bpr, k = block % bpr;
    const uint32_t first = (uint32_t)(((int)tm.rank - (int)(by % tm.world) + (int)tm.world) % (int)tm.world);
    const uint32_t bx = first + k * tm.world;
    const uint32_t wave = tid >> 6, lane = tid & 63u;
    px = bx * 16u + (wave & 1u) * 8u + (lane & 7u);
    py = by * 16u + (wave >> 1) * 8u + (lane >> 3);
    return bx < tm.blocksX && by < tm.blocksY && px < tm.coverW && py < tm.coverH;
}
inline PTTileMap pt_make_tile_map(const PTFrameParams& p, int rank, int world)
{
    PTTileMap tm;
    tm.rank = (uint32_t)rank;
    tm.world = (uint32_t)world;
    uint32_t covW = p.OutputWidth, covH = p.OutputHeight;
    if (p.DispatchGroupsX && p.DispatchGroupsX * 8u < covW) covW = p.DispatchGroupsX * 8u;
    if (p.DispatchGroupsY && p.DispatchGroupsY * 8u < covH) covH = p.DispatchGroupsY * 8u;
    tm.coverW = covW; tm.coverH = covH;
    tm.blocksX = (covW + 15u) / 16u;
    tm.blocksY = (covH + 15u) / 16u;
    return tm;
}

// multi-GPU frame assembly (pt_tiles.hip): packed[slot] = frame[pixel(slot)] and back
hipError_t pt_launch_pack_tiles(const PTTileMap& tm, uint32_t frameWidth, const float4* frame, float4* packed, hipStream_t stream);
hipError_t pt_launch_unpack_tiles(const PTTileMap& tm, uint32_t frameWidth, const float4* packed, float4* frame, hipStream_t stream);

// per-light constants (pt_device.h derive_light_rows): S.lights / S.lightCount must be set; lightConst holds 64 bytes per light
hipError_t pt_launch_derive_lights(const DScene& S, float4* lightConst, hipStream_t stream);

hipError_t pt_launch_megakernel(const DScene& S, const PTFrameParams& P, const float4* accumulated, float4* output,
                                const PTTileMap& tm, unsigned long long* gstats, bool fullStats, hipStream_t stream);

hipError_t pt_launch_process_mesh(const PTMeshDesc& M, const void* dVertexBuffer, const void* dIndexBuffer, float4* dVertexPositions,
                                  float4* dTriangleAttributes, hipStream_t stream);
hipError_t pt_launch_copy_texture(const float4* dTexture, uint32_t width, uint32_t height, uint32_t dataOffset, uint32_t descriptorOffset,
                                  int hasAlpha, uint32_t* dTextureData, hipStream_t stream);
hipError_t pt_launch_present(const PTPresentParams& Q, const float4* src, float4* dst, hipStream_t stream);

// ---- schedule 1 (wavefront): slot-indexed path state in HBM (see pt_wavefront.hip) ----
// float4 arrays of a state set, in carving order.  PT_F4_RAY0/1/2 are the RAY RECORDS of the three ray kinds a slot can have in
// flight (bounce ray, environment NEE, light NEE): 32 bytes per slot and kind, {origin.xyz, w0, direction.xyz, w1} at
// float4 index 2 * slot -- so each takes TWO strides.  A trace lane fetches its ray as one 32-byte read inside one 64-byte
// sector; round 2 kept origins and directions in separate 16-byte arrays and fetched two sectors per ray (12.2 GB of sectors
// per pass for 2.9 GB of rays).  w0 / w1 of the bounce record carry scatterPdf / maxRoughness.
enum : uint32_t { PT_F4_RAY0 = 0, PT_F4_RAY1 = 2, PT_F4_RAY2 = 4, PT_F4_RAD = 6, PT_F4_THR, PT_F4_COLOR, PT_F4_ENVC, PT_F4_LIGHTC,
                  PT_F4_PTHR, PT_F4_HIT, PT_F4_HIT2, PT_F4_PIXSUM, PT_F4_COUNT };
#ifndef PT_WF_LDS_STACK
#define PT_WF_LDS_STACK 8       // traversal-stack entries per lane kept in LDS by the refill / persistent trace kernels; deeper ones go to stackSpill
#endif
// HAS_TLAS refill kernel (pt_wf_trace_refill_tlas): waves per workgroup sharing one LDS copy of the top of the TLAS (1 = one-wave
// workgroups, nodes from memory), nodes in that copy (64 bytes each), CWBVH stack entries per lane kept in LDS by that variant.
// 8 waves x (4 x 512 B CWBVH stack + 8 x 128 B TLAS stack + 256 B exchange) + 400 x 64 B = 52.2 KB: three workgroups = 24 waves per CU.
// Measured (200-instance scene, 1080p / 8 spp, round 3): 1 wave 2,327 Mrays/s, 8 waves with all 399 nodes in LDS 2,155, 8 waves with
// one node in LDS 2,043 -- the copy is worth 6 %, the eight-wave workgroup costs 12 %: the TLAS walk is not bound by where its nodes
// come from.  Default 1; the stress build (csrc/Makefile) runs the 8-wave variant through the parity suite.
#ifndef PT_WF_TLAS_WG_WAVES
#define PT_WF_TLAS_WG_WAVES 1u
#endif
#ifndef PT_WF_TLAS_CACHE_NODES
#define PT_WF_TLAS_CACHE_NODES 400u
#endif
#ifndef PT_WF_TLAS_CHUNK
#define PT_WF_TLAS_CHUNK 64u     // slots per chunk the waves of such a workgroup take from their shared range (power of two <= 64 x ... <= PT_WF_RANGE)
#endif
#ifndef PT_WF_TLAS_BLAS_LDS_STACK
#define PT_WF_TLAS_BLAS_LDS_STACK (PT_WF_LDS_STACK < 4 ? PT_WF_LDS_STACK : 4)
#endif
// entries per slab row: the deepest overflow any trace kernel of the build can have
#define PT_WF_SPILL_ROW_ENTRIES (PT_BVH_STACK_SIZE - (PT_WF_TLAS_BLAS_LDS_STACK < PT_WF_LDS_STACK ? PT_WF_TLAS_BLAS_LDS_STACK : PT_WF_LDS_STACK))
struct PTWfBuffers {
    uint32_t* flags;            // [numSlots] packed state word
    uint32_t* rng;              // [numSlots]
    float4* ray[3];             // [2 * numSlots] each: ray records of kind 0 (bounce ray: origin, scatterPdf | direction, maxRoughness), 1 (environment NEE), 2 (light NEE)
    float4 *rad, *thr, *color;  // radiance, throughput, per-pixel sample sum
    float4 *envC, *lightC, *pthr;        // contributions of the two NEE rays and the throughput they apply to
    float4* hit;                // [numSlots] (t, u, v, triIndex bits) written by trace kind 0
    float4* hit2;               // [numSlots] HAS_TLAS only: (world hit position, instance index bits)
    float4* pixsum;             // [numSlots] schedule 4 only: a finished pixel's sample sum, indexed by PIXEL slot (the other arrays by context)
    // The float4 arrays above are carved back to back at a fixed stride: array k starts at f4base + k * f4stride
    // (order: PT_F4_*).  The trace kernels address them this way -- one base pointer instead of seven -- because at 8 waves/SIMD a
    // wave may hold 80 SGPRs, and every pointer costs two.
    float4* f4base;
    uint32_t f4stride;          // in float4 elements
    uint8_t* occl;              // [2][numSlots] written by trace kinds 1, 2
    uint2* stackSpill;          // [numSlots / 64 waves x 64 lanes][32 - PT_WF_LDS_STACK]: traversal-stack entries beyond the LDS part (refill / persistent trace kernels)
    uint32_t* tlasSpill;        // [numSlots][32]: HAS_TLAS refill kernel, TLAS-stack entries beyond its LDS part (allocated for HAS_TLAS scenes only)
    uint4* susp;                // [numSlots / 64][PT_WF_SUSPEND][6]: suspended rays of the refill trace kernel (pt_wavefront.hip)
    uint32_t* suspCount;        // [numSlots / 64]
    uint32_t* aliveFlags;       // [maxIterations][PT_WF_ALIVE_SHARDS] slots still alive after shade launch i of the pass (tail cut, pt_wavefront.hip)
    float4* recPlanes;          // [10][numSlots] shadeSorted == 2: stage-2 records (pt_wf_shade_pre / _hit / _post), float4 planes
    uint32_t* recDest;          // [numSlots] the record a slot's home lane was given (0xFFFFFFFF: none)
    uint32_t* recCount;         // [2] records allocated this iteration: surface hits (from the front), misses (from the back)
    uint32_t shadeSorted;       // wavefront shade launches: 2 = pre / hit / post launches over compacted records, 1 = pt_wf_shade_sorted (stage 2 regrouped through LDS), 0 = pt_wf_shade (slot order)
    uint32_t tailCut;           // later launches of a pass return at once when fewer slots than this are alive; pt_wf_cleanup finishes them
    uint32_t* chunkHeads;       // [8 shards x 32 words]: work counters of the persistent trace kernel, one 128-B line each
    unsigned long long* statRows;   // [numStatRows][16]
    uint32_t numSlots, numStatRows, maxIterations;
    uint32_t slotsPerPass;      // numSlots = passes of the batch x slotsPerPass; slot s belongs to pass s / slotsPerPass, pixel slot s % slotsPerPass
    uint32_t residentWaves;     // waves the device holds at the trace kernel's occupancy (CUs x 4 SIMDs x 8)
};

// Refill trace launches: 128 slots per wave while that still gives a quarter of the device's wave slots a wave (1080p: 16,320
// waves for 8,192 slots, +2.4 %; half and quarter frames: +2 %), 64 for smaller launches.  Shared by the launcher and by the
// arena sizing of pt_api.hip (a trace wave addresses 64 slab rows and PT_WF_SUSPEND records).
#ifndef PT_WF_WIDE_DIV
#define PT_WF_WIDE_DIV 4u       // with 12 sets in flight: 1/4 of a 1080p frame 5.09 vs 5.21 ms, 1/8 equal either way
#endif
inline bool pt_wf_wide_ranges(uint32_t numSlots, uint32_t residentWaves) { return (numSlots / 128u) >= residentWaves / PT_WF_WIDE_DIV; }
// most trace waves any schedule launches over a set of numSlots slots (refill: numSlots / 64 or / 128; persistent: <= residentWaves)
inline uint32_t pt_wf_max_trace_waves(uint32_t numSlots, uint32_t residentWaves)
{
    const uint32_t narrow = (numSlots + 63u) / 64u;
    if (!pt_wf_wide_ranges(numSlots, residentWaves)) return narrow;
    const uint32_t wide = (numSlots + 127u) / 128u, persist = residentWaves < narrow ? residentWaves : narrow;
    return wide > persist ? wide : persist;
}

// A BATCH of progressive passes rendered by ONE launch sequence (PTRenderPassBatchTo): the passes differ only in RngSeedRoot and
// CurrentSample, their paths are independent (the running mean of PathTracer.compute:89-98 only meets them in the resolve kernel,
// which applies the passes in order), so a launch covers count x slotsPerPass slots.
#define PT_MAX_BATCH 8
struct PTBatch {
    uint32_t count;
    uint32_t seedRoot[PT_MAX_BATCH];
    uint32_t currentSample[PT_MAX_BATCH];
};
__host__ __device__ inline void pt_batch_pick(const PTBatch& b, uint32_t j, uint32_t& seedRoot, uint32_t& currentSample)
{
    seedRoot = b.seedRoot[0]; currentSample = b.currentSample[0];
#pragma unroll
    for (uint32_t k = 1; k < PT_MAX_BATCH; ++k) if (j == k) { seedRoot = b.seedRoot[k]; currentSample = b.currentSample[k]; }
}

// pt_wavefront.hip is compiled twice with different scheduler flags (csrc/Makefile): the translation unit built with -DPT_WF_TU_B
// exports the same launcher under the name pt_launch_wavefront_b
#ifdef PT_WF_TU_B
#define pt_launch_wavefront pt_launch_wavefront_b
#endif
hipError_t pt_launch_wavefront(const DScene& S, const PTFrameParams& P, const PTBatch& batch, const float4* accumulated, float4* output,
                               const PTTileMap& tm, const PTWfBuffers& B, unsigned long long* gstats, bool fullStats,
                               hipStream_t stream, hipEvent_t orderAfter, bool zeroOutputFirst,
                               uint32_t* launchesOut, int traceVariant, uint32_t iterationsOverride);
#ifndef PT_WF_FUSED_GROUPS
#define PT_WF_FUSED_GROUPS 2u   // schedule 4: groups of 64 path contexts a persistent wave owns (power of two <= 4: numSlots is a multiple of 256).
                                // Sponza-class 1080p / 8 spp, one pass in flight: 1: 29.9 ms, 2: 25.2, 4: 27.9, 8: 31.4
#endif
#ifndef PT_WF_SUSPEND
#define PT_WF_SUSPEND 16u       // refill trace kernel: a wave whose range is exhausted stops when this many rays or fewer are left, and leaves them
                                // as records for the tail launch (pt_wavefront.hip); 0 = off.  Also the record slots per trace wave (pt_api.hip)
#endif
#ifndef PT_WF_SHADE_SORTED
#define PT_WF_SHADE_SORTED 0u     // shade kernel of the wavefront schedules (PTWfBuffers.shadeSorted): 0 = pt_wf_shade (slot order), 1 = pt_wf_shade_sorted.
                                  // Measured (Sponza-class 1080p / 8 spp): sorted issues 25 % fewer VALU instructions at 54 % instead of 38 % lane utilisation
                                  // and is 13 % SLOWER end to end -- see the kernel's header
#endif
#define PT_WF_ALIVE_SHARDS 64u   // words per iteration of PTWfBuffers.aliveFlags (power of two <= 64)
#ifndef PT_WF_TAIL_CUT
#define PT_WF_TAIL_CUT 0u        // 0 = never cut (every launch of the fixed sequence runs)
#endif
#ifndef PT_WF_SETS
#define PT_WF_SETS 12            // path-state sets = passes that can be in flight at once, each on its own stream (3 -> 6 sets with 8 hardware queues: +12 %;
                                 // 6 -> 12 sets with 16 queues: +2 % at 1080p, +7 % at 960x540, -14 % time for a 1/8 share of a 1080p frame; 16: no better, 24: worse)
#endif
