// pt_kernels.hip — render kernels for MI355X (gfx950) and their launchers.
//
// Schedule 0 ("megakernel"): one lane per pixel, one wave per 8x8-pixel tile, 256-thread workgroups
// covering 16x16-pixel blocks.  Each lane runs the reference's per-pixel program
// (PathTracer.compute:45-100 + util/pathtrace.hlsl:10-131) as a small state machine whose only
// traversal code is ONE CWBVH loop per iteration that serves, back to back, the (up to) two NEE shadow
// rays of the bounce just shaded and the next bounce ray.  That keeps a single copy of the divergent
// loop in the instruction stream, lets any-hit and closest-hit lanes share it, and keeps the per-lane
// operation order -- and therefore every fp32 value and every RNG draw -- identical to the reference.
#include "pt_device.h"
#include "pt_launch.h"

namespace {

#ifndef PT_MK_MIN_WAVES
#define PT_MK_MIN_WAVES 4      // waves per SIMD the register allocator must leave room for (4 = 128 VGPRs; measured best trade)
#endif

template <bool STATS, bool TLAS>
__global__ __launch_bounds__(256, PT_MK_MIN_WAVES) void pt_megakernel(DScene S, PTFrameParams P, const float4* __restrict__ accumulated,
                                                     float4* __restrict__ output, PTTileMap tm,
                                                     unsigned long long* __restrict__ gstats)
{
    __shared__ uint2 s_stack[PT_LDS_STACK][256];

    // ---- block -> 16x16 pixel block (interleaved ownership across ranks, include/ptmi_plugin.h PTSetTileOwnership)
    const uint32_t by = blockIdx.y;
    const uint32_t first = (uint32_t)(((int)tm.rank - (int)(by % tm.world) + (int)tm.world) % (int)tm.world);
    const uint32_t bx = first + blockIdx.x * tm.world;
    if (bx >= tm.blocksX) return;

    const uint32_t tid = threadIdx.x;
    const uint32_t wave = tid >> 6, lane = tid & 63u;
    const uint32_t pixelX = bx * 16u + (wave & 1u) * 8u + (lane & 7u);
    const uint32_t pixelY = by * 16u + (wave >> 1) * 8u + (lane >> 3);
    const bool inside = pixelX < tm.coverW && pixelY < tm.coverH;
    const uint32_t pixelIndex = pixelY * P.OutputWidth + pixelX;

    Counters cn = {};
    TravStack st;
    st.lds = PT_LDS_U2(&s_stack[0][tid]);
    st.stride = 256u;

    PathRegs r;
    r.ro = mk3(0.0f); r.rd = mk3(0.0f, 0.0f, 1.0f);
    r.state = PS_DONE;
    if (inside) path_init(P, P.RngSeedRoot, P.CurrentSample, pixelX, pixelY, pixelIndex, r, cn);

    while (r.state != PS_DONE) {
        // ---- traversal pass (<= 3 rays per lane); ray slots rotate through registers: slot 0 is traced, slots 1, 2 move down
        HitRecord ch;
        ch.h.t = PT_FAR_PLANE; ch.h.u = 0.0f; ch.h.v = 0.0f; ch.h.triIndex = 0u;
        ch.pos = mk3(0.0f); ch.inst = 0u;
        bool occEnv = false, occLight = false;
        bool a0 = r.hasPending && r.env.valid != 0u, a1 = r.hasPending && r.light.valid != 0u, a2 = r.state == PS_TRACE;
        v3 o0 = r.neeOrigin, d0 = r.env.dir, o1 = r.neeOrigin, d1 = r.light.dir, o2 = r.ro, d2 = r.rd;
#pragma nounroll
        for (int k = 0; k < 3; ++k) {
            const bool active = a0;
            const v3 o = o0, d = d0;
            a0 = a1; o0 = o1; d0 = d1;
            a1 = a2; o1 = o2; d1 = d2;
            if (active) {
                HitRecord h;
                h.h.t = PT_FAR_PLANE; h.h.u = 0.0f; h.h.v = 0.0f; h.h.triIndex = 0u;
                h.pos = mk3(0.0f); h.inst = 0u;
                bool occluded;
                if (TLAS) occluded = traverse_tlas<STATS>(S, o, d, k < 2, h, st, cn);          // util/tlas.hlsl:334-352
                else { traverse_cwbvh<STATS>(S, o, d, k < 2, h.h, st, cn); occluded = h.h.t < PT_FAR_PLANE; }
                if (k == 0) { occEnv = occluded; cn.shadowRays++; }
                else if (k == 1) { occLight = occluded; cn.shadowRays++; }
                else { ch = h; cn.closestRays++; }
            }
        }
        path_step<STATS>(S, P, r, ch, occEnv, occLight, pixelX, pixelY, pixelIndex, accumulated, output, cn);
    }

    // ---- counters: one atomic per wave per counter
    uint32_t vals[PT_NUM_COUNTERS];
    counters_to_array(cn, vals);
#pragma unroll
    for (int i = 0; i < PT_NUM_COUNTERS; ++i) {
        if (!STATS && (i >= 3 && i <= 9)) continue;
        if (!STATS && i >= 12) continue;
        if (i == 12) {
            uint32_t m = wave_max_u32(vals[i]);
            if (lane == 0 && m) atomicMax(&gstats[i], (unsigned long long)m);
        } else {
            uint32_t s = wave_sum_u32(vals[i]);
            if (lane == 0 && s) atomicAdd(&gstats[i], (unsigned long long)s);
        }
    }
}

// one lane per light: the constants of derive_light_rows (pt_device.h), evaluated once per scene
__global__ void pt_derive_lights(DScene S, float4* lightConst)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= S.lightCount) return;
    float4 rows[4];
    derive_light_rows(load_light(S, i), rows);
#pragma unroll
    for (int k = 0; k < 4; ++k) lightConst[(size_t)i * 4 + k] = rows[k];
}

} // namespace

hipError_t pt_launch_derive_lights(const DScene& S, float4* lightConst, hipStream_t stream)
{
    if (S.lightCount <= 0) return hipSuccess;
    hipLaunchKernelGGL(pt_derive_lights, dim3((S.lightCount + 63) / 64), dim3(64), 0, stream, S, lightConst);
    return hipGetLastError();
}

hipError_t pt_launch_megakernel(const DScene& S, const PTFrameParams& P, const float4* accumulated, float4* output,
                                const PTTileMap& tm, unsigned long long* gstats, bool fullStats, hipStream_t stream)
{
    dim3 grid((tm.blocksX + tm.world - 1) / tm.world, tm.blocksY, 1);
    dim3 block(256, 1, 1);
    if (S.hasTlas) {
        if (fullStats) hipLaunchKernelGGL((pt_megakernel<true, true>), grid, block, 0, stream, S, P, accumulated, output, tm, gstats);
        else hipLaunchKernelGGL((pt_megakernel<false, true>), grid, block, 0, stream, S, P, accumulated, output, tm, gstats);
    } else {
        if (fullStats) hipLaunchKernelGGL((pt_megakernel<true, false>), grid, block, 0, stream, S, P, accumulated, output, tm, gstats);
        else hipLaunchKernelGGL((pt_megakernel<false, false>), grid, block, 0, stream, S, P, accumulated, output, tm, gstats);
    }
    return hipGetLastError();
}
