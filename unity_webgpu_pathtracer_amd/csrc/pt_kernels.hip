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

enum : int { ST_TRACE = 0, ST_ENDING = 1, ST_DONE = 2 };

PT_DEV uint32_t wave_sum(uint32_t v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
PT_DEV uint32_t wave_max(uint32_t v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { uint32_t o = __shfl_xor(v, off, 64); v = o > v ? o : v; }
    return v;
}

template <bool STATS>
__global__ __launch_bounds__(256) void pt_megakernel(DScene S, PTFrameParams P, const float4* __restrict__ accumulated,
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

    Counters cn = {};
    TravStack st;
    st.lds = &s_stack[0][tid];
    st.stride = 256u;

    const uint32_t pixelIndex = pixelY * P.OutputWidth + pixelX;
    const int numSamples = P.SamplesPerPass > 1 ? P.SamplesPerPass : 1;
    const uint32_t maxRayBounces = P.MaxRayBounces > 1u ? P.MaxRayBounces : 1u;

    uint32_t rng = pixelIndex * (P.CurrentSample + 1u) + P.RngSeedRoot;         // PathTracer.compute:60
    int sampleIdx = 0;
    int state = inside ? ST_TRACE : ST_DONE;

    v3 color = mk3(0.0f), radiance = mk3(0.0f), throughput = mk3(1.0f);
    v3 ro = mk3(0.0f), rd = mk3(0.0f, 0.0f, 1.0f);
    uint32_t depth = 0;
    float scatterPdf = 0.0f, maxRoughness = 0.0f;
    NeeRay envRay, lightRay;
    envRay.valid = 0u; lightRay.valid = 0u;
    envRay.dir = mk3(0.0f); lightRay.dir = mk3(0.0f);
    envRay.contribution = mk3(0.0f); lightRay.contribution = mk3(0.0f);
    v3 neeOrigin = mk3(0.0f), pendThroughput = mk3(0.0f);
    bool hasPending = false, green = false;

    if (inside) {
        cn.paths++;
        generate_camera_ray(P, pixelX, pixelY, rng, ro, rd);
    }

    while (state != ST_DONE) {
        // ------------------------------------------------------------------ traversal pass (<= 3 rays per lane)
        TraceHit ch;
        ch.t = PT_FAR_PLANE; ch.u = 0.0f; ch.v = 0.0f; ch.triIndex = 0u;
        bool occEnv = false, occLight = false;
#pragma nounroll
        for (int r = 0; r < 3; ++r) {
            bool active;
            v3 o, d;
            if (r == 0) { active = hasPending && envRay.valid != 0u; o = neeOrigin; d = envRay.dir; }
            else if (r == 1) { active = hasPending && lightRay.valid != 0u; o = neeOrigin; d = lightRay.dir; }
            else { active = state == ST_TRACE; o = ro; d = rd; }
            if (active) {
                TraceHit h;
                h.t = PT_FAR_PLANE; h.u = 0.0f; h.v = 0.0f; h.triIndex = 0u;
                traverse_cwbvh<STATS>(S, o, d, r < 2, h, st, cn);
                if (r == 0) { occEnv = h.t < PT_FAR_PLANE; cn.shadowRays++; }
                else if (r == 1) { occLight = h.t < PT_FAR_PLANE; cn.shadowRays++; }
                else { ch = h; cn.closestRays++; }
            }
        }

        // ------------------------------------------------------------------ apply the NEE of the previous shade
        if (hasPending) {                                   // radiance += DirectLight(...) * throughput   (pathtrace.hlsl:93)
            v3 Ld = mk3(0.0f);
            if (envRay.valid == 1u && !occEnv) Ld = Ld + envRay.contribution;
            if (lightRay.valid == 1u && !occLight) Ld = Ld + lightRay.contribution;
            radiance = radiance + Ld * pendThroughput;
            if (green) radiance = mk3(0.0f, 1.0f, 0.0f);    // NaN BSDF sample (pathtrace.hlsl:100-104)
            hasPending = false;
            green = false;
        }

        // ------------------------------------------------------------------ shade the closest hit (pathtrace.hlsl:27-127)
        if (state == ST_TRACE) {
            SurfHit hit;
            hit.distance = PT_FAR_PLANE;
            hit.isLight = 0u;
            hit.triIndex = 0u;
            hit.materialIndex = 0;
            hit.position = mk3(0.0f); hit.normal = mk3(0.0f); hit.ffnormal = mk3(0.0f);
            hit.uv = {0.0f, 0.0f};
            if (ch.t < PT_FAR_PLANE) fetch_hit_attributes<STATS>(S, ro, rd, ch, hit, cn);
            intersect_lights<STATS>(S, ro, rd, hit, cn);

            if (!(hit.distance < PT_FAR_PLANE)) {
                v4 sky = sample_sky_radiance(P, rd, depth);
                float misWeight = 1.0f;
                if (depth > 0u) misWeight = power_heuristic(scatterPdf, sky.w);
                if (misWeight > 0.0f) radiance = radiance + misWeight * mk3(sky.x, sky.y, sky.z) * throughput;
                state = ST_ENDING;
            } else if (S.hasLights && hit.isLight) {
                DLight light = load_light(S, (int)hit.triIndex);
                if (STATS) cn.lightFetches++;
                radiance = radiance + light.emission * throughput;
                state = ST_ENDING;
            } else {
                Material material = get_material<STATS>(S, hit.materialIndex, rd, hit, cn);
                maxRoughness = pt_max(maxRoughness, material.roughness);
                material.roughness = maxRoughness;
                radiance = radiance + material.emission * throughput;
                if (depth >= maxRayBounces) {
                    state = ST_ENDING;
                } else {
                    v3 scatterL;
                    bool skip = false;
                    if (material.alphaMode == (float)PT_ALPHA_MODE_MASK && material.opacity < material.alphaCutoff) skip = true;
                    else if (material.alphaMode == (float)PT_ALPHA_MODE_BLEND && rnd(rng) > material.opacity) skip = true;
                    bool advance = true;
                    if (skip) {
                        scatterL = rd;                       // rayDepth-- then ++: depth unchanged (pathtrace.hlsl:85-88)
                    } else {
                        neeOrigin = nee_scatter_pos(hit);
                        nee_prepare_environment(P, rd, hit, material, rng, envRay);
                        nee_prepare_light<STATS>(S, rd, hit, material, neeOrigin, rng, lightRay, cn);
                        pendThroughput = throughput;
                        hasPending = true;
                        float pdf;
                        v3 f = sample_brdf(material, -rd, hit.ffnormal, scatterL, pdf, rng);
                        scatterPdf = pdf;
                        if (pt_isnan(f.x) || pt_isnan(f.y) || pt_isnan(f.z)) {
                            green = true;
                            state = ST_ENDING;
                            advance = false;
                        } else if (pdf > 0.0f) {
                            throughput = throughput * (f / pdf);
                        } else {
                            state = ST_ENDING;
                            advance = false;
                        }
                        if (advance) depth++;
                    }
                    if (advance) {
                        rd = scatterL;
                        ro = hit.position + rd * PT_EPSILON;
                        if (P.UseRussianRoulette) {
                            float rrPcont = pt_min(pt_max(throughput.x, pt_max(throughput.y, throughput.z)) + 0.001f, 0.95f);
                            if (rnd(rng) >= rrPcont) state = ST_ENDING;
                            else throughput = throughput / rrPcont;
                        }
                    }
                }
            }
        }

        // ------------------------------------------------------------------ end of a sample (PathTracer.compute:77-86)
        if (state == ST_ENDING && !hasPending) {
            if (P.UseFireflyFilter) {
                float lum = luminance3(radiance);
                if (lum > P.MaxFireflyLuminance) radiance = radiance * (P.MaxFireflyLuminance / lum);
            }
            color = color + radiance;
            sampleIdx++;
            if (sampleIdx < numSamples) {
                cn.paths++;
                radiance = mk3(0.0f);
                throughput = mk3(1.0f);
                depth = 0u;
                scatterPdf = 0.0f;
                maxRoughness = 0.0f;
                generate_camera_ray(P, pixelX, pixelY, rng, ro, rd);
                state = ST_TRACE;
            } else {
                const float fSamples = (float)numSamples;
                float4 o4;
                if (P.CurrentSample > 0u) {                   // PathTracer.compute:89-94
                    float4 a = accumulated[pixelIndex];
                    cn.pixelsRead++;
                    float cs = (float)P.CurrentSample;
                    v3 acc = (color + mk3(a.x, a.y, a.z) * cs) / (cs + fSamples);
                    o4 = make_float4(acc.x, acc.y, acc.z, 1.0f);
                } else {
                    v3 acc = color / fSamples;
                    o4 = make_float4(acc.x, acc.y, acc.z, 1.0f);
                }
                output[pixelIndex] = o4;
                cn.pixelsWritten++;
                state = ST_DONE;
            }
        }
    }

    // ---- counters: one atomic per wave per counter
    uint32_t vals[14] = {cn.paths, cn.closestRays, cn.shadowRays, cn.nodeVisits, cn.triTests, cn.attrFetches,
                         cn.materialFetches, cn.lightFetches, cn.texelFetches, cn.texDescFetches,
                         cn.pixelsWritten, cn.pixelsRead, cn.maxStack, cn.overflows};
#pragma unroll
    for (int i = 0; i < 14; ++i) {
        if (!STATS && (i >= 3 && i <= 9)) continue;
        if (!STATS && i >= 12) continue;
        if (i == 12) {
            uint32_t m = wave_max(vals[i]);
            if (lane == 0 && m) atomicMax(&gstats[i], (unsigned long long)m);
        } else {
            uint32_t s = wave_sum(vals[i]);
            if (lane == 0 && s) atomicAdd(&gstats[i], (unsigned long long)s);
        }
    }
}

} // namespace

hipError_t pt_launch_megakernel(const DScene& S, const PTFrameParams& P, const float4* accumulated, float4* output,
                                const PTTileMap& tm, unsigned long long* gstats, bool fullStats, hipStream_t stream)
{
    dim3 grid((tm.blocksX + tm.world - 1) / tm.world, tm.blocksY, 1);
    dim3 block(256, 1, 1);
    if (fullStats) hipLaunchKernelGGL(pt_megakernel<true>, grid, block, 0, stream, S, P, accumulated, output, tm, gstats);
    else hipLaunchKernelGGL(pt_megakernel<false>, grid, block, 0, stream, S, P, accumulated, output, tm, gstats);
    return hipGetLastError();
}
