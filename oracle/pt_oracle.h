/*
 * oracle/pt_oracle.h — C interface of the CPU oracle (TEST INFRASTRUCTURE ONLY).
 *
 * The oracle is a scalar CPU restatement of the reference's compute-shader path
 * (Assets/Resources/PathTracer.compute + Assets/Resources/util/ *.hlsl).  It is the checker the
 * MI355X kernels are compared against; it is never shipped, never linked into the product
 * library, and only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * Parity pinning: the reference has NO tests, golden vectors or fixtures of its
 * own, and its HLSL cannot be executed in the build container (no dxc/tint/Unity).  The oracle is
 * therefore pinned by (1) byte-exact CWBVH/TLAS data from the reference plugin compiled in place
 * (oracle/_ref), (2) the restated CWBVH traversal against the reference's OWN walker
 * BVH8_CWBVH::Intersect, bit for bit in (t, u, v, primitive) (oracle/ref_driver_cwbvh.cpp), and against
 * BVH::Intersect / IsOccluded (oracle/ref_driver.cpp), (3) the RNG known-answer vectors of SURVEY.md
 * Appendix D, (4) per-lobe distribution tests of the BSDF (pdf integrates to the usable fraction, sampled
 * directions follow it) and of SampleEnvMap, (5) closed-form images and one-bounce radiances
 * (tests/test_oracle_analytic.py: camera, thin lens, light hit, textures, spot / range falloff, sky,
 * rectangle / point / spot / uniform-sky lighting) and value, pdf and sampled direction of the BSDF
 * pointwise against a second, float64 transcription, (6) the same frames from a second compiler.
 * What stays UNPINNED (nothing here can execute HLSL): rounding of the shader's intrinsics, the
 * evaluation-order rules of DESIGN.md section 3 (the alpha test's unconditional draw among them) and
 * Unity's texture sampler; DESIGN.md section 7 says so.
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H

#include <stdint.h>
#include "ptmi_plugin.h"

#ifdef __cplusplus
extern "C" {
#endif

/* One traced ray as the oracle saw it (for the CPU-baseline ray list). */
typedef struct OracleRay {
    float origin[3];
    float direction[3];
    float tmax;
    float kind;        /* 0 = closest hit (RayIntersect), 1 = shadow (ShadowRayIntersect) */
} OracleRay;

typedef struct OracleOptions {
    int32_t  x0, y0, x1, y1;      /* pixel window [x0,x1) x [y0,y1); all zero = whole frame        */
    int32_t  threads;             /* worker threads (rows are interleaved); <= 0 -> 1               */
    int32_t  shadowAnyHit;        /* 0 = reference-faithful traversal;
                                     1 = count work as the GPU kernels do it: shadow rays stop at the first
                                     accepted triangle, and rays carrying a NaN return their (certain) miss
                                     without walking the tree.  Same frame bit for bit; only nodeVisits /
                                     triTests / maxStackDepth change (used to check the GPU's counters)  */
    int32_t  tileRank, tileWorld; /* PTSetTileOwnership emulation: non-owned pixels are written 0   */
    OracleRay* rayDump;           /* optional: receives up to rayDumpCapacity rays (single thread!) */
    uint64_t rayDumpCapacity;
    uint64_t rayDumpCount;        /* out */
} OracleOptions;

/* Renders one pass exactly as one DispatchCompute of the reference kernel would
 * (PathTracer.compute:45-100).  `accumulated` may be NULL when CurrentSample == 0.
 * `output` = width*height float4; pixels outside the window are left untouched. */
int oracle_render(const PTSceneDesc* scene, const PTFrameParams* params, const float* accumulated,
                  float* output, PTStats* stats, OracleOptions* opt);

/* Traces explicit rays through the restated CWBVH traversal (util/bvh.hlsl:126-215).
 * rays: n x OracleRay.  outT: hit distance (tmax on a miss); outPrim: hit.triIndex (0xFFFFFFFF on a miss);
 * outSteps: nodes visited.  Shadow rays (kind 1) report prim = 1 / t = 0 when occluded. */
int oracle_trace(const PTSceneDesc* scene, const OracleRay* rays, uint64_t n, float* outT,
                 uint32_t* outPrim, uint32_t* outSteps);

/* The same traversal reporting the full closest-hit record per ray: out4[i] = {t, u, v, primitive bits} (0xFFFFFFFF, u = v = 0 on a
 * miss), and the node visits / triangle tests of the batch.  Compared bit for bit with the reference's own CWBVH walker
 * (BVH8_CWBVH::Intersect, tiny_bvh.h:7006-7114; oracle/ref_driver_cwbvh.cpp) by tests/test_oracle.py. */
int oracle_trace_uv(const PTSceneDesc* scene, const OracleRay* rays, uint64_t n, float* out4, uint64_t* outNodeVisits, uint64_t* outTriTests);

/* Scalar entry points for unit tests. */
float    oracle_random_float(uint32_t* state);                       /* util/random.hlsl:12-16 */
float    oracle_math(int fn, float x, float y);                      /* ptmi_math.h: 0 sin 1 cos 2 log 3 log2 4 exp2 5 pow 6 acos 7 asin 8 sqrt 9 rcp 10 atan2(x = y-arg, y = x-arg) 11 fmod 12 wrap01 (closed form) 13 wrap01 (literal loops) */
/* Evaluates / samples the Disney BSDF in isolation (util/brdf.hlsl:114-340).
 * material: 32 floats of MaterialData (no textures); V, N, L world-space; returns f via out[0..2], pdf out[3]. */
void     oracle_eval_brdf(const float* material, const float* V, const float* N, const float* L, float eta, float* out);
void     oracle_sample_brdf(const float* material, const float* V, const float* N, float eta, uint32_t* rngState, float* outL, float* outFPdf);
/* Batch forms for the distribution tests (per lobe: pdf integrates to the probability of a usable sample; samples follow the pdf):
 * L: n x 3 -> out: n x 4; n consecutive SampleBRDF calls on ONE RNG chain -> outL: n x 3, outFPdf: n x 4. */
void     oracle_eval_brdf_batch(const float* material, const float* V, const float* N, float eta, const float* L, uint64_t n, float* out);
void     oracle_sample_brdf_batch(const float* material, const float* V, const float* N, float eta, uint32_t* rngState, uint64_t n, float* outL, float* outFPdf);

/* Environment-texture functions in isolation (util/sky.hlsl:7-88), `scene` with HAS_ENVIRONMENT_TEXTURE:
 *   what 0: EnvironmentCDF[i] for i = in[k] (as float index)            -> out[k]            (n values)
 *   what 1: BinarySearch(in[k])                                         -> out[2k..2k+1]     (uv)
 *   what 2: EvalEnvMap(in[3k..3k+2], intensity 1)                       -> out[4k..4k+3]     (rgb, pdf)
 *   what 3: SampleEnvMap with rngState = (uint)in[k] bits               -> out[8k..8k+7]     (dir xyz, pdf, colour rgb, 0)
 * Returns EnvironmentCdfSum. */
float    oracle_env_probe(const PTSceneDesc* scene, const PTFrameParams* params, int what, const float* in, uint64_t n, float* out);

/* Scene ingestion: one Dispatch of MeshProcessing.compute:59-139 (writes triangles [OutputTriangleStart, +TriangleCount) of both
 * output buffers) and the texture loop of BVHScene.cs:386-417 over CopyTextureData.compute:13-36. */
int      oracle_process_mesh(const PTMeshDesc* mesh, float* vertexPositionBuffer, void* triangleAttributesBuffer);
int      oracle_copy_texture_data(const PTTextureDesc* textures, uint32_t count, uint32_t* textureData);

/* The presentation blit, Assets/Resources/Presentation.shader:36-73 + util/tonemap.hlsl, one pixel after the other. */
int      oracle_present(const PTPresentParams* params, const float* srcRGBA, float* dstRGBA);

#ifdef __cplusplus
}
#endif
#endif
